"""Dev tool: stage times (HIP events) and wall time of one EI step at the C2 size, ei_flow off / on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for (N, M, D, H) in ((256, 20000, 8, 10), (2048, 200000, 32, 20)):
    comp, cand, vals, hyp = synthetic_problem(N, M, D, H, 11)
    for flow in (0, 1):
        eng.set_option("ei_flow", flow)
        eng.ei_grid(comp, vals, cand, hyp, want_mean=False)
        reps = 20 if N < 1000 else 2
        t = time.time()
        for _ in range(reps):
            eng.ei_grid(comp, vals, cand, hyp, want_mean=False)
        wall = (time.time() - t) / reps * 1e3
        eng.set_option("timing", 1)
        for _ in range(3):
            eng.ei_grid(comp, vals, cand, hyp, want_mean=False)
        tm = eng.timings(); eng.set_option("timing", 0)
        print("N=%d ei_flow=%d wall %.3f ms | " % (N, flow, wall) + "  ".join("%s %.3f (%d)" % (k, v[0] / 3, v[1] // 3) for k, v in tm.items() if v[1]))
eng.set_option("ei_flow", -1)

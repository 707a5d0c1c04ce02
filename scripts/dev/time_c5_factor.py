"""Dev tool: factor stage of the per-second EI path (2 H draws) at the C5 size, ei_flow off / on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for (N, M, D, H) in ((1024, 62500, 16, 20), (512, 20000, 8, 30), (2048, 20000, 32, 20)):
    comp, cand, vals, hyp, ld, th = synthetic_problem(N, M, D, H, 11, per_sec=True)
    for flow in (0, 1):
        eng.set_option("ei_flow", flow)
        eng.ei_per_sec_grid(comp, vals, ld, cand, hyp, th, want_mean=False)
        eng.set_option("timing", 1)
        for _ in range(3):
            eng.ei_per_sec_grid(comp, vals, ld, cand, hyp, th, want_mean=False)
        tm = eng.timings(); eng.set_option("timing", 0)
        print("N=%d 2H=%d ei_flow=%d | " % (N, 2 * H, flow) + "  ".join("%s %.3f (%d)" % (k, v[0] / 3, v[1] // 3) for k, v in tm.items() if v[1] and k in ("cov_self", "chol_diag", "chol_panel", "trinv", "factor_total", "ei_run_total")))
eng.set_option("ei_flow", -1)

"""Dev tool: spx_gp_logprob wall time per call by batch size (default modes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1024, 16)):
    row = []
    for H in (1, 2, 4, 6, 8, 12):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
        t = time.time()
        for _ in range(30):
            eng.set_hypers(hypers); eng.gp_logprob()
        row.append("H=%d %.3f" % (H, (time.time() - t) / 30 * 1e3))
    print("N=%d  " % N + "  ".join(row))

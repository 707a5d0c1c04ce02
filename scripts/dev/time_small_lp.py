"""Dev: spx_gp_logprob wall time per call at small N (python scripts/dev/time_small_lp.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
if len(sys.argv) > 1:            # e.g. lean_merge=0
    k, v = sys.argv[1].split("=")
    eng.set_option(k, int(v))
    print("option %s = %s" % (k, v))
for N, D in ((10, 2), (20, 2), (50, 4), (64, 8), (65, 8), (128, 8), (129, 8), (192, 8), (200, 8), (256, 8), (320, 8)):
    line = "N=%3d D=%2d |" % (N, D)
    for H in (1, 6, 12):
        comp, cand, vals, hyp = synthetic_problem(N, 10, D, H, 5)
        eng.set_observations(comp, vals)
        best = 1e9
        for rnd in range(3):
            eng.set_hypers(hyp); eng.gp_logprob()
            t = time.time()
            for _ in range(200):
                eng.set_hypers(hyp); eng.gp_logprob()
            best = min(best, (time.time() - t) / 200 * 1e6)
        line += "  H=%2d %.1f us" % (H, best)
    print(line, flush=True)

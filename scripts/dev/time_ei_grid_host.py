"""Dev: spx_ei_grid (host buffers in, results out) call by call at a typical Spearmint size: cold and warm.   python scripts/dev/time_ei_grid_host.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for (N, M, D, H) in ((40, 20000, 4, 10), (40, 20, 4, 10), (40, 20000, 4, 10), (2048, 200000, 32, 20)):
    comp, cand, vals, hyp = synthetic_problem(N, M, D, H, 3)
    ts = []
    for rep in range(6):
        t = time.time()
        eng.ei_grid(comp, vals, cand, hyp, want_mean=True, want_draws=True)
        ts.append((time.time() - t) * 1e3)
    t = time.time(); eng.ei_grid(comp, vals, cand, hyp, want_mean=False, want_draws=False); nores = (time.time() - t) * 1e3
    print("N=%d M=%d D=%d H=%d: ei_grid calls (mean + draws fetched) %s ms; without fetching %.3f ms" % (N, M, D, H, " ".join("%.3f" % x for x in ts), nores), flush=True)

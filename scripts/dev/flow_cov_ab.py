"""Dev tool: k_lean_flow with the tiles of K(X,X) built in the launch (option lean_flow_cov=1, default) against k_cov first (0).
   python scripts/dev/flow_cov_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1000, 16)):
    for H in (1, 4, 6, 8, 12):
        comp, cand, vals, hyp = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals)
        out = []
        for m in (1, 0, 1, 0):
            eng.set_option("lean_flow_cov", m)
            for _ in range(5):
                eng.set_hypers(hyp); eng.gp_logprob()
            ts = []
            for _ in range(60):
                t = time.perf_counter(); eng.set_hypers(hyp); eng.gp_logprob(); ts.append(time.perf_counter() - t)
            out.append(np.median(ts) * 1e3)
        print("N=%d H=%2d  in-launch tiles %.3f / %.3f ms   k_cov first %.3f / %.3f ms" % (N, H, out[0], out[2], out[1], out[3]))

"""Dev tool (GPU box): one log-likelihood evaluation on the host (hostgp, numpy/scipy) vs on the GPU, small N --
where `gpu_logprob=auto` should switch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd import hostgp
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((8, 2), (16, 2), (24, 4), (32, 4), (48, 4), (64, 8), (96, 8), (128, 8)):
    comp, cand, vals, hyp = synthetic_problem(N, 10, D, 6, 3)
    h = hyp[0]
    for _ in range(20):
        hostgp.data_logprob(comp, vals, h[0], h[2], h[1], h[3:], "Matern52")
    t = time.time()
    for _ in range(300):
        hostgp.data_logprob(comp, vals, h[0], h[2], h[1], h[3:], "Matern52")
    th = (time.time() - t) / 300 * 1e3
    eng.set_observations(comp, vals)
    out = []
    for rows in (1, 6):
        eng.set_hypers(hyp[:rows]); eng.gp_logprob()
        t = time.time()
        for _ in range(300):
            eng.set_hypers(hyp[:rows]); eng.gp_logprob()
        out.append((time.time() - t) / 300 * 1e3)
    print("N=%3d  host %.3f ms per evaluation | GPU %.3f ms per call of 1 row, %.3f ms per call of 6 rows (%.3f per row)"
          % (N, th, out[0], out[1], out[1] / 6))

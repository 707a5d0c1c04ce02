"""Dev tool (GPU box): the refinement objective (EI + gradient summed over the hyper draws) for a batch of points on the
host (hostgp.PointModel, one point at a time) vs on the GPU (one spx_ei_grad_batch call), small N -- where
`gpu_refine=auto` should switch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd import hostgp
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
P, H = 20, 10
for N, D in ((8, 2), (16, 2), (24, 4), (32, 4), (48, 4), (64, 8), (96, 8), (128, 8)):
    comp, cand, vals, hyp = synthetic_problem(N, 1000, D, H, 3)
    pts = cand[:P].copy()
    hs = [(h[0], h[1], h[2], h[3:]) for h in hyp]
    models = [hostgp.PointModel(comp, vals, h, "Matern52") for h in hs]
    def host_all():
        for i in range(P):
            tot, g = 0.0, np.zeros(D)
            for m in models:
                e, gg = m.neg_ei_and_grad(pts[i]); tot += e; g = g + gg
    host_all()
    t = time.time()
    for _ in range(5): host_all()
    th = (time.time() - t) / 5 * 1e3
    t = time.time()
    for _ in range(5): models = [hostgp.PointModel(comp, vals, h, "Matern52") for h in hs]
    tsetup = (time.time() - t) / 5 * 1e3
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp); eng.factor()
    eng.ei_grad_batch(pts)
    t = time.time()
    for _ in range(50): eng.ei_grad_batch(pts)
    tg = (time.time() - t) / 50 * 1e3
    t = time.time()
    for _ in range(50): eng.ei_grad_batch(pts[:1])
    tg1 = (time.time() - t) / 50 * 1e3
    print("N=%3d H=%d  host: %d points %.2f ms (%.3f per point; models built in %.2f ms) | GPU: one call for %d points %.3f ms, for 1 point %.3f ms"
          % (N, H, P, th, th / P, tsetup, P, tg, tg1))

"""Dev: the refinement objective (spx_ei_grad_batch) per call -- plain, per second, with fantasies -- and a whole lbfgs_many.
   python scripts/dev/time_refine_paths.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd import refine
from spearmint_amd.engine import Engine, FLAG_PER_SEC
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for (N, D, H) in ((2048, 32, 20), (300, 6, 10), (40, 4, 10)):
    prob = synthetic_problem(N, 2000, D, H, 9, per_sec=True)
    comp, cand, vals, hyp, ld, th = prob
    rs = np.random.RandomState(0)
    out = []
    # plain
    eng.ei_grid(comp, vals, cand, hyp)
    for P in (1, 20):
        eng.ei_grad_batch(cand[:P]); t = time.time()
        for _ in range(10): eng.ei_grad_batch(cand[:P])
        out.append("plain P=%d %.3f ms" % (P, (time.time() - t) / 10 * 1e3))
    t = time.time(); refine.lbfgs_many(eng.ei_grad_batch, cand[:20], [(0, 1)] * D); out.append("lbfgs_many(20 points) %.1f ms" % ((time.time() - t) * 1e3))
    # per second
    eng.ei_per_sec_grid(comp, vals, ld, cand, hyp, th)
    eng.ei_grad_batch(cand[:20]); t = time.time()
    for _ in range(10): eng.ei_grad_batch(cand[:20])
    out.append("per-sec P=20 %.3f ms" % ((time.time() - t) / 10 * 1e3))
    # fantasies
    S = 100
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp); eng.factor()
    fant = rs.randn(H, N, S) * 0.1 + vals[None, :, None]
    eng.set_fantasies(fant, fant.min(axis=1)); eng.ei_run()
    t = time.time(); eng.ei_grad_batch(cand[:20]); first = (time.time() - t) * 1e3
    t = time.time()
    for _ in range(10): eng.ei_grad_batch(cand[:20])
    out.append("fantasies S=100 P=20 %.3f ms (first call %.1f ms)" % ((time.time() - t) / 10 * 1e3, first))
    print("N=%d D=%d H=%d | " % (N, D, H) + "  ".join(out), flush=True)

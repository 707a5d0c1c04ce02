"""Odd shapes / repeated calls on one handle (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
from oracle import gp_ei_oracle as orc
eng = Engine(0)
for (N, M, D, H, seed) in [(5, 3, 1000, 2, 1), (4097, 700, 3, 1, 2), (33, 2_000_000, 2, 2, 3), (260, 129, 70, 3, 4),
                           (2, 1, 1, 1, 5), (640, 5000, 12, 21, 6), (100, 100, 4, 130, 7)]:
    comp, cand, vals, hyp = synthetic_problem(N, M, D, H, seed, near=min(10, M))
    t = time.time()
    idx, val, mean, dr = eng.ei_grid(comp, vals, cand, hyp, want_draws=(M <= 10000))
    dt = time.time() - t
    sub = np.arange(min(M, 300))
    ref = orc.ei_over_hypers(comp, cand[sub], vals, hyp)
    got = dr[sub] if dr is not None else None
    if got is None:
        eng2 = eng.ei_grid(comp, vals, cand[sub], hyp, want_draws=True)[3]; got = eng2
    ok = ref > 1e-280
    err = np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) if ok.any() else 0.0
    print("N=%d M=%d D=%d H=%d: %.3fs  max rel err %.2e  mean==np.mean %s" % (N, M, D, H, dt, err,
          np.array_equal(mean, np.mean(dr, axis=1)) if dr is not None else "n/a"))
print("ok")

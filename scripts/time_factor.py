"""Per-kernel times of the factorisation for several batch sizes H (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N in (2048, 1024):
    for H in (1, 2, 4, 8, 20):
        comp, cand, vals, hypers = synthetic_problem(N, 1024, 32, H, 5)
        eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
        eng.factor()
        eng.set_option("timing", 1)
        for _ in range(3):
            eng.factor()
        tm = eng.timings()
        eng.set_option("timing", 0)
        print("N=%d H=%2d " % (N, H) + "  ".join("%s %.3f" % (k, tm[k][0] / 3) for k in
              ("cov_self", "chol_diag", "chol_panel", "trinv", "gamma_alpha", "factor_total")))

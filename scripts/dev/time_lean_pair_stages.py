import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for pair in (0, 1):
    eng.set_option("lean_pair", pair)
    for N, D in ((2048, 32),):
        for H in (1, 8):
            comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
            eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
            eng.set_option("timing", 1)
            for _ in range(5):
                eng.set_hypers(hypers); eng.gp_logprob()
            tm = eng.timings()
            eng.set_option("timing", 0)
            print("pair=%d N=%d H=%2d | " % (pair, N, H) + "  ".join("%s %.3f (%d)" % (k, tm[k][0] / 5, tm[k][1] // 5) for k in
                  ("scale_rows", "cov_self", "chol_diag", "chol_panel", "gamma_alpha", "factor_total") if tm[k][1]))

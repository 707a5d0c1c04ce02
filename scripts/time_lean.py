"""Log-likelihood path (spx_gp_logprob): wall time per call and per-stage HIP-event times (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
if os.environ.get('SPX_LEAN_PS'):
    eng.set_option('lean_ps', int(os.environ['SPX_LEAN_PS']))
for N, D in ((2048, 32), (1024, 16), (256, 8)):
    for H in (1, 4, 8, 20):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
        t = time.time()
        for _ in range(10):
            eng.set_hypers(hypers); eng.gp_logprob()
        wall = (time.time() - t) / 10 * 1e3
        eng.set_option("timing", 1)
        for _ in range(5):
            eng.set_hypers(hypers); eng.gp_logprob()
        tm = eng.timings()
        eng.set_option("timing", 0)
        print("N=%d H=%2d wall %.3f ms | " % (N, H, wall) + "  ".join(
            "%s %.3f (%d)" % (k, tm[k][0] / 5, tm[k][1] // 5) for k in
            ("scale_rows", "cov_self", "chol_diag", "chol_panel", "gamma_alpha", "factor_total") if tm[k][1]))

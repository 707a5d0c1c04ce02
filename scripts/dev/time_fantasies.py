"""Dev: the EI pass with pending-experiment fantasies (spx_set_fantasies) against the plain pass, stage times.
   python scripts/dev/time_fantasies.py [N M D H S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
args = [int(a) for a in sys.argv[1:]]
eng = Engine(0)
for (N, M, D, H, S) in ([tuple(args)] if len(args) == 5 else [(2048, 200000, 32, 20, 100), (1024, 100000, 16, 10, 100), (256, 20000, 8, 10, 100), (100, 20000, 4, 10, 100)]):
    P = 4
    comp, cand, vals, hyp = synthetic_problem(N + P, M, D, H, 21)
    rs = np.random.RandomState(1)
    fant = rs.randn(H, N + P, S) * 0.1 + vals[None, :, None]
    bests = fant.min(axis=1)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp)
    eng.factor(); eng.ei_run()
    t = time.time(); eng.ei_run(); plain = (time.time() - t) * 1e3
    eng.set_fantasies(fant, bests)
    eng.ei_run()
    t = time.time(); eng.ei_run(); withf = (time.time() - t) * 1e3
    eng.set_option("timing", 1); eng.ei_run(); tm = eng.timings(); eng.set_option("timing", 0)
    print("N=%d+%d M=%d D=%d H=%d S=%d: plain pass %.2f ms, with fantasies %.2f ms (x%.2f) | %s" % (
        N, P, M, D, H, S, plain, withf, withf / plain,
        "  ".join("%s %.2f (%d)" % (k, v[0], v[1]) for k, v in tm.items() if v[1] and k in ("cov_cross", "predict_gemm", "ei_finalize", "scale_rows"))), flush=True)

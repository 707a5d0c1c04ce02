"""Dev tool: spx_gp_logprob per call -- three launches (lean_one=0) vs one (lean_one=1), with / without the zero-copy hyper
rows (lean_zc) and the polled completion (lean_poll) -- by size and batch.  Values must not change."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
FORMS = (("3 launches", (0, 0, 0)), ("one", (1, 0, 0)), ("one+zc", (1, 1, 0)), ("one+poll", (1, 0, 1)), ("one+zc+poll", (1, 1, 1)))
print("us per call: " + " | ".join(f[0] for f in FORMS))
for N, D in ((20, 2), (64, 8), (128, 8), (256, 8), (512, 8), (1024, 16), (2048, 32)):
    for H in (1, 8, 24):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals)
        res = []
        for name, (one, zc, poll) in FORMS:
            eng.set_option("lean_one", one); eng.set_option("lean_zc", zc); eng.set_option("lean_poll", poll)
            eng.set_hypers(hypers); a = eng.gp_logprob()
            reps = 300 if N <= 512 else 40
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                for _ in range(reps):
                    eng.set_hypers(hypers); eng.gp_logprob()
                best = min(best, (time.perf_counter() - t) / reps)
            res.append((best * 1e6, a))
        same = all(np.array_equal(r[1], res[0][1]) for r in res)
        print("N=%4d D=%2d H=%2d  " % (N, D, H) + " | ".join("%7.1f" % r[0] for r in res) + ("" if same else "   DIFF"), flush=True)
for o in ("lean_one", "lean_zc", "lean_poll"):
    eng.set_option(o, -1)

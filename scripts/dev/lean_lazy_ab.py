"""Dev: lazy (two steps per pass) vs eager trailing updates in the log-likelihood path -- identical bits, wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1024, 16), (4096, 8), (300, 8), (64, 3), (130, 2)):
    for H in (1, 2, 4, 6, 8, 12, 20):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        if H >= 4:
            hypers[1, 2] = -1.0        # one draw not positive definite
        eng.set_observations(comp, vals)
        res = {}
        for mode, v in (("eager", 0), ("lazy", 1)):
            eng.set_option("lean_lazy", v)
            eng.set_hypers(hypers); lp = eng.gp_logprob()
            t = time.time()
            for _ in range(20):
                eng.set_hypers(hypers); eng.gp_logprob()
            res[mode] = (lp, (time.time() - t) / 20 * 1e3)
        same = np.array_equal(res["lazy"][0], res["eager"][0], equal_nan=True)
        print("N=%4d H=%2d  eager %.3f ms  lazy %.3f ms  identical bits %s" % (N, H, res["eager"][1], res["lazy"][1], same))

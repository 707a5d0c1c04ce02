"""Dev tool: spx_gp_logprob wall time per call by batch size for the lazy / eager trailing updates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1024, 16), (512, 16)):
    for H in (1, 2, 3, 4, 6, 8, 12, 20):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals)
        out = []
        ref = None
        for lazy in (0, 1):
            eng.set_option("lean_lazy", lazy)
            eng.set_hypers(hypers); lp = eng.gp_logprob()
            ref = lp if ref is None else ref
            assert np.array_equal(lp, ref)
            t = time.time()
            for _ in range(20):
                eng.set_hypers(hypers); eng.gp_logprob()
            out.append((time.time() - t) / 20 * 1e3)
        print("N=%d H=%2d  eager %.3f ms  lazy %.3f ms  (%+.1f %%)" % (N, H, out[0], out[1], (out[1] / out[0] - 1) * 100))
eng.set_option("lean_lazy", -1)

"""Dev tool: spx_gp_logprob with a handle option off / on: bits and wall time per call.
   python scripts/dev/lean_option_ab.py lean_ps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
opt = sys.argv[1]
eng = Engine(0)
for N, D in ((2048, 32), (1000, 16), (512, 16), (256, 8), (100, 4), (4096, 32)):
    for H in (1, 2, 4, 6, 8, 12, 20, 32):
        if N == 4096 and H > 4:
            continue
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        if H >= 4:
            hypers[2, 2] = -1.0      # a non-PD draw in the batch
        eng.set_observations(comp, vals)
        res, tms = [], []
        for on in (0, 1):
            eng.set_option(opt, on)
            eng.set_hypers(hypers); res.append(eng.gp_logprob())
            t = time.time()
            for _ in range(20):
                eng.set_hypers(hypers); eng.gp_logprob()
            tms.append((time.time() - t) / 20 * 1e3)
        same = np.array_equal(res[0], res[1])
        print("N=%4d H=%2d  %s=0 %.3f ms  =1 %.3f ms  (%+.1f %%)  bit-identical %s%s"
              % (N, H, opt, tms[0], tms[1], (tms[1] / tms[0] - 1) * 100, same, "" if same else "   <-- FAIL  max diff %.3e" % np.nanmax(np.abs(res[0] - res[1]))))
eng.set_option(opt, -1)

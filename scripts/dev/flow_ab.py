"""Dev tool: k_lean_flow (option lean_flow=1) against the per-column launches: bits and wall time per call.
   python scripts/dev/flow_ab.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
quick = len(sys.argv) > 1
eng = Engine(0)
cases = ((512, 16), (2048, 32)) if quick else ((2048, 32), (1000, 16), (512, 16), (256, 8), (100, 4), (64, 4), (4096, 32))
for N, D in cases:
    for H in ((1, 4, 8) if quick else (1, 2, 4, 8, 12, 20, 32)):
        if N == 4096 and H > 4:
            continue
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        if H >= 4:
            hypers[2, 2] = -1.0      # a non-PD draw in the batch
        eng.set_observations(comp, vals)
        res, tms = [], []
        for on in (0, 1):
            eng.set_option("lean_flow", on)
            try:
                eng.set_hypers(hypers); res.append(eng.gp_logprob())
                t = time.time()
                for _ in range(20):
                    eng.set_hypers(hypers); eng.gp_logprob()
                tms.append((time.time() - t) / 20 * 1e3)
            except Exception as e:
                print("N=%d H=%d lean_flow=%d: %s" % (N, H, on, e)); res.append(None); tms.append(float("nan"))
        same = res[0] is not None and res[1] is not None and np.array_equal(res[0], res[1], equal_nan=True)
        print("N=%4d H=%2d  flow=0 %.3f ms  =1 %.3f ms  (%+.1f %%)  bit-identical %s%s"
              % (N, H, tms[0], tms[1], (tms[1] / tms[0] - 1) * 100, same,
                 "" if same or res[1] is None else "   <-- FAIL  max diff %.3e" % np.nanmax(np.abs(res[0] - res[1]))))
        if not same and res[1] is not None:
            print("   flow=0:", res[0][:6], "\n   flow=1:", res[1][:6])
        sys.stdout.flush()
eng.set_option("lean_flow", -1)

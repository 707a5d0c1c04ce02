"""Dev tool: k_lean_flow variants (option lean_flow = 1 | 2 plain history loads | 4 priority) against the per-column launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
modes = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 5]
eng = Engine(0)
for N, D in ((2048, 32), (1000, 16)):
    for H in (1, 2, 4, 8, 16):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals)
        out = []
        ref = None
        for m in modes:
            eng.set_option("lean_flow", m)
            eng.set_hypers(hypers); r = eng.gp_logprob()
            if ref is None: ref = r
            bad = 0
            t = time.time()
            for _ in range(30):
                eng.set_hypers(hypers); bad += not np.array_equal(eng.gp_logprob(), ref, equal_nan=True)
            out.append("m%d %.3f ms%s" % (m, (time.time() - t) / 30 * 1e3, "" if not bad else " (%d MISMATCHES)" % bad))
        print("N=%4d H=%2d  " % (N, H) + "  ".join(out)); sys.stdout.flush()

"""Dev tool: k_lean_flow residency modes -- by the size rule (default) | two workgroups per CU yielding to a neighbour's
diagonal block | two per CU, no yielding | one per CU -- wall per call, bits compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
MODES = (("rule", -1, -1), ("2/CU+yield", 0, 1), ("2/CU", 0, 0), ("1/CU", 1, -1))
eng = Engine(0)
for N, D in ((2048, 32), (1000, 16), (4096, 32)):
    for H in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 6, 8, 12, 20, 32)):
        if N == 4096 and H > 4:
            continue
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals)
        out, ref = [], None
        for name, cu, yl in MODES:
            eng.set_option("lean_flow_cu", cu); eng.set_option("lean_flow_yield", yl)
            eng.set_hypers(hypers); r = eng.gp_logprob()
            ref = r if ref is None else ref
            best, bad = 1e9, 0
            for rep in range(3):
                t = time.time()
                for _ in range(20):
                    eng.set_hypers(hypers); bad += not np.array_equal(eng.gp_logprob(), ref, equal_nan=True)
                best = min(best, (time.time() - t) / 20 * 1e3)
            out.append("%s %.3f%s" % (name, best, "" if not bad else " (%d MISMATCHES)" % bad))
        print("N=%4d H=%2d  " % (N, H) + "  ".join(out)); sys.stdout.flush()

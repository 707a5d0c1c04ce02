"""Dev: spx_ei_step with / without the producer work beside the factorisation (option step_overlap), interleaved, same box.
   python scripts/dev/step_overlap_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine, FLAG_PER_SEC
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for name, (N, M, D, H, ps) in (("c2", (256, 20000, 8, 10, False)), ("c5/5", (1024, 100000, 16, 20, True)), ("c3/10", (2048, 20000, 32, 20, False)),
                               ("n64", (64, 20000, 8, 10, False)), ("n128ps", (128, 20000, 8, 10, True)), ("n600", (600, 20000, 8, 10, False))):
    prob = synthetic_problem(N, M, D, H, 11, per_sec=ps)
    comp, cand, vals, hyp = prob[:4]
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp)
    if ps:
        eng.set_time_model(prob[4], prob[5])
    fl = FLAG_PER_SEC if ps else 0
    reps = 30 if N <= 600 else 5
    best = {0: 1e9, 1: 1e9}
    res = {}
    for rnd in range(4):
        for ov in (0, 1):
            eng.set_option("step_overlap", ov)
            eng.ei_step(fl)
            t = time.time()
            for _ in range(reps):
                eng.ei_step(fl)
            best[ov] = min(best[ov], (time.time() - t) / reps * 1e3)
            res[ov] = (eng.best(), eng.ei_mean())
    same = res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    print("%-7s N=%4d M=%6d H=%2d per_sec=%d | behind %.3f ms  beside %.3f ms  (%+.1f %%)  same bits: %s"
          % (name, N, M, H, ps, best[0], best[1], (best[1] / best[0] - 1) * 100, same), flush=True)
eng.set_option("step_overlap", -1)

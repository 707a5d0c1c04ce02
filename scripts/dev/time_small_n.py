"""Dev tool: one EI step (spx_factor + spx_ei_run, data resident) at small N, general three-stage path vs the fused kernel.
   python scripts/dev/time_small_n.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = Engine(0)
for (N, M, D, H) in ((20, 20000, 2, 10), (64, 20000, 8, 10), (128, 20000, 8, 10), (128, 200000, 8, 10), (128, 20000, 32, 10), (100, 1000, 4, 10)):
    comp, cand, vals, hyp = synthetic_problem(N, M, D, H, 11)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp)
    line = "N=%3d M=%6d D=%2d H=%2d |" % (N, M, D, H)
    res = []
    for fused in (0, 1):
        eng.set_option("ei_fused", fused)
        eng.factor(); eng.ei_run()
        t = time.time()
        for _ in range(reps):
            eng.factor(); eng.ei_run()
        two = (time.time() - t) / reps * 1e3
        eng.ei_step()
        t = time.time()
        for _ in range(reps):
            eng.ei_step()
        step = (time.time() - t) / reps * 1e3
        t = time.time()
        for _ in range(reps):
            eng.ei_run()
        run = (time.time() - t) / reps * 1e3
        eng.set_option("timing", 1)
        for _ in range(5):
            eng.factor(); eng.ei_run()
        tm = eng.timings(); eng.set_option("timing", 0)
        res.append((eng.best(), eng.ei_draws()))
        line += "  fused=%d step %.3f ms (two calls %.3f), ei_run alone %.3f ms [%s]" % (fused, step, two, run, " ".join(
            "%s %.0fus" % (k[:10], v[0] / 5 * 1e3) for k, v in tm.items() if v[1] and k in ("cov_cross", "predict_gemm", "ei_finalize", "scale_rows", "mean_argmax", "factor_total", "ei_run_total")))
    same = res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    print(line + "  bits equal: %s" % same, flush=True)
eng.set_option("ei_fused", -1)

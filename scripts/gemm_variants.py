"""A/B of predict-GEMM variants (dev tool, GPU box): bit comparison with the production kernel on a problem with and
without fantasies, then per-launch time at C3 from the library's own HIP events.
python scripts/gemm_variants.py 31 32 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem

variants = [int(v) for v in sys.argv[1:]] or [31, 32]
eng = Engine(0)
# ---- bits -----------------------------------------------------------------------------------
comp, cand, vals, hypers = synthetic_problem(700, 5000, 9, 3, 5)
ref = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)[3]
rs = np.random.RandomState(1)
fant = rs.randn(3, 700, 7); bests = fant.min(axis=1)


def with_fantasies():
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.set_fantasies(fant, bests); eng.ei_run()
    return eng.ei_draws()


ref_f = with_fantasies()
for v in variants:
    eng.set_option("gemm_waves", v)
    got = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)[3]
    got_f = with_fantasies()
    rel = lambda x, y: float(np.max(np.abs(x - y)[y > 1e-250] / y[y > 1e-250]))
    print("variant %d: identical bits %s (max rel diff %.1e), with fantasies %s (%.1e)"
          % (v, np.array_equal(got, ref), rel(got, ref), np.array_equal(got_f, ref_f), rel(got_f, ref_f)))
    eng.set_option("gemm_waves", 0)
# ---- time at C3 ------------------------------------------------------------------------------
comp, cand, vals, hypers = synthetic_problem(2048, 200000, 32, 20, 3000)
eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
for rep in range(2):
    for v in [0] + variants:
        eng.set_option("gemm_waves", v)
        eng.factor(); eng.ei_run()
        eng.set_option("timing", 1)
        eng.factor(); eng.ei_run()
        tm = eng.timings()
        eng.set_option("timing", 0)
        ms, n = tm["predict_gemm"]
        print("variant %2d: %.4f ms per launch (%d launches), best %s" % (v, ms / n, n, eng.best()))

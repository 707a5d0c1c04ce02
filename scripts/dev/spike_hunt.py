"""Dev: where do the occasional ~20 ms stalls of the first library call after an EI pass come from?"""
import os, sys, time, gc, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine, _dp
from spearmint_amd.synthetic import synthetic_problem
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "nogc":
    gc.disable()
eng = Engine(0)
N, M, D, H = 256, 20000, 8, 10
probs = [synthetic_problem(N, M, D, H, 100 + r) for r in range(6)]
ts, tm = [], []
for rep in range(120):
    comp, cand, vals, hypers = probs[rep % 6] if mode != "fresh" else synthetic_problem(N, M, D, H, 100 + rep)
    t = time.perf_counter(); eng.set_observations(comp, vals); ts.append(time.perf_counter() - t)
    for _ in range(10):
        eng.set_hypers(hypers); eng.gp_logprob()
    t = time.perf_counter()
    if mode == "nomean":
        eng.ei_grid(comp, vals, cand, hypers, want_mean=False)
    else:
        eng.ei_grid(comp, vals, cand, hypers, want_mean=True)
    tm.append(time.perf_counter() - t)
ts = np.array(ts[1:]) * 1e3; tm = np.array(tm[1:]) * 1e3
print("%-6s set_observations: median %.3f max %.2f  >1ms: %d of %d at %s | ei_grid: median %.3f max %.2f >3ms: %d" % (
    mode, np.median(ts), ts.max(), (ts > 1).sum(), len(ts), list(np.nonzero(ts > 1)[0] + 1), np.median(tm), tm.max(), (tm > 3).sum()))

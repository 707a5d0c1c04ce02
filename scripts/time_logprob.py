"""Host vs GPU log-likelihood per call (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd import hostgp
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in [(64, 4), (256, 8), (1024, 16), (2048, 32)]:
    comp, cand, vals, hypers = synthetic_problem(N, 16, D, 1, 5)
    h = hypers[0]
    eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
    t = time.time(); n = 20
    for _ in range(n):
        eng.set_hypers(hypers); g = eng.gp_logprob()[0]
    tg = (time.time() - t) / n
    t = time.time(); n2 = 3
    for _ in range(n2):
        c = hostgp.data_logprob(comp, vals, h[0], h[2], h[1], h[3:])
    tc = (time.time() - t) / n2
    print("N=%d D=%d  gpu %.2f ms  host %.2f ms  (%.1fx)  rel diff %.1e" % (N, D, tg * 1e3, tc * 1e3, tc / tg, abs(g - c) / abs(c)))

"""gp_logprob latency vs number of hyper draws in the batch (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
for N, D in ((2048, 32), (1024, 16), (256, 8)):
    row = []
    for H in (1, 4, 8, 12, 19, 27, 35, 48):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
        t = time.time()
        for _ in range(10):
            eng.set_hypers(hypers); eng.gp_logprob()
        row.append("H=%d %.2f" % (H, (time.time() - t) / 10 * 1e3))
    print("N=%d D=%d ms per call: " % (N, D) + "  ".join(row))

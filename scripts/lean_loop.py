"""Dev tool: N calls of spx_gp_logprob at one size, for rocprofv3 --kernel-trace --stats.  python scripts/lean_loop.py N D H calls"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
N, D, H, calls = [int(v) for v in sys.argv[1:5]]
eng = Engine(0)
comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
t = time.time()
for _ in range(calls):
    eng.set_hypers(hypers); eng.gp_logprob()
print("N=%d H=%d wall per call %.3f ms" % (N, H, (time.time() - t) / calls * 1e3))

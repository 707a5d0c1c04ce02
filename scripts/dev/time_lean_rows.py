"""Dev tool: spx_gp_logprob wall time per call by batch size (default modes).
python scripts/dev/time_lean_rows.py [N:D ...]   (default 2048:32 1024:16)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2048, 32), (1024, 16)]
for N, D in shapes:
    row = []
    for H in (1, 2, 4, 6, 8, 12, 16, 20, 24, 32, 48):
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5)
        eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.gp_logprob()
        reps = 200 if N <= 512 else 30
        t = time.perf_counter()
        for _ in range(reps):
            eng.set_hypers(hypers); eng.gp_logprob()
        row.append("H=%d %.3f" % (H, (time.perf_counter() - t) / reps * 1e3))
    print("N=%d D=%d  " % (N, D) + "  ".join(row), flush=True)

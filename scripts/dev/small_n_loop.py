"""Dev: a loop of EI steps (spx_factor + spx_ei_run, data resident) at a small size, for rocprofv3.
   python scripts/dev/small_n_loop.py N M D H reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
N, M, D, H, reps = [int(x) for x in sys.argv[1:6]]
eng = Engine(0)
comp, cand, vals, hyp = synthetic_problem(N, M, D, H, 11)
eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hyp)
for _ in range(reps):
    eng.factor(); eng.ei_run()
print(eng.best())

"""Dev: time of spx_ei_grad_batch at C3 size (20 draws, 20 points) + bits vs the default library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
comp, cand, vals, hypers = synthetic_problem(2048, 2000, 32, 20, 9)
eng = Engine(0)
eng.ei_grid(comp, vals, cand, hypers)
for P in (1, 8, 20):
    f, g = eng.ei_grad_batch(cand[:P])
    t = time.time()
    for _ in range(10):
        eng.ei_grad_batch(cand[:P])
    print("P=%2d  %.3f ms per call   checksum %.17g %.17g" % (P, (time.time() - t) / 10 * 1e3, f.sum(), np.abs(g).sum()))

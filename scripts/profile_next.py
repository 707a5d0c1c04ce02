"""cProfile of one GPEIOptChooser.next() (dev tool): python scripts/profile_next.py N M D"""
import sys, os, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
N, M, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
durations = np.ones(N + M)
complete = np.arange(N); candidates = np.arange(N, N + M); pending = np.array([], dtype=int)
ch = GPEIOptChooser.init(tempfile.mkdtemp(), "burnin=2,use_multiprocessing=0," + (sys.argv[5] if len(sys.argv) > 5 else "mcmc_iters=4,grid_subset=4") + (("," + sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else ""))
npr.seed(3)
ch.engine().set_observations(comp, vals)   # GPU / library warm-up outside the profile
pr = cProfile.Profile()
t = time.time(); pr.enable()
job = ch.next(grid, values, durations, candidates, pending, complete)
pr.disable(); print("next() %.2f s" % (time.time() - t))
pstats.Stats(pr).sort_stats(os.environ.get("SPX_PROF_SORT", "cumulative")).print_stats(22)

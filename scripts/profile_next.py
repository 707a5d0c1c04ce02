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
ARGS = "burnin=2,use_multiprocessing=0," + (sys.argv[5] if len(sys.argv) > 5 else "mcmc_iters=4,grid_subset=4") + (("," + sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else "")
# a whole next() first (library load, HIP context, buffer allocation, the sampler's depth calibration): the profile below is a
# WARM call of a second chooser object that reuses the first one's engine -- what a long-lived driver (main.py) sees from its
# second proposal on
warm = GPEIOptChooser.init(tempfile.mkdtemp(), ARGS)
npr.seed(3)
warm.next(grid, values, durations, candidates, pending, complete)
ch = GPEIOptChooser.init(tempfile.mkdtemp(), ARGS)
ch._eng = warm._eng
ch._depth_cache = warm._depth_cache
npr.seed(3)
pr = cProfile.Profile()
t = time.time(); pr.enable()
job = ch.next(grid, values, durations, candidates, pending, complete)
pr.disable(); print("next() %.4f s (warm)   sampler %s" % (time.time() - t, {k: v for k, v in ch.sampler_stats.items() if k != "calls_by_rows"}))
pstats.Stats(pr).sort_stats(os.environ.get("SPX_PROF_SORT", "cumulative")).print_stats(22)

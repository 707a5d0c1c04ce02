"""Dev: rows per spx_gp_logprob call during one GPEIOptChooser.next() (GPU box)."""
import os, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
N, M, D = 2048, 200000, 32
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
ch = GPEIOptChooser.init(tempfile.mkdtemp(), "burnin=2,use_multiprocessing=0,mcmc_iters=20,grid_subset=20" + (("," + sys.argv[1]) if len(sys.argv) > 1 else ""))
npr.seed(3)
eng = ch.engine()
eng.set_observations(comp, vals)
hist = collections.Counter(); tms = collections.defaultdict(float)
orig = eng.gp_logprob
def wrapped(*a, **k):
    n = eng.H
    t = time.perf_counter(); r = orig(*a, **k); tms[n] += time.perf_counter() - t
    hist[n] += 1
    return r
eng.gp_logprob = wrapped
t = time.time()
ch.next(grid, values, np.ones(N + M), np.arange(N, N + M), np.array([], dtype=int), np.arange(N))
print("next() %.2f s" % (time.time() - t))
for n in sorted(hist):
    print("rows %2d: %4d calls, %.3f s, %.2f ms per call" % (n, hist[n], tms[n], 1e3 * tms[n] / hist[n]))
print("rows evaluated", sum(n * c for n, c in hist.items()), "calls", sum(hist.values()))

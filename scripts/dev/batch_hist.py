"""Dev: rows per spx_gp_logprob call during one warm GPEIOptChooser.next() (GPU box).
python scripts/dev/batch_hist.py [N M D] [chooser args, e.g. mcmc_iters=10,burnin=10,grid_subset=20,lookahead=12]"""
import os, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
from spearmint_amd import engine as E
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
A = [a for a in sys.argv[1:] if "=" not in a]
N, M, D = (int(A[0]), int(A[1]), int(A[2])) if len(A) >= 3 else (2048, 200000, 32)
EXTRA = [a for a in sys.argv[1:] if "=" in a]
ARGS = "use_multiprocessing=0," + (EXTRA[0] if EXTRA else "burnin=2,mcmc_iters=20,grid_subset=20")
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
hist = collections.Counter(); tms = collections.defaultdict(float)
orig = E.Engine.gp_logprob
def wrapped(self, *a, **k):
    n = self.H
    t = time.perf_counter(); r = orig(self, *a, **k); tms[n] += time.perf_counter() - t
    hist[n] += 1
    return r
E.Engine.gp_logprob = wrapped
for rep in range(3):
    ch = GPEIOptChooser.init(tempfile.mkdtemp(), ARGS)
    npr.seed(3)
    hist.clear(); tms.clear()
    t = time.perf_counter()
    job = ch.next(grid, values, np.ones(N + M), np.arange(N, N + M), np.array([], dtype=int), np.arange(N))
    wall = time.perf_counter() - t
    print("next() %.4f s -> %s" % (wall, str(job)[:70].replace("\n", " ")))
print("N=%d M=%d D=%d %s" % (N, M, D, ARGS))
for n in sorted(hist):
    print("rows %2d: %4d calls, %.4f s, %.3f ms per call" % (n, hist[n], tms[n], 1e3 * tms[n] / hist[n]))
print("rows evaluated", sum(n * c for n, c in hist.items()), "calls", sum(hist.values()), "time in calls %.4f s of %.4f" % (sum(tms.values()), wall))

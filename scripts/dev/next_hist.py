"""Dev tool: one GPEIOptChooser.next() at C3 size -- wall time (no profiler) and the batch sizes of its log-likelihood calls."""
import sys, os, time, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
from spearmint_amd import engine as E
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
N, M, D = 2048, 200000, 32
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
durations = np.ones(N + M)
complete = np.arange(N); candidates = np.arange(N, N + M); pending = np.array([], dtype=int)
hist = collections.Counter(); tsum = collections.Counter()
orig_set = E.Engine.set_hypers; orig_lp = E.Engine.gp_logprob
last = [0]
def set_hypers(self, h, *a, **k):
    last[0] = len(h); return orig_set(self, h, *a, **k)
def gp_logprob(self, *a, **k):
    t = time.time(); r = orig_lp(self, *a, **k); tsum[last[0]] += time.time() - t; hist[last[0]] += 1; return r
extra = ("," + sys.argv[1]) if len(sys.argv) > 1 else ""
for rep in range(2):
    ch = GPEIOptChooser.init(tempfile.mkdtemp(), "burnin=2,use_multiprocessing=0,mcmc_iters=20,grid_subset=20" + extra)
    npr.seed(3)
    ch.engine().set_observations(comp, vals)
    if rep == 1:
        E.Engine.set_hypers = set_hypers; E.Engine.gp_logprob = gp_logprob
    t = time.time()
    job = ch.next(grid, values, durations, candidates, pending, complete)
    print("next() %.3f s  (job %s)" % (time.time() - t, str(job)[:60]))
print("log-likelihood calls by batch size: " + "  ".join("%d: %d (%.0f ms)" % (k, hist[k], tsum[k] * 1e3) for k in sorted(hist)))
print(extra, "total %d calls, %.3f s" % (sum(hist.values()), sum(tsum.values())))

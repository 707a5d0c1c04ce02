"""Dev tool (GPU): per-move statistics of the speculative slice sampler during one GPEIOptChooser.next()."""
import sys, os, time, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd import util
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
N, M, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
durations = np.ones(N + M)
complete = np.arange(N); candidates = np.arange(N, N + M); pending = np.array([], dtype=int)
ch = GPEIOptChooser.init(tempfile.mkdtemp(), "burnin=2,use_multiprocessing=0,mcmc_iters=20,grid_subset=20" + (("," + sys.argv[4]) if len(sys.argv) > 4 else ""))
npr.seed(3)
ch.engine().set_observations(comp, vals)
calls = []
orig = ch.data_logprob_many
def counted(c, v, rows):
    calls.append(len(rows)); return orig(c, v, rows)
ch.data_logprob_many = counted
moves = []
orig_along = util._slice_along_batched
def along(*a, **k):
    n0 = len(calls); r = orig_along(*a, **k); moves.append(tuple(calls[n0:])); return r
util._slice_along_batched = along
t = time.time(); ch.next(grid, values, durations, candidates, pending, complete); print("next() %.2f s" % (time.time() - t))
print("moves %d  gpu calls %d  rows %d  (%.2f calls/move, %.2f rows/call)" % (len(moves), len(calls), sum(calls), len(calls) / len(moves), sum(calls) / len(calls)))
hist = collections.Counter(len(m) for m in moves)
print("calls per move:", dict(sorted(hist.items())))
first = collections.Counter(m[0] for m in moves if m)
print("rows in the first call of a move:", dict(sorted(first.items())))

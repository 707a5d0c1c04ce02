"""Dev tool (GPU): what the slice sampler's moves look like at a given size -- how far each side steps out and why it
stops (slice level or the priors' support), which shrink proposal is accepted -- to plan the speculative batches."""
import sys, os, time, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd import util
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None
N, D = int(sys.argv[1]), int(sys.argv[2])
comp, cand, vals, _ = synthetic_problem(N, 100, D, 1, 9)
ch = GPEIOptChooser.init(tempfile.mkdtemp(), "burnin=2,use_multiprocessing=0,mcmc_iters=6,grid_subset=2")
ch._real_init(D, vals)
npr.seed(3)
moves = collections.Counter(); acc = collections.Counter()
def along(direction, x0, logprob_many, sigma, step_out, max_steps_out, lookahead):
    adm = logprob_many.admissible if hasattr(logprob_many, "admissible") else None
    def f(z):
        return logprob_many([direction * z + x0]).get(0)
    hi = sigma * npr.rand(); lo = hi - sigma
    level = np.log(npr.rand()) + f(0.0)
    def side(z, step):
        n = 0
        while True:
            x = direction * z + x0
            a = ch_adm(x)
            v = f(z)
            if not (v > level and n < max_steps_out):
                return z, n, ("prior" if not a else "level")
            n += 1; z = z + step
    lo, nl, wl = side(lo, -sigma); hi, nh, wh = side(hi, sigma)
    moves[(nl, wl, nh, wh)] += 1
    k = 0
    while True:
        z = (hi - lo) * npr.rand() + lo; lp = f(z); k += 1
        if lp > level:
            acc[k] += 1
            return z * direction + x0
        if z < 0: lo = z
        elif z > 0: hi = z
        else: raise Exception("zero")
util._slice_along_batched = along
# admissibility of the CURRENT move's closure: patched in through _speculative_logprob
orig_spec = ch._speculative_logprob
state = {}
def spec(c, v, to_row, finish):
    m = orig_spec(c, v, to_row, finish)
    state["adm"] = lambda x: to_row(x) is not None
    return m
ch._speculative_logprob = spec
def ch_adm(x): return state["adm"](x)
t = time.time()
for it in range(4):
    ch.sample_hypers(comp, vals)
print("N=%d D=%d: %.1f s" % (N, D, time.time() - t))
tot = sum(moves.values())
for k, v in sorted(moves.items(), key=lambda kv: -kv[1])[:14]:
    print("  lo: %d steps, stop by %-5s | hi: %d steps, stop by %-5s : %5.1f %%" % (k[0], k[1], k[2], k[3], 100.0 * v / tot))
ta = sum(acc.values())
print("  accepted at proposal:", {k: round(v / ta, 3) for k, v in sorted(acc.items())})

"""Dev: a warm next() of the three choosers at one size (phases by perf_counter)."""
import os, sys, tempfile, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
import spearmint_amd.chooser._base as b
b.log = lambda *a: None
N, M, D = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 20000, 8)
rs = np.random.RandomState(1)
grid = rs.rand(N + M, D)
values = np.full(N + M, np.nan); values[:N] = np.sin(3 * grid[:N]).sum(axis=1) + 0.05 * rs.randn(N)
durations = np.full(N + M, np.nan); durations[:N] = 1.0 + 3.0 * grid[:N, 0] + np.sin(5 * grid[:N, 1]) ** 2
complete, candidates, pending = np.arange(N), np.arange(N, N + M), np.array([], dtype=int)
for name in ("GPEIChooser", "GPEIOptChooser", "GPEIperSecChooser"):
    mod = importlib.import_module("spearmint_amd.chooser." + name); mod.log = lambda *a: None
    args = "mcmc_iters=10" + ("" if name == "GPEIChooser" else ",burnin=10,grid_subset=20,use_multiprocessing=0")
    for extra in ("sampler=python,lookahead=6,follow=0:0", "sampler=native"):
        ts = []
        for rep in range(3):
            ch = mod.init(tempfile.mkdtemp(), args + "," + extra)
            npr.seed(3)
            t = time.perf_counter(); job = ch.next(grid, values, durations, candidates, pending, complete); ts.append(time.perf_counter() - t)
        print("%-18s %-40s next() %.4f s (best of the last two)  sampler %s" % (name, extra, min(ts[1:]), {k: v for k, v in ch.sampler_stats.items() if k in ("calls", "moves", "free_moves")}), flush=True)

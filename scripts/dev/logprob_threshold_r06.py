"""Dev: a warm GPEIOptChooser.next() at tiny N -- the reference's serial sampler on the host (gpu_logprob=0) vs the native sampler
on the GPU (gpu_logprob=1): where should "auto" switch?   python scripts/dev/logprob_threshold_r06.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser as mod
import spearmint_amd.chooser._base as b
b.log = mod.log = lambda *a: None
M, D = 2000, 4
rs = np.random.RandomState(2)
for N in (2, 3, 4, 6, 8, 12, 16, 24, 32):
    grid = rs.rand(N + M, D)
    values = np.full(N + M, np.nan); values[:N] = np.sin(3 * grid[:N]).sum(axis=1) + 0.05 * rs.randn(N)
    out = []
    for g in (0, 1):
        ts = []; job = None
        for rep in range(3):
            ch = mod.init(tempfile.mkdtemp(), "mcmc_iters=10,burnin=10,grid_subset=20,use_multiprocessing=0,gpu_logprob=%d" % g)
            npr.seed(4)
            t = time.perf_counter(); job = ch.next(grid, values, np.ones(N + M), np.arange(N, N + M), np.array([], dtype=int), np.arange(N)); ts.append(time.perf_counter() - t)
        out.append((min(ts[1:]), job))
    same = (out[0][1][0] == out[1][1][0]) and np.allclose(out[0][1][1], out[1][1][1], atol=1e-6)
    print("N=%2d  host sampler %.4f s   GPU native sampler %.4f s   same proposal %s" % (N, out[0][0], out[1][0], same), flush=True)

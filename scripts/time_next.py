"""End-to-end GPEIOptChooser.next() wall time: host sampler/refinement vs GPU ones (dev tool);
`python scripts/time_next.py --with-reference`: the reference's own next() beside ours (bench.next_baseline)."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--with-reference" in sys.argv:
    # the reference's own GPEIOptChooser.next() beside ours, same seeded state, N = 256 x 20 000 x mcmc_iters 10:
    # bench.py's cpu_baseline leg end to end (the reference loader lives behind bench.py, not in the product)
    import json
    import bench
    print(json.dumps(bench.next_baseline(), indent=1))
    sys.exit(0)
import numpy as np, numpy.random as npr
from spearmint_amd.chooser import GPEIOptChooser
from spearmint_amd.synthetic import synthetic_problem
from spearmint_amd import helpers
helpers.log = lambda *a: None
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log = o.log = lambda *a: None

N, M, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comp, cand, vals, _ = synthetic_problem(N, M, D, 1, 9)
grid = np.vstack((comp, cand)); values = np.concatenate((vals, np.full(M, np.nan)))
durations = np.ones(N + M)
complete = np.arange(N); candidates = np.arange(N, N + M); pending = np.array([], dtype=int)
LA = os.environ.get("SPX_LOOKAHEAD", "4")
for label, extra in (("gpu sampler+refine", "gpu_logprob=1,gpu_refine=1,lookahead=" + LA), ("host sampler+refine", "gpu_logprob=0,gpu_refine=0")):
    if label.startswith("host") and len(sys.argv) > 4 and sys.argv[4] == "nohost":
        continue
    ch = GPEIOptChooser.init(tempfile.mkdtemp(), "mcmc_iters=4,burnin=2,grid_subset=4,use_multiprocessing=0," + extra)
    npr.seed(3); np.random.seed(3)
    t = time.time()
    job = ch.next(grid, values, durations, candidates, pending, complete)
    print("%-22s N=%d M=%d D=%d: next() %.1f s -> %s" % (label, N, M, D, time.time() - t,
          job if not isinstance(job, tuple) else (job[0], np.round(job[1][:3], 4))))

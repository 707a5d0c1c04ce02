"""Dev: which call leaves device memory behind after spx_destroy (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spearmint_amd.engine import Engine
from spearmint_amd import sobol
from spearmint_amd.synthetic import synthetic_problem
comp, cand, vals, hypers, ld, th = synthetic_problem(300, 20000, 6, 3, 77, per_sec=True)
rs = np.random.RandomState(0)
ops = {
    "create only": lambda e: None,
    "set_observations": lambda e: e.set_observations(comp, vals),
    "ei_grid": lambda e: e.ei_grid(comp, vals, cand, hypers, want_draws=True),
    "per_sec": lambda e: e.ei_per_sec_grid(comp, vals, ld, cand, hypers, th),
    "grad": lambda e: (e.ei_grid(comp, vals, cand, hypers), e.ei_grad_batch(cand[:5])),
    "logprob": lambda e: (e.set_observations(comp, vals), e.set_hypers(hypers), e.gp_logprob()),
    "sobol": lambda e: e.sobol_grid(sobol.load_dirs("bf40"), 8, 50000, 1),
}
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info(0)[0]
e = Engine(0); e.ei_grid(comp, vals, cand, hypers); e.close()
for name, op in ops.items():
    f0 = free()
    for _ in range(4):
        e = Engine(0); op(e); e.close()
    print("%-18s %.1f MiB per lifetime" % (name, (f0 - free()) / 4 / 2.0 ** 20))

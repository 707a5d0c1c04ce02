"""Sobol grid generator: kernel time and write bandwidth (dev tool; `python scripts/time_sobol.py`)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd import sobol

eng = Engine(0)
out = []
for table, m, n in [("jk1111", 32, 1 << 20), ("jk1111", 32, 200000), ("jk1111", 8, 20000), ("bf40", 16, 500000),
                    ("jk1111", 32, 1 << 23), ("jk1111", 1000, 100000)]:
    V = sobol.load_dirs(table)
    eng.sobol_grid(V, m, n, 1, fetch=False)
    ms = min(eng.sobol_grid(V, m, n, 1, fetch=False)[1] for _ in range(5))
    t = time.time(); g, _ = eng.sobol_grid(V, m, n, 1); wall = time.time() - t
    out.append({"table": table, "dim": m, "n": n, "kernel_ms": round(ms, 4), "GB_per_s": round(n * m * 8 / ms / 1e6, 1),
                "with_copy_to_host_ms": round(wall * 1e3, 2)})
    print(out[-1])
json.dump({"kernel": "k_sobol_grid", "bound": "hbm (8 B written per element)", "peak_GB_per_s": 8000, "runs": out},
          open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "sobol_timing.json"), "w"), indent=1)

"""Dev tool (GPU): stress the in-launch hand-offs of k_lean_flow (default) and k_lean_step_ps -- random sizes and batch
sizes, repeated, with another engine keeping the GPU busy from a second host thread -- every result compared bit for bit
with the two-launch path.   python scripts/dev/ps_stress.py [configurations] [flow|ps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
form = sys.argv[2] if len(sys.argv) > 2 else "flow"
rs = np.random.RandomState(7)
eng = Engine(0)
stop = []
def noise():
    e2 = Engine(0)
    comp, cand, vals, hyp = synthetic_problem(1024, 30000, 16, 6, 99)
    while not stop:
        e2.ei_grid(comp, vals, cand, hyp, want_mean=False)
    e2.close()
th = threading.Thread(target=noise); th.start()
bad = 0; calls = 0; t0 = time.time()
for c in range(ncfg):
    N = int(rs.choice([40, 100, 260, 330, 512, 700, 1000, 1100, 1500, 2048, 2100, 3000]))
    H = int(rs.randint(1, 33 if N < 3000 else 6))
    D = int(rs.choice([2, 5, 16, 32]))
    comp, cand, vals, hyp = synthetic_problem(N, 10, D, H, int(rs.randint(1 << 30)))
    if rs.rand() < 0.2:
        hyp[rs.randint(H), 2] = -1.0
    eng.set_observations(comp, vals)
    eng.set_option("lean_flow", 0); eng.set_option("lean_ps", 0)
    eng.set_hypers(hyp); ref = eng.gp_logprob()
    eng.set_option("lean_flow", 1 if form == "flow" else 0); eng.set_option("lean_ps", 1)
    eng.set_option("lean_flow_cu", int(rs.randint(-1, 2)))
    eng.set_option("lean_flow_cov", int(rs.randint(0, 2)))
    eng.set_option("lean_flow_yield", int(rs.randint(0, 2)))
    for rep in range(6):
        eng.set_hypers(hyp); got = eng.gp_logprob(); calls += 1
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1
            print("MISMATCH N=%d H=%d rep=%d max diff %.3e" % (N, H, rep, np.nanmax(np.abs(got - ref))))
stop.append(1); th.join()
print("form %s: %d configurations, %d hand-off calls, %d mismatches, %.1f s" % (form, ncfg, calls, bad, time.time() - t0))
sys.exit(1 if bad else 0)

"""Dev: fresh engines, the chooser's first calls (depth calibration: 1 row x 3, 17 rows x 3, then the native sampler) -- does the
first sample_hypers of a fresh handle always give the same rows?   python scripts/dev/fresh_engine_stress.py [n=150]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, numpy.random as npr
from spearmint_amd.engine import Engine, SamplerCfg, RngState
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N, D = 40, 3
rs = np.random.RandomState(N)
comp = rs.rand(N, D); vals = np.sin(3 * comp).sum(axis=1) + 0.05 * rs.randn(N)
row = np.concatenate(([vals.mean(), 1e-3, np.std(vals) + 1e-4], np.ones(D)))
want = None
bad = 0; fallbacks = 0
for t in range(n):
    eng = Engine(0)
    eng.set_observations(comp, vals)
    lp1 = lp17 = None
    for k in (1, 17):
        for _ in range(3):
            eng.set_hypers(np.repeat(row[None, :], k, axis=0)); lp = eng.gp_logprob()
        if k == 1: lp1 = lp
        else: lp17 = lp
    ok_cal = np.all(lp17 == lp1[0])
    cfg = SamplerCfg(D=D, n_iter=3, noiseless=0, check_mean=1, amp2_prior_on_sqrt=1, lookahead=8, follow_props=4, follow_hyps=2,
                     max_rows=32, noise_scale=0.1, amp2_scale=1.0, max_ls=2.0, vals_min=float(vals.min()), vals_max=float(vals.max()))
    npr.seed(5)
    try:
        rows, st = eng.sample_hypers(cfg, row.copy(), np.zeros(12))
        res = rows
    except Exception as ex:
        res = "ERR %s" % type(ex).__name__
    fb = eng.stat("flow_fallbacks"); fallbacks += fb
    if want is None and not isinstance(res, str):
        want = res
    same = (not isinstance(res, str)) and np.array_equal(res, want)
    if not same or not ok_cal or fb:
        bad += 1
        print("trial %d: %s  calibration rows equal: %s  fallbacks %d" % (t, "DIFFERENT" if not isinstance(res, str) else res, ok_cal, fb), flush=True)
    eng.close()
print("%d of %d fresh engines misbehaved; %d fallbacks" % (bad, n, fallbacks))

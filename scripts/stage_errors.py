"""Dev tool: measured per-stage differences GPU vs oracle at N=2048 (SURVEY 8(d) stage criteria),
with sampled noise and with the noiseless setting.  python scripts/stage_errors.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_ei_oracle as orc
from spearmint_amd.engine import Engine, FLAG_KEEP_MOMENTS
from spearmint_amd.synthetic import synthetic_problem

eng = Engine(0)
out = []
for N, D, noise in ((2048, 32, None), (2048, 32, 1e-3), (1024, 8, 1e-3)):
    comp, cand, vals, hypers = synthetic_problem(N, 3000, D, 2, 3100)
    if noise is not None:
        hypers[:, 1] = noise
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.ei_run(FLAG_KEEP_MOMENTS)
    draws = eng.ei_draws()
    for h in range(2):
        st = {}
        ref = orc.compute_ei(comp, cand, vals, hypers[h], stages=st)
        K, L, alpha = eng.get_factor(h)
        m, v = eng.get_moments(h)
        ok = ref > 1e-280
        out.append(dict(N=N, D=D, noise=float(hypers[h, 1]), amp2=float(hypers[h, 2]),
                        K_rel=float(np.max(np.abs(K - st["K"]) / np.abs(st["K"]))),
                        LLt=float(np.linalg.norm(L @ L.T - st["K"]) / np.linalg.norm(st["K"])),
                        alpha_rel=float(np.max(np.abs(alpha - st["alpha"])) / np.abs(st["alpha"]).max()),
                        m_abs=float(np.max(np.abs(m - st["func_m"]))),
                        v_abs_over_amp2=float(np.max(np.abs(v - st["func_v"])) / hypers[h, 2]),
                        v_rel=float(np.max(np.abs(v - st["func_v"]) / np.abs(st["func_v"]))),
                        v_min=float(st["func_v"].min()),
                        ei_rel=float(np.max(np.abs(draws[ok, h] - ref[ok]) / ref[ok]))))
        print(json.dumps(out[-1]))

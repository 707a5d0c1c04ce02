"""Dev: where a pending-experiment EI pass (chooser._ei_with_pending_gpu) spends its time.   python scripts/dev/time_pending.py [N M D H P]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd import hostgp
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
a = [int(x) for x in sys.argv[1:]]
eng = Engine(0)
for (N, M, D, H, P) in ([tuple(a)] if len(a) == 5 else [(2048, 200000, 32, 20, 4), (300, 20000, 6, 10, 3), (60, 20000, 4, 10, 2)]):
    comp, cand, vals, hyp = synthetic_problem(N, M, D, H, 31)
    rs = np.random.RandomState(2)
    pend = rs.rand(P, D); S = 100
    randn = [rs.randn(P, S) for _ in range(H)]
    for rep in range(2):
        t = {}
        t0 = time.time()
        comp_pend = np.concatenate((comp, pend))
        eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(P)))); eng.set_candidates(cand); eng.set_hypers(hyp)
        eng.factor(); t["upload+factor"] = time.time() - t0
        fant = np.empty((H, N + P, S)); bests = np.empty((H, S))
        tg = tf = 0.0
        for h in range(H):
            t1 = time.time(); l_rows, gam = eng.get_factor_rows(h, N, P); tg += time.time() - t1
            t1 = time.time(); fant[h], bests[h] = hostgp.fantasize_from_factor_rows(vals, hyp[h], l_rows, gam, randn[h]); tf += time.time() - t1
        t["get_factor_rows x H"] = tg; t["host fantasize x H"] = tf
        t1 = time.time(); eng.set_fantasies(fant, bests); t["set_fantasies"] = time.time() - t1
        t1 = time.time(); eng.ei_run(); eng.best(); m = eng.ei_mean(); t["ei_run + mean"] = time.time() - t1
        t["total"] = time.time() - t0
    print("N=%d M=%d D=%d H=%d P=%d S=%d: " % (N, M, D, H, P, S) + "  ".join("%s %.1f ms" % (k, v * 1e3) for k, v in t.items()), flush=True)

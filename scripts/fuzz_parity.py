"""Randomised parity sweep: GPU engine vs the CPU oracle on random shapes (dev tool, GPU box).
python scripts/fuzz_parity.py [n_cases] [seed] [covar: Matern52 | Matern32 | ARDSE | SE | mix]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_ei_oracle as orc
from oracle import sobol_oracle as so
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
from spearmint_amd import sobol

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
covar_arg = sys.argv[3] if len(sys.argv) > 3 else "Matern52"
eng = Engine(0)
worst = 0.0
t0 = time.time()
for c in range(ncases):
    kname = str(rs.choice(["Matern52", "Matern32", "ARDSE", "SE"])) if covar_arg == "mix" else covar_arg
    eng.set_covar(kname)
    ctx = orc.covar(kname); ctx.__enter__()
    N = int(rs.choice([2, 3, 17, 63, 64, 65, 127, 128, 129, 200, 255, 257, 383, 511, 700, 1025, 1500, 2049]))
    M = int(rs.choice([10, 11, 63, 64, 65, 127, 129, 500, 1000, 4097, 20001]))
    D = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 40]))
    H = int(rs.choice([1, 2, 3, 5, 8]))
    per_sec = bool(rs.rand() < 0.3)
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, int(rs.randint(1 << 30)))
    if per_sec:
        durs = np.exp(0.5 * np.sin(comp.sum(axis=1)))
        thyp = hypers.copy(); thyp[:, 0] = 0.1; thyp[:, 3:] = rs.uniform(0.3, 5.0, (H, D))
        best, val, mean, draws = eng.ei_per_sec_grid(comp, vals, np.log(durs), cand, hypers, thyp, want_draws=True)
        ref = np.stack([orc.compute_ei_per_s(comp, cand, vals, np.log(durs), hypers[h], thyp[h]) for h in range(H)], axis=1)
    else:
        best, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        ref = np.stack([orc.compute_ei(comp, cand, vals, hypers[h]) for h in range(H)], axis=1)
    big = ref >= 1e-250
    err = float(np.max(np.abs(draws[big] - ref[big]) / ref[big])) if big.any() else 0.0
    ok_arg = best == int(np.argmax(np.mean(ref, axis=1))) or np.isclose(np.mean(ref, axis=1)[best], np.max(np.mean(ref, axis=1)), rtol=1e-9)
    eng.set_observations(comp, vals); eng.set_hypers(hypers)
    lp = eng.gp_logprob()
    lref = np.array([orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:]) for h in range(H)])
    # the two terms of the log-likelihood (-sum log diag L and the quadratic form) are each O(N) and may cancel:
    # the error is measured against their size, not against the cancelled sum (in the one case of the round-2
    # sweeps where that mattered -- ARDSE, N=1025, lp = 6.47 -- LAPACK itself is 1.6e-9 off an 80-bit evaluation)
    lerr = float(np.max(np.abs(lp - lref) / np.maximum(np.abs(lref), float(N))))
    worst = max(worst, err)
    # north-star tolerance 1e-5; the GPU tests assert 1e-7 on well-conditioned problems.  Errors grow with
    # eps * cond(K): D = 1 with >1000 observations on a line reaches 6e-7 (W = L^-1 vs LAPACK substitution).
    flag = "" if (err < 1e-5 and ok_arg and lerr < 1e-9) else "   <-- FAIL"
    if not flag and err >= 1e-7:
        flag = "   (ill-conditioned: > 1e-7)"
    ctx.__exit__()
    print("case %2d %-8s N=%4d M=%5d D=%2d H=%d per_sec=%d  ei rel err %.2e  logprob rel err %.1e  argmax %s%s"
          % (c, kname, N, M, D, H, per_sec, err, lerr, ok_arg, flag))
# Sobol: random (dim, n, skip)
for c in range(10):
    table = "bf40" if rs.rand() < 0.5 else "jk1111"
    V = sobol.load_dirs(table)
    m = int(rs.randint(1, V.shape[0] + 1)); n = int(rs.randint(1, 20000)); skip = int(rs.randint(-3, 1 << 20))
    g, _ = eng.sobol_grid(V, m, n, skip)
    same = np.array_equal(g.T, so.i4_sobol_generate(m, n, skip, V))
    print("sobol %s m=%d n=%d skip=%d bit-identical %s%s" % (table, m, n, skip, same, "" if same else "   <-- FAIL"))
print("worst EI rel err %.2e, %.1f s" % (worst, time.time() - t0))

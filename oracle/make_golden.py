#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz by RUNNING THE REFERENCE.

Runs the lib2to3-converted reference (see oracle/ref_py3.py; build container
only -- needs /root/reference) on seeded inputs and stores inputs + outputs.
Only numbers are written; no reference source enters the repo.

    python oracle/make_golden.py            # (re)writes tests/golden/*.npz

Fixtures (all float64, ref = the reference's own functions):
  ei_small_*.npz   GPEIChooser.compute_ei / GPEIOptChooser.ei_over_hypers on
                   seeded synthetic problems (several N, M, D, H), with the
                   per-stage arrays the reference computes internally
                   recomputed through its gp.Matern52 / cov.
  ei_pending.npz   GPEIChooser.compute_ei pending branch (fixed randn matrix,
                   obtained by seeding numpy.random right before the call).
  ei_persec.npz    GPEIperSecChooser.compute_ei_per_s (+ the literal
                   ei_over_hypers with its early return).
  branin_c1.npz    BASELINE config 1: the reference Sobol grid
                   (sobol_lib.i4_sobol_generate(2,1000,1).T), Branin values on
                   the first 20 points, a seeded GPEIChooser.next() call: the
                   slice-sampled hypers of each draw, overall_ei, chosen index.
  chooser_next.npz the reference's GPEIOptChooser.next / GPEIperSecChooser.next on Branin
                   (seeded; burnin + MCMC + refinement), the point they propose.
  chooser_next_pending.npz  the same with three pending jobs (fantasy branch).
  slice_sampler.npz  util.slice_sample traces under a seeded RNG.
  chooser_next_noiseless.npz  seeded next() of the three choosers with noiseless=1.
  branin_trajectory.npz  whole optimisation runs (24 / 14 / 12 proposals) of the three choosers on Branin.
  covar_{Matern32,ARDSE,SE}.npz  the other covar= choices: K, K*, EI (plain and with pending jobs), the
                         refinement objective and seeded next() calls of the three choosers.
  chooser_two_calls.npz  next(), restart from the state pickle with a new chooser object, next() again.
  chooser_next_ml2.npz  GPEIChooser.next with mcmc_iters=0 (ML-II hypers, gp.py:181-292).
  main_loop.npz    BASELINE config 1 through the reference's OWN primary driver: main.py's main() loop (GPEIChooser,
                   examples/braninpy, --grid-size=1000 --grid-seed=1, mcmc_iters=10) and its attempt_dispatch under a
                   fixed schedule with --max-concurrent=2 (GPEIOptChooser: pending branch, tuple return ->
                   add_to_grid), run by tests/run_reference_main.py with the reference's own choosers: the job ids in
                   dispatch order, their points and values as expt-grid.pkl holds them.
  ei_grad.npz      the refinement objective: GPEIOptChooser.grad_optimize_ei_over_hypers
                   without and with pending jobs, GPEIperSecChooser.grad_optimize_ei_over_hypers
                   (value + gradient at several points each).
"""
import os
import sys
import tempfile

import numpy as np
import numpy.random as npr

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_py3  # noqa: E402
from spearmint_amd.synthetic import synthetic_problem  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _mk_chooser(mod, cls, tmp, **kw):
    ch = getattr(mod, cls)(tmp, **kw)
    return ch


def _set(ch, hyper):
    ch.mean, ch.noise, ch.amp2, ch.ls = hyper[0], hyper[1], hyper[2], hyper[3:].copy()


def gen_ei_small(mods, tmp):
    cases = [  # name, N, M, D, H, seed
        ("a", 24, 300, 2, 3, 11),
        ("b", 64, 500, 8, 4, 12),
        ("c", 200, 700, 5, 3, 13),
        ("d", 130, 257, 32, 2, 14),
    ]
    for name, N, M, D, H, seed in cases:
        comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
        ch = _mk_chooser(mods["GPEIChooser"], "GPEIChooser", tmp, mcmc_iters=H)
        ch.D = D
        pend = np.zeros((0, D))
        ei = np.zeros((M, H))
        K = np.zeros((H, N, N)); Kstar = np.zeros((H, N, M))
        for h in range(H):
            _set(ch, hypers[h])
            ei[:, h] = ch.compute_ei(comp, pend, cand, vals)
            K[h] = ch.cov(comp) + ch.noise * np.eye(N)
            Kstar[h] = ch.cov(comp, cand)
        # the Opt chooser's ei_over_hypers must agree (same math)
        opt = _mk_chooser(mods["GPEIOptChooser"], "GPEIOptChooser", tmp, mcmc_iters=H)
        opt.D = D
        opt.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
        ei_opt = opt.ei_over_hypers(comp, pend, cand, vals)
        assert np.array_equal(ei, ei_opt)
        best = int(np.argmax(np.mean(ei, axis=1)))
        extra = dict(K=K, Kstar=Kstar) if name == "a" else {}
        np.savez_compressed(os.path.join(OUT, "ei_small_%s.npz" % name),
                            comp=comp, cand=cand, vals=vals, hypers=hypers,
                            ei=ei, best=best, **extra)
        # keep __del__ quiet
        _set(ch, hypers[0]); _set(opt, hypers[0])


def gen_pending(mods, tmp):
    N, M, D, H, P, S = 40, 200, 3, 2, 3, 7
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 21)
    pend = np.random.RandomState(22).rand(P, D)
    ch = _mk_chooser(mods["GPEIChooser"], "GPEIChooser", tmp, mcmc_iters=H, pending_samples=S)
    ch.D = D
    ei = np.zeros((M, H)); z = np.zeros((H, P, S))
    for h in range(H):
        _set(ch, hypers[h])
        npr.seed(100 + h)
        z[h] = npr.randn(P, S)
        npr.seed(100 + h)
        ei[:, h] = ch.compute_ei(comp, pend, cand, vals)
    np.savez_compressed(os.path.join(OUT, "ei_pending.npz"), comp=comp, cand=cand, pend=pend,
                        vals=vals, hypers=hypers, randn=z, ei=ei)


def gen_persec(mods, tmp):
    N, M, D, H = 50, 300, 4, 3
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(N, M, D, H, 31, per_sec=True)
    ch = _mk_chooser(mods["GPEIperSecChooser"], "GPEIperSecChooser", tmp, mcmc_iters=H)
    ch.D = D
    pend = np.zeros((0, D))
    ei = np.zeros((M, H))
    for h in range(H):
        _set(ch, hypers[h])
        ch.time_mean, ch.time_noise, ch.time_amp2, ch.time_ls = th[h, 0], th[h, 1], th[h, 2], th[h, 3:].copy()
        ei[:, h] = ch.compute_ei_per_s(comp, pend, cand, vals, log_durs)
    ch.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
    ch.time_hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in th]
    literal = ch.ei_over_hypers(comp, pend, cand, vals, log_durs)  # early-return bug: draw 0 only
    np.savez_compressed(os.path.join(OUT, "ei_persec.npz"), comp=comp, cand=cand, vals=vals,
                        hypers=hypers, log_durs=log_durs, time_hypers=th, ei=ei, literal=literal)


def branin(x0, x1):
    """examples/braninpy/branin.py:6-16 on the unit-cube point the driver
    passes in (config.pb: X is FLOAT size 2 in [0,1])."""
    a = x0 * 15
    b = (x1 * 15) - 5
    return (np.square(b - (5.1 / (4 * np.square(np.pi))) * np.square(a) + (5 / np.pi) * a - 6)
            + 10 * (1 - (1. / (8 * np.pi))) * np.cos(a) + 10)


def gen_branin_c1(mods, tmp):
    """Config 1: 2-D Branin, 20 observations, grid 1000, mcmc_iters 10,
    GPEIChooser.next driven exactly as attempt_dispatch does (S/main.py:205-210,
    :254): grid = i4_sobol_generate(2,1000,1).T, values NaN except completed."""
    sob = mods["sobol_lib"]
    grid = np.transpose(sob.i4_sobol_generate(2, 1000, 1))
    G = grid.shape[0]
    values = np.zeros(G) + np.nan
    durations = np.zeros(G) + np.nan
    status = np.zeros(G, dtype=int)          # 0 candidate, 2 complete
    for i in range(20):
        values[i] = branin(grid[i, 0], grid[i, 1])
        durations[i] = 1.0
        status[i] = 2
    candidates = np.nonzero(status == 0)[0]
    pending = np.nonzero(status == 1)[0]
    complete = np.nonzero(status == 2)[0]

    H = 10
    mod = mods["GPEIChooser"]
    # The reference sampler itself occasionally dies on raw Branin values
    # ("Slice sampler shrank to zero!", S/util.py:68-69); take the first seed
    # from 1234 upwards for which the reference's own next() completes.
    for seed in range(1234, 1300):
        # fresh expt_dir each time: GPEIChooser.__del__ pickles its state into
        # expt_dir and _real_init would pick a stale one up (GPEIChooser.py:66-98)
        ch = mod.GPEIChooser(tempfile.mkdtemp(prefix="spx_golden_c1_"), mcmc_iters=H)
        hyp = []; eis = []
        orig_ei = ch.compute_ei

        def rec_ei(comp, pend, cand, vals, ch=ch, orig_ei=orig_ei, hyp=hyp, eis=eis):
            hyp.append(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)))
            e = orig_ei(comp, pend, cand, vals)
            eis.append(e.copy())
            return e
        ch.compute_ei = rec_ei
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, candidates, pending, complete)
        except Exception as e:  # reference failure, try next seed
            print("seed", seed, "reference raised:", e)
            ch.ls = np.ones(2); ch.amp2 = ch.noise = ch.mean = 0.0
            continue
        break
    np.savez_compressed(os.path.join(OUT, "branin_c1.npz"), grid=grid, values=values,
                        durations=durations, candidates=candidates, pending=pending,
                        complete=complete, hypers=np.array(hyp), ei=np.array(eis).T,
                        job=int(job), seed=seed)


def _branin_inputs(mods, n_done=20, G=1000):
    sob = mods["sobol_lib"]
    grid = np.transpose(sob.i4_sobol_generate(2, G, 1))
    values = np.zeros(G) + np.nan
    durations = np.zeros(G) + np.nan
    status = np.zeros(G, dtype=int)
    for i in range(n_done):
        values[i] = branin(grid[i, 0], grid[i, 1])
        durations[i] = 1.0 + 3.0 * grid[i, 0] + np.sin(5 * grid[i, 1]) ** 2   # synthetic run times (s)
        status[i] = 2
    return (grid, values, durations, np.nonzero(status == 0)[0], np.nonzero(status == 1)[0],
            np.nonzero(status == 2)[0])


def gen_chooser_next(mods, tmp):
    """Whole-plugin golden runs: the reference's own GPEIOptChooser.next and
    GPEIperSecChooser.next on Branin (seeded numpy RNG, serial refinement)."""
    grid, values, durations, cand, pend, comp = _branin_inputs(mods, 12, 400)
    out = {}
    for seed in range(500, 540):
        ch = mods["GPEIOptChooser"].GPEIOptChooser(tempfile.mkdtemp(prefix="spx_golden_opt_"),
                                                   mcmc_iters=4, burnin=6, grid_subset=5,
                                                   use_multiprocessing=0)
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, cand, pend, comp)
        except Exception as e:
            print("opt seed", seed, "reference raised:", e)
            continue
        out.update(opt_seed=seed, opt_is_new=int(isinstance(job, tuple)),
                   opt_index=int(job[0] if isinstance(job, tuple) else job),
                   opt_point=np.asarray(job[1] if isinstance(job, tuple) else grid[job]),
                   opt_hypers=np.array([np.concatenate(([h[0], h[1], h[2]], h[3])) for h in ch.hyper_samples]))
        break
    for seed in range(600, 640):
        ch = mods["GPEIperSecChooser"].GPEIperSecChooser(tempfile.mkdtemp(prefix="spx_golden_ps_"),
                                                         mcmc_iters=3, burnin=4, grid_subset=4)
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, cand, pend, comp)
        except Exception as e:
            print("persec seed", seed, "reference raised:", e)
            continue
        out.update(ps_seed=seed, ps_is_new=int(isinstance(job, tuple)),
                   ps_index=int(job[0] if isinstance(job, tuple) else job),
                   ps_point=np.asarray(job[1] if isinstance(job, tuple) else grid[job]),
                   ps_hypers=np.array([np.concatenate(([h[0], h[1], h[2]], h[3])) for h in ch.hyper_samples]),
                   ps_time_hypers=np.array([np.concatenate(([h[0], h[1], h[2]], h[3]))
                                            for h in ch.time_hyper_samples]))
        break
    np.savez_compressed(os.path.join(OUT, "chooser_next.npz"), grid=grid, values=values,
                        durations=durations, candidates=cand, pending=pend, complete=comp, **out)


def gen_chooser_next_pending(mods, tmp):
    """The reference's GPEIChooser.next and GPEIOptChooser.next with PENDING experiments
    (fantasy branch), seeded, on Branin."""
    grid, values, durations, cand, pend, comp = _branin_inputs(mods, 12, 300)
    pend = cand[:3].copy(); cand = cand[3:].copy()       # three jobs still running
    out = {}
    for seed in range(700, 740):
        ch = mods["GPEIChooser"].GPEIChooser(tempfile.mkdtemp(prefix="spx_golden_pg_"), mcmc_iters=3,
                                             pending_samples=9)
        eis = []
        orig = ch.compute_ei
        ch.compute_ei = lambda c, p, x, v, orig=orig, eis=eis: (eis.append(orig(c, p, x, v)) or eis[-1])
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, cand, pend, comp)
        except Exception as e:
            print("pending gpei seed", seed, "reference raised:", e)
            ch.ls = np.ones(2); ch.amp2 = ch.noise = ch.mean = 0.0
            continue
        out.update(g_seed=seed, g_job=int(job), g_ei=np.array(eis).T)
        break
    for seed in range(800, 840):
        ch = mods["GPEIOptChooser"].GPEIOptChooser(tempfile.mkdtemp(prefix="spx_golden_po_"), mcmc_iters=3,
                                                   burnin=4, grid_subset=3, pending_samples=8,
                                                   use_multiprocessing=0)
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, cand, pend, comp)
        except Exception as e:
            print("pending opt seed", seed, "reference raised:", e)
            continue
        out.update(o_seed=seed, o_is_new=int(isinstance(job, tuple)),
                   o_index=int(job[0] if isinstance(job, tuple) else job),
                   o_point=np.asarray(job[1] if isinstance(job, tuple) else grid[job]))
        break
    for seed in range(900, 940):
        ch = mods["GPEIperSecChooser"].GPEIperSecChooser(tempfile.mkdtemp(prefix="spx_golden_pp_"), mcmc_iters=2,
                                                         burnin=3, grid_subset=3, pending_samples=6)
        npr.seed(seed)
        try:
            job = ch.next(grid, values, durations, cand, pend, comp)
        except Exception as e:
            print("pending persec seed", seed, "reference raised:", e)
            continue
        out.update(p_seed=seed, p_is_new=int(isinstance(job, tuple)),
                   p_index=int(job[0] if isinstance(job, tuple) else job),
                   p_point=np.asarray(job[1] if isinstance(job, tuple) else grid[job]))
        break
    np.savez_compressed(os.path.join(OUT, "chooser_next_pending.npz"), grid=grid, values=values,
                        durations=durations, candidates=cand, pending=pend, complete=comp, **out)


def gen_slice(mods, tmp):
    util = mods["util"]
    comp, cand, vals, hypers = synthetic_problem(30, 10, 3, 1, 41)
    gp = mods["gp"]
    import scipy.linalg as spla

    def lp_ls(ls):
        if np.any(ls < 0) or np.any(ls > 2):
            return -np.inf
        c = 1.3 * (gp.Matern52(ls, comp, None) + 1e-6 * np.eye(30)) + 1e-3 * np.eye(30)
        chol = spla.cholesky(c, lower=True)
        solve = spla.cho_solve((chol, True), vals - 0.1)
        return -np.sum(np.log(np.diag(chol))) - 0.5 * np.dot(vals - 0.1, solve)

    npr.seed(77)
    xs = [np.ones(3)]
    for _ in range(5):
        xs.append(util.slice_sample(xs[-1], lp_ls, compwise=True))
    npr.seed(78)
    ys = [np.array([0.3, 1.0, 0.5])]
    for _ in range(5):
        ys.append(util.slice_sample(ys[-1], lambda v: -0.5 * np.sum((v - 0.2) ** 2) / 0.3, compwise=False))
    np.savez_compressed(os.path.join(OUT, "slice_sampler.npz"), comp=comp, vals=vals,
                        compwise=np.array(xs), joint=np.array(ys),
                        lp_at_ones=lp_ls(np.ones(3)))


def gen_ml2(mods, tmp):
    """mcmc_iters=0: GPEIChooser.next with the ML-II hyper optimisation of gp.GP.optimize_hypers
    (gp.py:181-292).  (The same setting makes the reference's GPEIOptChooser and GPEIperSecChooser
    raise -- ValueError / UnboundLocalError -- so there is nothing to record for them.)"""
    grid, values, durations, cand, pend, comp = _branin_inputs(mods, 12, 400)
    ch = mods["GPEIChooser"].GPEIChooser(tempfile.mkdtemp(prefix="spx_golden_ml2_"), mcmc_iters=0)
    npr.seed(5)
    job = ch.next(grid, values, durations, cand, pend, comp)
    np.savez_compressed(os.path.join(OUT, "chooser_next_ml2.npz"), grid=grid, values=values, durations=durations,
                        candidates=cand, pending=pend, complete=comp, job=int(job),
                        hyper=np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)))


def gen_noiseless(mods, tmp):
    """noiseless=1 (noise pinned to 1e-3, the joint slice move over [mean, amp2] only:
    GPEIChooser.py:268-270, :316-346; GPEIOptChooser.py:621-626, :672-706): seeded next() of the three choosers."""
    grid, values, durations, cand, pend, comp = _branin_inputs(mods, 12, 300)
    out = {}
    specs = (("g", "GPEIChooser", dict(mcmc_iters=3, noiseless=1), 1000),
             ("o", "GPEIOptChooser", dict(mcmc_iters=3, burnin=4, grid_subset=3, noiseless=1, use_multiprocessing=0), 1100),
             ("p", "GPEIperSecChooser", dict(mcmc_iters=2, burnin=3, grid_subset=3, noiseless=1), 1200))
    for tag, name, kw, seed0 in specs:
        for seed in range(seed0, seed0 + 40):
            ch = getattr(mods[name], name)(tempfile.mkdtemp(prefix="spx_golden_nl_"), **kw)
            npr.seed(seed)
            try:
                job = ch.next(grid, values, durations, cand, pend, comp)
            except Exception as e:
                print("noiseless", name, "seed", seed, "reference raised:", e)
                if name == "GPEIChooser":
                    ch.ls = np.ones(2); ch.amp2 = ch.noise = ch.mean = 0.0
                continue
            out.update({tag + "_seed": seed, tag + "_is_new": int(isinstance(job, tuple)),
                        tag + "_index": int(job[0] if isinstance(job, tuple) else job),
                        tag + "_point": np.asarray(job[1] if isinstance(job, tuple) else grid[job]),
                        tag + "_hyper": np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls))})
            break
    np.savez_compressed(os.path.join(OUT, "chooser_next_noiseless.npz"), grid=grid, values=values,
                        durations=durations, candidates=cand, pending=pend, complete=comp, **out)


def gen_two_calls(mods, tmp):
    """Restart semantics (spearmint-lite builds a fresh chooser on every invocation): next() with 12 completed
    jobs, the chooser dropped (its state pickled), a NEW chooser object in the same expt_dir, next() with 14."""
    grid, values, durations, cand, pend, comp = _branin_inputs(mods, 14, 300)
    first = (grid, np.where(np.arange(len(values)) < 12, values, np.nan), np.where(np.arange(len(values)) < 12, durations, np.nan),
             np.arange(12, len(values)), pend, np.arange(12))
    second = (grid, values, durations, cand, pend, comp)
    out = {}
    specs = (("g", "GPEIChooser", dict(mcmc_iters=3), 1300),
             ("o", "GPEIOptChooser", dict(mcmc_iters=3, burnin=4, grid_subset=3, use_multiprocessing=0), 1400),
             ("p", "GPEIperSecChooser", dict(mcmc_iters=2, burnin=3, grid_subset=3), 1500))
    for tag, name, kw, seed0 in specs:
        for seed in range(seed0, seed0 + 40):
            d = tempfile.mkdtemp(prefix="spx_golden_2c_")
            try:
                ch = getattr(mods[name], name)(d, **kw)
                npr.seed(seed)
                job1 = ch.next(*first)
                h1 = np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls))
                if name == "GPEIChooser":
                    ch.__del__()                 # what the interpreter does when the driver lets go of it
                ch2 = getattr(mods[name], name)(d, **kw)
                npr.seed(seed + 7)
                job2 = ch2.next(*second)
                h2 = np.concatenate(([ch2.mean, ch2.noise, ch2.amp2], ch2.ls))
            except Exception as e:
                print("two calls", name, "seed", seed, "reference raised:", e)
                continue
            def idx(j): return int(j[0] if isinstance(j, tuple) else j)
            def pt(j): return np.asarray(j[1] if isinstance(j, tuple) else grid[j])
            out.update({tag + "_seed": seed, tag + "_new1": int(isinstance(job1, tuple)), tag + "_idx1": idx(job1),
                        tag + "_pt1": pt(job1), tag + "_h1": h1, tag + "_new2": int(isinstance(job2, tuple)),
                        tag + "_idx2": idx(job2), tag + "_pt2": pt(job2), tag + "_h2": h2})
            break
    np.savez_compressed(os.path.join(OUT, "chooser_two_calls.npz"), grid=grid, values=values, durations=durations,
                        candidates=cand, pending=pend, complete=comp, **out)


def run_trajectory(make_chooser, grid, iters, seed, first=2):
    """A whole optimisation run as spearmint-lite drives it (spearmint-lite.py:171-197 -- a fresh chooser per
    invocation, restarting from its state pickle; one proposal per invocation; a proposed NEW point is appended
    to the grid): returns the proposals [(is_new, index, point)] and the objective values."""
    grid = np.array(grid, copy=True)
    values = np.zeros(grid.shape[0]) + np.nan
    durations = np.zeros(grid.shape[0]) + np.nan
    done = np.zeros(grid.shape[0], dtype=bool)
    out = []
    for it in range(iters):
        ch = make_chooser()
        npr.seed(seed + it)
        cand, comp = np.nonzero(~done)[0], np.nonzero(done)[0]
        job = ch.next(grid, values, durations, cand, np.zeros(0, dtype=int), comp)
        if isinstance(job, tuple):
            idx, pt = int(job[0]), np.asarray(job[1], dtype=float).ravel()
            grid = np.vstack((grid, pt[None, :]))
            values, durations, done = np.append(values, np.nan), np.append(durations, np.nan), np.append(done, False)
            idx = grid.shape[0] - 1
            out.append((1, idx, pt))
        else:
            idx = int(job)
            out.append((0, idx, grid[idx].copy()))
        values[idx] = branin(grid[idx, 0], grid[idx, 1])
        durations[idx] = 1.0 + 3.0 * grid[idx, 0] + np.sin(5 * grid[idx, 1]) ** 2
        done[idx] = True
        if type(ch).__name__ == "GPEIChooser" and hasattr(ch, "ls"):
            ch.__del__()             # pickles the chain state; the interpreter calls it when the driver exits
    return out, values[np.isfinite(values)], grid


def gen_trajectory(mods, tmp):
    """Whole Branin runs of the reference's choosers under a spearmint-lite-style loop: the sequence of
    experiments a user would see (README.md:136-137 quotes the optimum 0.39 as the expected outcome)."""
    sob = mods["sobol_lib"]
    grid = np.transpose(sob.i4_sobol_generate(2, 400, 1))
    out = {"grid": grid}
    for tag, name, kw, iters, seed in (("g", "GPEIChooser", dict(mcmc_iters=4), 24, 7000),
                                       ("o", "GPEIOptChooser", dict(mcmc_iters=3, burnin=5, grid_subset=4, use_multiprocessing=0), 14, 7100),
                                       ("p", "GPEIperSecChooser", dict(mcmc_iters=2, burnin=4, grid_subset=3), 12, 7200)):
        d = tempfile.mkdtemp(prefix="spx_golden_traj_")
        props, vals, _ = run_trajectory(lambda: getattr(mods[name], name)(d, **kw), grid, iters, seed)
        out[tag + "_seed"], out[tag + "_iters"] = seed, iters
        out[tag + "_new"] = np.array([p[0] for p in props])
        out[tag + "_idx"] = np.array([p[1] for p in props])
        out[tag + "_pts"] = np.array([p[2] for p in props])
        out[tag + "_best"] = float(np.min([branin(p[2][0], p[2][1]) for p in props]))
        print("trajectory", name, "best", out[tag + "_best"], "new points", int(out[tag + "_new"].sum()))
    np.savez_compressed(os.path.join(OUT, "branin_trajectory.npz"), **out)


def gen_covar(mods, tmp):
    """The other covariance functions a chooser can be built with (covar=, gp.py:87-118): stage arrays and EI of
    compute_ei, the refinement objective, and whole seeded next() calls, all from the reference itself."""
    for kname in ("Matern32", "ARDSE", "SE"):
        out = {}
        # (1) K, K*, EI
        N, M, D, H = 80, 400, 4, 3
        comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 60 + len(kname))
        ch = _mk_chooser(mods["GPEIChooser"], "GPEIChooser", tmp, mcmc_iters=H, covar=kname)
        ch.D = D
        pend = np.zeros((0, D))
        ei = np.zeros((M, H)); K = np.zeros((H, N, N)); Kstar = np.zeros((H, N, M))
        for h in range(H):
            _set(ch, hypers[h])
            ei[:, h] = ch.compute_ei(comp, pend, cand, vals)
            K[h] = ch.cov(comp) + ch.noise * np.eye(N)
            Kstar[h] = ch.cov(comp, cand)
        out.update(comp=comp, cand=cand, vals=vals, hypers=hypers, ei=ei, K=K[0], Kstar=Kstar[0][:, :64],   # draw 0
                   best=int(np.argmax(np.mean(ei, axis=1))))
        # with two pending jobs (fantasies)
        S = 9
        chp = _mk_chooser(mods["GPEIChooser"], "GPEIChooser", tmp, mcmc_iters=H, covar=kname, pending_samples=S)
        chp.D = D
        pnd = np.random.RandomState(5).rand(2, D)
        eip = np.zeros((M, H)); z = np.zeros((H, 2, S))
        for h in range(H):
            _set(chp, hypers[h])
            npr.seed(300 + h)
            z[h] = npr.randn(2, S)
            npr.seed(300 + h)
            eip[:, h] = chp.compute_ei(comp, pnd, cand, vals)
        out.update(pend=pnd, randn=z, ei_pending=eip)
        # (2) the refinement objective
        opt = _mk_chooser(mods["GPEIOptChooser"], "GPEIOptChooser", tmp, mcmc_iters=H, covar=kname)
        opt.D = D
        _set(opt, hypers[0])
        opt.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
        rs = np.random.RandomState(7)
        pts = np.vstack((cand[:3], comp[np.argmin(vals)] + 1e-3 * rs.randn(D), rs.rand(D)))
        try:
            f = np.zeros(len(pts)); g = np.zeros((len(pts), D))
            for i, x in enumerate(pts):
                f[i], g[i] = opt.grad_optimize_ei_over_hypers(x.copy(), comp, pend, vals)
            out.update(points=pts, f=f, g=g, grad_raises=0)
        except AttributeError as e:
            print(kname, "refinement objective raises:", e)
            out.update(points=pts, grad_raises=1)
        # (3) whole next() calls on Branin
        grid, values, durations, cnd, pnd_idx, cmp_idx = _branin_inputs(mods, 12, 300)
        specs = [("g", "GPEIChooser", dict(mcmc_iters=3))]
        if kname != "SE":
            specs += [("o", "GPEIOptChooser", dict(mcmc_iters=3, burnin=4, grid_subset=3, use_multiprocessing=0)),
                      ("p", "GPEIperSecChooser", dict(mcmc_iters=2, burnin=3, grid_subset=3))]
        for tag, name, kw in specs:
            for seed in range(1600, 1640):
                c = getattr(mods[name], name)(tempfile.mkdtemp(prefix="spx_golden_cv_"), covar=kname, **kw)
                npr.seed(seed)
                try:
                    job = c.next(grid, values, durations, cnd, pnd_idx, cmp_idx)
                except Exception as e:
                    print(kname, name, "seed", seed, "reference raised:", e)
                    continue
                out.update({tag + "_seed": seed, tag + "_is_new": int(isinstance(job, tuple)),
                            tag + "_index": int(job[0] if isinstance(job, tuple) else job),
                            tag + "_point": np.asarray(job[1] if isinstance(job, tuple) else grid[job]),
                            tag + "_hyper": np.concatenate(([c.mean, c.noise, c.amp2], c.ls))})
                break
        # (4) mcmc_iters=0: ML-II hypers (gp.GP(covar).optimize_hypers) + one EI pass, GPEIChooser
        c = mods["GPEIChooser"].GPEIChooser(tempfile.mkdtemp(prefix="spx_golden_cv_"), covar=kname, mcmc_iters=0)
        npr.seed(5)
        out.update(ml2_job=int(c.next(grid, values, durations, cnd, pnd_idx, cmp_idx)),
                   ml2_hyper=np.concatenate(([c.mean, c.noise, c.amp2], c.ls)))
        out.update(grid=grid, values=values, durations=durations, candidates=cnd, pending=pnd_idx, complete=cmp_idx)
        np.savez_compressed(os.path.join(OUT, "covar_%s.npz" % kname), **out)


def gen_ei_grad(mods, tmp):
    """The L-BFGS-B objective of the local refinement, evaluated by the reference itself
    (GPEIOptChooser.py:360-525, GPEIperSecChooser.py:322-434)."""
    out = {}
    # (1) no pending jobs, two problem sizes
    for tag, (N, D, H, seed) in (("a", (40, 3, 3, 51)), ("b", (150, 9, 4, 52))):
        comp, cand, vals, hypers = synthetic_problem(N, 12, D, H, seed)
        opt = _mk_chooser(mods["GPEIOptChooser"], "GPEIOptChooser", tmp, mcmc_iters=H)
        opt.D = D
        _set(opt, hypers[0])
        opt.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
        rs = np.random.RandomState(seed)
        pts = np.vstack((cand[:4], comp[np.argmin(vals)] + 1e-3 * rs.randn(D), comp[3] + 1e-3 * rs.randn(D), rs.rand(D)))
        pend = np.zeros((0, D))
        f = np.zeros(len(pts)); g = np.zeros((len(pts), D))
        for i, x in enumerate(pts):
            f[i], g[i] = opt.grad_optimize_ei_over_hypers(x.copy(), comp, pend, vals)
        out.update({"%s_comp" % tag: comp, "%s_vals" % tag: vals, "%s_hypers" % tag: hypers,
                    "%s_points" % tag: pts, "%s_f" % tag: f, "%s_g" % tag: g})
    # (2) with pending jobs: fantasies from the replayed RNG state (:476-477)
    N, D, H, P, S, seed = 60, 4, 3, 3, 11, 53
    comp, cand, vals, hypers = synthetic_problem(N, 12, D, H, seed)
    rs = np.random.RandomState(seed)
    pend = rs.rand(P, D)
    opt = _mk_chooser(mods["GPEIOptChooser"], "GPEIOptChooser", tmp, mcmc_iters=H, pending_samples=S)
    opt.D = D
    _set(opt, hypers[0])
    opt.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
    npr.seed(4242)
    opt.randomstate = npr.get_state()
    randn = npr.randn(P, S)                      # what every grad_optimize_ei call will draw
    pts = np.vstack((cand[:3], comp[np.argmin(vals)] + 1e-3 * rs.randn(D), pend[0] + 1e-2 * rs.randn(D), rs.rand(D)))
    f = np.zeros(len(pts)); g = np.zeros((len(pts), D))
    for i, x in enumerate(pts):
        fi, gi = opt.grad_optimize_ei_over_hypers(x.copy(), comp, pend, vals)
        f[i], g[i] = float(np.ravel(fi)[0]), gi
    out.update(p_comp=comp, p_pend=pend, p_vals=vals, p_hypers=hypers, p_randn=randn, p_points=pts, p_f=f, p_g=g)
    # (3) EI per second
    N, D, H, seed = 70, 5, 3, 54
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(N, 12, D, H, seed, per_sec=True)
    ps = _mk_chooser(mods["GPEIperSecChooser"], "GPEIperSecChooser", tmp, mcmc_iters=H)
    ps.D = D
    _set(ps, hypers[0])
    ps.time_mean, ps.time_noise, ps.time_amp2, ps.time_ls = th[0, 0], th[0, 1], th[0, 2], th[0, 3:].copy()
    ps.hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in hypers]
    ps.time_hyper_samples = [(h[0], h[1], h[2], h[3:].copy()) for h in th]
    rs = np.random.RandomState(seed)
    pts = np.vstack((cand[:3], comp[np.argmin(vals)] + 1e-3 * rs.randn(D), rs.rand(D)))
    f = np.zeros(len(pts)); g = np.zeros((len(pts), D))
    for i, x in enumerate(pts):
        f[i], g[i] = ps.grad_optimize_ei_over_hypers(x.copy(), comp, vals, log_durs)
    out.update(s_comp=comp, s_vals=vals, s_log_durs=log_durs, s_hypers=hypers, s_time_hypers=th,
               s_points=pts, s_f=f, s_g=g)
    np.savez_compressed(os.path.join(OUT, "ei_grad.npz"), **out)


MAIN_LOOP_RUNS = {
    # tag: (mode, seed, schedule, main.py's own options)
    "g": ("main", 5, "", ["--method=GPEIChooser", "--method-args=mcmc_iters=10", "--grid-size=1000", "--grid-seed=1",
                          "--max-finished-jobs=8", "--polling-time=0.05"]),
    "o": ("dispatch", 9, "d,w0,d,w1,d,r0,r1,d,w2,d,w3,r2,d,w4,r3,r4,d,w5,d,w6,r5,r6",
          ["--method=GPEIOptChooser", "--method-args=mcmc_iters=10,burnin=20,grid_subset=5,use_multiprocessing=0",
           "--grid-size=1000", "--grid-seed=1", "--max-concurrent=2"]),
}


def run_main_loop(tag, engine, workdir, zip_path=None):
    """One run of tests/run_reference_main.py (a process of its own: the reference's main.py is a script, and its
    protobuf module needs the pure-Python protobuf implementation selected before google.protobuf is imported).
    Returns the JSON record."""
    import json
    import subprocess
    import zipfile
    mode, seed, sched, opts = MAIN_LOOP_RUNS[tag]
    tree = os.path.join(workdir, "tree_%s_%s" % (tag, engine))
    os.makedirs(tree)
    if zip_path is None:
        ref_py3.convert_main_tree(tree)
    else:
        with zipfile.ZipFile(zip_path) as z:
            z.extractall(tree)
    out = os.path.join(workdir, "rec_%s_%s.json" % (tag, engine))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "run_reference_main.py"), "--tree", tree, "--engine", engine,
           "--mode", mode, "--seed", str(seed), "--out", out]
    if sched:
        cmd += ["--schedule", sched]
    cmd += ["--"] + opts + [os.path.join(tree, "examples", "braninpy", "config.pb")]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    if p.returncode != 0:
        raise RuntimeError("run_reference_main failed:\n" + p.stderr.decode()[-4000:])
    rec = json.load(open(out))
    rec["tree"], rec["stderr"] = tree, p.stderr.decode()
    return rec


def gen_main_loop(mods, tmp):
    """SURVEY section 8 / BASELINE configs[0] through main.py itself, with the reference's own choosers in place."""
    out = {}
    for tag in sorted(MAIN_LOOP_RUNS):
        rec = run_main_loop(tag, "reference", tmp)
        order = rec["order"]
        out[tag + "_order"] = np.array(order)
        out[tag + "_points"] = np.array([rec["points"][str(j)] for j in order])
        out[tag + "_values"] = np.array([rec["values"][str(j)] for j in order])
        out[tag + "_grid_rows"] = rec["grid_rows"]
        if rec["steps"] and isinstance(rec["steps"], list):
            out[tag + "_step_job"] = np.array([-1 if s["job"] is None else s["job"] for s in rec["steps"]])
            out[tag + "_step_npending"] = np.array([len(s["pending_before"]) for s in rec["steps"]])
            out[tag + "_step_ncomplete"] = np.array([s["complete_before"] for s in rec["steps"]])
        print("main loop", tag, "jobs", order, "best", float(np.min(out[tag + "_values"])))
    np.savez_compressed(os.path.join(OUT, "main_loop.npz"), **out)


def main():
    if len(sys.argv) > 1:                    # python oracle/make_golden.py gen_main_loop [...]: only these generators
        os.makedirs(OUT, exist_ok=True)
        mods = ref_py3.load()
        for name in sys.argv[1:]:
            globals()[name](mods, tempfile.mkdtemp(prefix="spx_golden_"))
        return
    os.makedirs(OUT, exist_ok=True)
    mods = ref_py3.load()
    for gen in (gen_ei_small, gen_pending, gen_persec, gen_branin_c1, gen_chooser_next, gen_chooser_next_pending, gen_slice, gen_ei_grad, gen_ml2, gen_noiseless, gen_two_calls, gen_trajectory, gen_covar, gen_main_loop):
        gen(mods, tempfile.mkdtemp(prefix="spx_golden_"))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

"""BASELINE.json configurations at FULL size on one MI355X against the CPU oracle
(SURVEY.md 8(d) parity criterion: identical best index; EI within 1e-5 relative -- asserted at
1e-6 -- on a candidate subsample >= 20k plus the oracle over every candidate that could win).

The oracle cannot score 4e6 .. 2e7 evaluations in test time, so the argmax is pinned by a margin
argument: the oracle scores the GPU's top-K candidates by mean EI (all draws) plus a large random
sample; on that set EI agrees to <= 1e-6 and the oracle's own argmax is the GPU's; every candidate
outside the top-K has a GPU mean EI below (1 - 1e-3) x the best, i.e. a thousand tolerances away
from winning."""
import numpy as np
import pytest

import bench
from oracle import gp_ei_oracle as orc
from spearmint_amd import dist as sd
from spearmint_amd.engine import FLAG_KEEP_MOMENTS, FLAG_PER_SEC
from spearmint_amd.synthetic import synthetic_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from spearmint_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def rel_err(got, ref):
    ok = np.isfinite(ref) & (ref >= 1e-280)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    return float(np.max(np.abs(got[ok] - ref[ok]) / ref[ok])) if ok.any() else 0.0


def check_winner(mean, idx, draws_sub, ref_sub, sub, topk):
    """draws_sub / ref_sub: GPU and oracle EI (len(sub), H) on the checked candidates `sub`,
    whose first `topk` entries are the GPU's top-k by mean EI."""
    assert rel_err(draws_sub, ref_sub) <= 1e-6
    assert sub[orc.choose(ref_sub)] == idx                      # the oracle's own argmax over the checked set
    order = np.argsort(mean)
    assert order[-1] == idx or mean[order[-1]] == mean[idx]
    unchecked_best = mean[order[-topk - 1]]                     # best candidate the oracle did not score
    assert unchecked_best < (1.0 - 1e-3) * mean[idx], (unchecked_best, mean[idx])


def test_c3_full_size_oracle_decides_the_argmax(eng):
    """C3: N_obs=2048, 32-D, 200 000 candidates, 20 draws."""
    N, M, D, H = 2048, 200000, 32, 20
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 3000)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert np.array_equal(mean, np.mean(draws, axis=1)) and val == mean[idx]      # numpy order on the device
    assert np.isfinite(draws).all() and (draws >= 0).all()
    topk = 2000
    top = np.argsort(mean)[::-1][:topk]
    rnd = np.random.RandomState(0).choice(M, 20000, replace=False)
    sub = np.concatenate((top, np.setdiff1d(rnd, top)))
    ref = orc.ei_grid_chunked(comp, cand[sub], vals, hypers, chunk=20000)          # all 20 draws, ~70 s
    check_winner(mean, idx, draws[sub], ref, sub, topk)
    # per-candidate results do not depend on which other candidates share the launch
    i2, _, _, d2 = eng.ei_grid(comp, vals, cand[sub[:3000]], hypers, want_draws=True)
    assert np.array_equal(d2, draws[sub[:3000]]) and sub[i2] == idx


def test_c3_full_size_every_value_against_the_chunked_oracle(eng):
    """SURVEY 8(d)'s criterion literally, at the headline configuration: "the full-size oracle argmax computed in chunks"
    (GPEIChooser.py:143-153) -- EVERY one of the 200 000 x 20 EI values of C3 against the oracle (<= 1e-6 relative), and
    the oracle's own argmax of the mean over all 200 000 candidates equals the GPU's winner.  4e6 oracle evaluations are
    minutes of host BLAS; the first chunk is timed and the test is skipped -- loudly -- if the rest would not fit
    SPX_FULL_C3_BUDGET seconds (default 480; scripts/gate.sh sets SPX_FULL_C3=1: no skipping)."""
    import os
    import time
    N, M, D, H = 2048, 200000, 32, 20
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 3000)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    chunk = 20000
    budget = float(os.environ.get("SPX_FULL_C3_BUDGET", "480"))
    force = os.environ.get("SPX_FULL_C3", "0") not in ("", "0")
    ref = np.empty((M, H))
    worst = 0.0
    t0 = time.time()
    for c0 in range(0, M, chunk):
        ref[c0:c0 + chunk] = orc.ei_grid_chunked(comp, cand[c0:c0 + chunk], vals, hypers, chunk=chunk)
        worst = max(worst, rel_err(draws[c0:c0 + chunk], ref[c0:c0 + chunk]))
        assert worst <= 1e-6, (c0, worst)
        if c0 == 0 and not force:
            est = (time.time() - t0) * (M // chunk)
            if est > budget:
                pytest.skip("the oracle needs ~%.0f s for all 4e6 evaluations on this host (budget %.0f s; the first %d x %d "
                            "agreed to %.1e); SPX_FULL_C3=1 forces the full run" % (est, budget, chunk, H, worst))
    assert orc.choose(ref) == idx                                # the oracle's own argmax over ALL candidates
    assert np.argmax(np.mean(ref, axis=1)) == np.argmax(mean)
    print("C3 every value: worst relative EI error %.2e over %d evaluations, %.0f s of oracle" % (worst, M * H, time.time() - t0))


@pytest.mark.parametrize("noise", [None, 1e-3])
def test_stage_arrays_at_2048_observations(eng, noise):
    """Per-stage criteria of SURVEY 8(d) at the largest N of BASELINE.json, with sampled noise and
    with the noiseless setting (noise pinned to 1e-3, GPEIChooser.py:270), where cond(K) is worst."""
    N, M, D, H = 2048, 3000, 32, 2
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 3100)
    if noise is not None:
        hypers[:, 1] = noise
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    eng.factor()
    eng.ei_run(FLAG_KEEP_MOMENTS)
    draws = eng.ei_draws()
    for h in range(H):
        st = {}
        ref = orc.compute_ei(comp, cand, vals, hypers[h], stages=st)
        K, L, alpha = eng.get_factor(h)
        assert np.allclose(K, st["K"], rtol=1e-12, atol=1e-14)
        assert np.linalg.norm(L @ L.T - st["K"]) / np.linalg.norm(st["K"]) <= 1e-13
        assert np.allclose(alpha, st["alpha"], rtol=1e-7, atol=1e-9 * np.abs(st["alpha"]).max())
        m, v = eng.get_moments(h)
        assert np.allclose(m, st["func_m"], rtol=1e-9, atol=1e-9)
        # func_v = amp2(1+1e-6) - |beta|^2 cancels near the observations: both LAPACK's substitution and
        # the W K* product carry an error ~ eps cond(L) |beta|^2, so the criterion is relative to the prior
        assert np.max(np.abs(v - st["func_v"])) <= 1e-9 * hypers[h, 2]
        assert rel_err(draws[:, h], ref) <= 1e-6


def test_c2_noiseless_at_1024_observations(eng):
    comp, cand, vals, hypers = synthetic_problem(1024, 20000, 8, 4, 2100)
    hypers[:, 1] = 1e-3
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_grid_chunked(comp, cand, vals, hypers)
    assert rel_err(draws, ref) <= 1e-6
    assert idx == orc.choose(ref)


def _strong(name):
    cfg = dict(bench.STRONG[name])
    prob, comp, vals, hypers = bench.strong_problem(cfg)
    cand = bench.strong_rows(cfg, comp, vals, 0, cfg["M"])
    return cfg, prob, comp, vals, hypers, cand


def test_c4_full_million_candidates_and_eight_shards(eng):
    """C4: 1 000 000 candidates (bench.py's strong-scaling grid, jittered incumbents in front) on one
    GPU; the 8 contiguous shards of the 8-GPU run give the same bits and the same winner."""
    cfg, _, comp, vals, hypers, cand = _strong("c4")
    M, H = cfg["M"], cfg["H"]
    eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.factor()
    eng.set_candidates(cand)
    eng.ei_run()
    idx, val = eng.best()
    mean = eng.ei_mean()
    draws = eng.ei_draws()
    assert np.array_equal(mean, np.mean(draws, axis=1)) and val == mean[idx] and idx == int(np.argmax(mean))
    recs = []
    for r in range(8):
        lo, hi = sd.shard_bounds(M, 8, r)
        eng.set_candidates(cand[lo:hi], index_base=lo)
        eng.ei_run()
        assert np.array_equal(eng.ei_mean(), mean[lo:hi])          # bit-identical per candidate
        recs.append(list(eng.best()[::-1]))
    assert sd.pick_best(recs) == (idx, val)
    topk = 1000
    top = np.argsort(mean)[::-1][:topk]
    rnd = np.random.RandomState(1).choice(M, 20000, replace=False)      # an oracle subsample >= 20k
    sub = np.concatenate((top, np.setdiff1d(rnd, top)))
    ref = orc.ei_grid_chunked(comp, cand[sub], vals, hypers)
    check_winner(mean, idx, draws[sub], ref, sub, topk)


def test_c5_full_half_million_candidates_per_second(eng):
    """C5: GPEIperSec dual GP, 16-D, N_obs=1024, 500 000 candidates, 20 draws."""
    cfg, prob, comp, vals, hypers, cand = _strong("c5")
    log_durs, th = prob[4], prob[5]
    M = cfg["M"]
    idx, val, mean, draws = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    assert np.array_equal(mean, np.mean(draws, axis=1)) and val == mean[idx] and idx == int(np.argmax(mean))
    recs = []
    for r in range(8):       # the factorisations (both GPs) stay resident; only the shard changes
        lo, hi = sd.shard_bounds(M, 8, r)
        eng.set_candidates(cand[lo:hi], index_base=lo)
        eng.ei_run(FLAG_PER_SEC)
        assert np.array_equal(eng.ei_mean(), mean[lo:hi])
        recs.append(list(eng.best()[::-1]))
    assert sd.pick_best(recs) == (idx, val)
    topk = 1000
    top = np.argsort(mean)[::-1][:topk]
    rnd = np.random.RandomState(2).choice(M, 20000, replace=False)      # an oracle subsample >= 20k
    sub = np.concatenate((top, np.setdiff1d(rnd, top)))
    ref = orc.ei_per_s_over_hypers(comp, cand[sub], vals, log_durs, hypers, th)
    check_winner(mean, idx, draws[sub], ref, sub, topk)


# ---- sizes beyond BASELINE.json (the reference has no limits of its own; these pin ours) ----------
@pytest.mark.parametrize("N,M,D,H,seed", [(4096, 1500, 48, 2, 8100), (8192, 600, 16, 1, 8200), (300, 900, 300, 2, 8300),
                                          (500, 700, 6, 50, 8400)])
def test_large_observation_counts_dimensions_and_draw_counts(eng, N, M, D, H, seed):
    """N = 4096 and 8192 observations (64 / 128 block columns), 300 input dimensions (10 Gram chunks),
    50 hyper draws in one call -- against the oracle, every value."""
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
    idx, _, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_grid_chunked(comp, cand, vals, hypers)
    assert rel_err(draws, ref) <= 1e-6
    assert idx == orc.choose(ref)
    assert np.array_equal(mean, np.mean(draws, axis=1))
    eng.set_observations(comp, vals); eng.set_hypers(hypers[:min(H, 3)])
    lp = eng.gp_logprob()
    for h in range(len(lp)):
        lref = orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:])
        assert np.isclose(lp[h], lref, rtol=1e-10)

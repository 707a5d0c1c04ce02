"""The other covariance functions a chooser can be built with -- covar=Matern32 / ARDSE / SE
(spearmint/spearmint/gp.py:87-118; option "covar" of a libspx handle) -- against the vectors the reference
itself produced (tests/golden/covar_*.npz) and against the oracle, on the GPU through the C ABI."""
import os

import numpy as np
import pytest

from oracle import gp_ei_oracle as orc
from spearmint_amd.synthetic import synthetic_problem
from tests.test_gpu_a_parity import assert_ei_close

pytestmark = pytest.mark.gpu
KINDS = ["Matern32", "ARDSE", "SE"]


@pytest.fixture()
def eng():
    from spearmint_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _g(golden_dir, kname):
    return np.load(os.path.join(golden_dir, "covar_%s.npz" % kname))


@pytest.mark.parametrize("kname", KINDS)
def test_stage_arrays_and_ei_match_reference(eng, golden_dir, kname):
    g = _g(golden_dir, kname)
    eng.set_covar(kname)
    idx, val, mean, draws = eng.ei_grid(g["comp"], g["vals"], g["cand"], g["hypers"], want_draws=True)
    assert_ei_close(draws, g["ei"])
    assert idx == int(g["best"])
    K, L, alpha = eng.get_factor(0)
    assert np.allclose(K, g["K"], rtol=1e-13, atol=1e-15)
    assert np.allclose(L @ L.T, g["K"], rtol=1e-12, atol=1e-14)
    assert np.allclose(eng.get_cross_cov(0)[:, :64], g["Kstar"], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("kname", KINDS)
def test_pending_branch_matches_reference(eng, golden_dir, kname):
    g = _g(golden_dir, kname)
    comp, pend, vals, hypers = g["comp"], g["pend"], g["vals"], g["hypers"]
    H, P = hypers.shape[0], pend.shape[0]
    with orc.covar(kname):
        fb = [orc.fantasize(comp, pend, vals, hypers[h], g["randn"][h]) for h in range(H)]
    eng.set_covar(kname)
    eng.set_observations(np.concatenate((comp, pend)), np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(g["cand"]); eng.set_hypers(hypers); eng.factor()
    eng.set_fantasies(np.array([x[0] for x in fb]), np.array([x[1] for x in fb]))
    eng.ei_run()
    assert_ei_close(eng.ei_draws(), g["ei_pending"])


@pytest.mark.parametrize("kname", ["Matern32", "ARDSE"])
def test_refinement_objective_matches_reference(eng, golden_dir, kname):
    g = _g(golden_dir, kname)
    eng.set_covar(kname)
    eng.ei_grid(g["comp"], g["vals"], g["points"], g["hypers"])
    f, gr = eng.ei_grad_batch(g["points"])
    assert np.allclose(f, g["f"], rtol=1e-7, atol=1e-300)
    assert np.allclose(gr, g["g"], rtol=1e-6, atol=1e-9 * np.abs(g["g"]).max())


@pytest.mark.parametrize("kname", KINDS)
@pytest.mark.parametrize("N,M,D,H,seed", [(300, 3000, 12, 4, 71), (1100, 1500, 33, 2, 72)])
def test_oracle_parity_and_loglikelihood(eng, kname, N, M, D, H, seed):
    """Larger shapes against the oracle: EI, argmax, per-second EI and the log-likelihood (both factorisation
    paths); one draw has very short length scales, where exp(-r^2/2) underflows to 0 as numpy's does."""
    comp, cand, vals, hypers, ld, th = synthetic_problem(N, M, D, H, seed, per_sec=True)
    hypers[H - 1, 3:] = 0.02
    eng.set_covar(kname)
    with orc.covar(kname):
        ref = orc.ei_over_hypers(comp, cand, vals, hypers)
        ref_ps = orc.ei_per_s_over_hypers(comp, cand, vals, ld, hypers, th)
        lp_ref = [orc.gp_logprob(comp, vals, h[0], h[2], h[1], h[3:]) for h in hypers]
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert_ei_close(draws, ref)
    assert idx == orc.choose(ref)
    idx, _, _, draws = eng.ei_per_sec_grid(comp, vals, ld, cand, hypers, th, want_draws=True)
    assert_ei_close(draws, ref_ps)
    assert idx == orc.choose(ref_ps)
    eng.set_observations(comp, vals); eng.set_hypers(hypers)
    assert np.allclose(eng.gp_logprob(), lp_ref, rtol=1e-10)
    eng.set_hypers(np.repeat(hypers, 12, axis=0))          # > 32 rows: the row-major factorisation
    assert np.allclose(eng.gp_logprob(), np.repeat(lp_ref, 12), rtol=1e-10)


def test_se_ignores_length_scales_and_option_is_validated(eng):
    comp, cand, vals, hypers = synthetic_problem(100, 500, 5, 2, 73)
    eng.set_covar("SE")
    a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)[3]
    other = hypers.copy(); other[:, 3:] = 0.37
    b = eng.ei_grid(comp, vals, cand, other, want_draws=True)[3]
    assert np.array_equal(a, b)
    ones = hypers.copy(); ones[:, 3:] = 1.0
    eng.set_covar("ARDSE")
    assert np.array_equal(eng.ei_grid(comp, vals, cand, ones, want_draws=True)[3], a)
    # changing the covariance invalidates the factorisation
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.set_covar("Matern32")
    with pytest.raises(Exception):
        eng.ei_run()
    with pytest.raises(AttributeError):
        eng.set_covar("Periodic")
    with pytest.raises(ValueError):
        eng.set_option("covar", 9)
    eng.set_covar("Matern52")
    c = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)[3]
    assert_ei_close(c, orc.ei_over_hypers(comp, cand, vals, hypers))


@pytest.mark.parametrize("kname", KINDS)
@pytest.mark.parametrize("extra", ["", ",gpu_logprob=1,gpu_refine=1"])
def test_choosers_on_gpu_match_reference(golden_dir, tmp_path, kname, extra):
    """The reference's seeded next() calls with covar=..., on the real engine (with the device log-likelihood
    and the device refinement forced on as well)."""
    from tests.test_host_logic import _covar_runs
    _covar_runs(golden_dir, tmp_path, kname, lambda k: None, extra)

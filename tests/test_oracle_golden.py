"""The CPU oracle must reproduce what the REFERENCE ITSELF produced.

tests/golden/*.npz were written by oracle/make_golden.py, which runs the
lib2to3-converted reference (build container only).  These tests need neither
the reference tree nor a GPU."""
import os

import numpy as np
import numpy.random as npr
import pytest

from oracle import gp_ei_oracle as orc

TOL = 1e-12  # same math, same library calls; allows for a different BLAS build


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(a, b, rtol=TOL):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.allclose(a[m], b[m], rtol=rtol, atol=1e-300)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_ei_matches_reference(golden_dir, case):
    g = _load(golden_dir, "ei_small_%s.npz" % case)
    ei = orc.ei_over_hypers(g["comp"], g["cand"], g["vals"], g["hypers"])
    _close(ei, g["ei"], rtol=1e-9)
    assert orc.choose(ei) == int(g["best"])
    # chunked evaluation (what the CPU baseline times) is the same function
    ei_c = orc.ei_grid_chunked(g["comp"], g["cand"], g["vals"], g["hypers"], chunk=97)
    _close(ei_c, g["ei"], rtol=1e-9)


def test_stage_arrays_match_reference(golden_dir):
    g = _load(golden_dir, "ei_small_a.npz")
    for h in range(g["hypers"].shape[0]):
        st = {}
        orc.compute_ei(g["comp"], g["cand"], g["vals"], g["hypers"][h], stages=st)
        _close(st["K"], g["K"][h])
        _close(st["Kstar"], g["Kstar"][h])


def test_pending_matches_reference(golden_dir):
    g = _load(golden_dir, "ei_pending.npz")
    for h in range(g["hypers"].shape[0]):
        ei = orc.compute_ei_pending(g["comp"], g["pend"], g["cand"], g["vals"],
                                    g["hypers"][h], g["randn"][h])
        _close(ei, g["ei"][:, h], rtol=1e-9)


def test_fantasy_half_of_pending_branch(golden_dir):
    """compute_ei_fantasies (what the GPU path mirrors) + the host fantasy draw
    reproduce the reference's pending branch."""
    from spearmint_amd import hostgp
    g = _load(golden_dir, "ei_pending.npz")
    comp, pend, cand, vals = g["comp"], g["pend"], g["cand"], g["vals"]
    comp_pend = np.concatenate((comp, pend))
    for h in range(g["hypers"].shape[0]):
        chol = orc.posterior(comp_pend, np.concatenate((vals, np.zeros(len(pend)))), g["hypers"][h])[1]
        fant, bests = hostgp.fantasize_pending(comp, pend, vals, g["hypers"][h], chol[:len(comp), :len(comp)],
                                               g["randn"][h])
        ei = orc.compute_ei_fantasies(comp_pend, cand, g["hypers"][h], fant, bests)
        _close(ei, g["ei"][:, h], rtol=1e-9)


def test_persec_matches_reference(golden_dir):
    g = _load(golden_dir, "ei_persec.npz")
    ei = orc.ei_per_s_over_hypers(g["comp"], g["cand"], g["vals"], g["log_durs"],
                                  g["hypers"], g["time_hypers"])
    _close(ei, g["ei"], rtol=1e-9)
    lit = orc.ei_per_s_over_hypers(g["comp"], g["cand"], g["vals"], g["log_durs"],
                                   g["hypers"], g["time_hypers"], ref_compat=True)
    _close(lit, g["literal"], rtol=1e-9)
    assert np.all(lit[:, 1:] == 0.0)  # GPEIperSecChooser.py:302 early return


def test_branin_c1_ei_and_choice(golden_dir):
    """BASELINE config 1: the reference's own next() on examples/braninpy."""
    g = _load(golden_dir, "branin_c1.npz")
    grid, values = g["grid"], g["values"]
    comp = grid[g["complete"]]; cand = grid[g["candidates"]]; vals = values[g["complete"]]
    ei = orc.ei_over_hypers(comp, cand, vals, g["hypers"])
    _close(ei, g["ei"], rtol=1e-9)
    assert int(g["candidates"][orc.choose(ei)]) == int(g["job"])


def test_slice_sampler_matches_reference(golden_dir):
    g = _load(golden_dir, "slice_sampler.npz")
    comp, vals = g["comp"], g["vals"]

    def lp_ls(ls):
        if np.any(ls < 0) or np.any(ls > 2):
            return -np.inf
        return orc.gp_logprob(comp, vals, 0.1, 1.3, 1e-3, ls)

    assert np.isclose(lp_ls(np.ones(3)), float(g["lp_at_ones"]), rtol=1e-12)
    npr.seed(77)
    x = g["compwise"][0]
    for k in range(1, g["compwise"].shape[0]):
        x = orc.slice_sample(x, lp_ls, compwise=True)
        _close(x, g["compwise"][k], rtol=1e-9)
    npr.seed(78)
    y = g["joint"][0]
    for k in range(1, g["joint"].shape[0]):
        y = orc.slice_sample(y, lambda v: -0.5 * np.sum((v - 0.2) ** 2) / 0.3, compwise=False)
        _close(y, g["joint"][k], rtol=1e-9)


def test_unpack_args():
    assert orc.unpack_args("mcmc_iters=20, noiseless = 1") == {"mcmc_iters": "20", "noiseless": "1"}
    assert orc.unpack_args("") == {}


def test_known_answers():
    # Matern at r=0 is 1; N=1 closed form for the posterior
    x = np.array([[0.3, 0.7]])
    assert orc.matern52(np.ones(2), x)[0, 0] == 1.0
    hyper = np.array([0.0, 1e-3, 2.0, 1.0, 1.0])
    cand = np.array([[0.3, 0.7], [0.9, 0.1]])
    st = {}
    orc.compute_ei(x, cand, np.array([1.5]), hyper, stages=st)
    k = 2.0 * (1 + 1e-6) + 1e-3
    assert np.isclose(st["func_m"][0], 2.0 * 1.5 / k)
    assert np.isclose(st["func_v"][0], 2.0 * (1 + 1e-6) - 4.0 / k)
    # numpy argmax/mean tie + NaN rules the device code must reproduce
    assert np.argmax(np.array([1.0, np.nan, 5.0, np.nan])) == 1
    assert np.argmax(np.array([2.0, 5.0, 5.0])) == 1


def test_refinement_objective_matches_reference(golden_dir):
    """grad_optimize_ei_over_hypers of the reference (GPEIOptChooser.py:360-525,
    GPEIperSecChooser.py:322-434): value and gradient, without / with pending jobs and per second."""
    g = np.load(os.path.join(golden_dir, "ei_grad.npz"))
    for tag in "ab":
        for x, f_ref, g_ref in zip(g[tag + "_points"], g[tag + "_f"], g[tag + "_g"]):
            f, gr = orc.grad_optimize_ei_over_hypers(x, g[tag + "_comp"], g[tag + "_vals"], g[tag + "_hypers"])
            assert np.isclose(f, f_ref, rtol=1e-12, atol=0) and np.allclose(gr, g_ref, rtol=1e-10, atol=1e-300)
    for x, f_ref, g_ref in zip(g["p_points"], g["p_f"], g["p_g"]):
        f, gr = orc.grad_optimize_ei_over_hypers(x, g["p_comp"], g["p_vals"], g["p_hypers"],
                                                 pend=g["p_pend"], randn_ps=g["p_randn"])
        assert np.isclose(f, f_ref, rtol=1e-12, atol=0) and np.allclose(gr, g_ref, rtol=1e-10, atol=1e-300)
    for x, f_ref, g_ref in zip(g["s_points"], g["s_f"], g["s_g"]):
        f, gr = orc.grad_optimize_ei_over_hypers(x, g["s_comp"], g["s_vals"], g["s_hypers"],
                                                 log_durs=g["s_log_durs"], time_hypers=g["s_time_hypers"])
        assert np.isclose(f, f_ref, rtol=1e-12, atol=0) and np.allclose(gr, g_ref, rtol=1e-10, atol=1e-300)


def test_refinement_objective_is_a_gradient():
    """Central differences of the restated objective: the reference's gradient carries a factor
    one half (GPEIOptChooser.py:437), so d f / d x = 2 * grad."""
    from spearmint_amd.synthetic import synthetic_problem
    comp, cand, vals, hypers = synthetic_problem(50, 5, 3, 2, 91, near=0)
    x = cand[1].copy()
    f0, gr = orc.grad_optimize_ei_over_hypers(x, comp, vals, hypers)
    for d in range(3):
        e = np.zeros(3); e[d] = 1e-6
        num = (orc.grad_optimize_ei_over_hypers(x + e, comp, vals, hypers)[0]
               - orc.grad_optimize_ei_over_hypers(x - e, comp, vals, hypers)[0]) / 2e-6
        assert np.isclose(0.5 * num, gr[d], rtol=1e-4, atol=1e-10)


@pytest.mark.parametrize("kname", ["Matern32", "ARDSE", "SE"])
def test_other_covariances_match_reference(golden_dir, kname):
    """covar= Matern32 / ARDSE / SE (gp.py:87-118): K, K*, EI, the pending branch and the refinement objective
    of the reference's own choosers built with that covariance."""
    g = _load(golden_dir, "covar_%s.npz" % kname)
    with orc.covar(kname):
        st = {}
        orc.compute_ei(g["comp"], g["cand"], g["vals"], g["hypers"][0], stages=st)
        _close(st["K"], g["K"])
        _close(st["Kstar"][:, :64], g["Kstar"])
        ei = orc.ei_over_hypers(g["comp"], g["cand"], g["vals"], g["hypers"])
        _close(ei, g["ei"], rtol=1e-9)
        assert orc.choose(ei) == int(g["best"])
        for h in range(g["hypers"].shape[0]):
            eip = orc.compute_ei_pending(g["comp"], g["pend"], g["cand"], g["vals"], g["hypers"][h], g["randn"][h])
            _close(eip, g["ei_pending"][:, h], rtol=1e-9)
        if int(g["grad_raises"]):
            with pytest.raises(AttributeError):       # gp.py has no grad_SE
                orc.grad_optimize_ei_over_hypers(g["points"][0], g["comp"], g["vals"], g["hypers"])
        else:
            for x, f_ref, g_ref in zip(g["points"], g["f"], g["g"]):
                f, gr = orc.grad_optimize_ei_over_hypers(x, g["comp"], g["vals"], g["hypers"])
                assert np.isclose(f, f_ref, rtol=1e-12, atol=0) and np.allclose(gr, g_ref, rtol=1e-10, atol=1e-300)
    # the default is untouched afterwards
    assert orc._active_covar == "Matern52"


def test_bench_cpu_baseline_times_the_reference_itself():
    """bench.py's `cpu_baseline`: with the reference tree (build container) or build()'s archive oracle/_ref/chooser_py3.zip
    (GPU box) present the timed CPU code is the reference's OWN GPEIChooser.compute_ei loop (kind "reference") and the
    oracle's restatement, timed beside it, returns the same EI matrix bit for bit; without either it is the port."""
    import bench
    from oracle import ref_py3
    w = {"N": 60, "D": 3, "H": 4, "M": 5000}
    have = ref_py3.available() or os.path.isfile(ref_py3.CHOOSER_ZIP)
    out = bench.cpu_baseline(w, 700, 1)
    assert out["unit"] == "EI evals/s" and out["value"] > 0 and out["cores"] >= 1
    if have:
        assert out["kind"] == "reference" and out["port_equals_reference"] is True and out["port_value"] > 0
        assert "GPEIChooser.compute_ei" in out["sample"]
    else:
        assert out["kind"] == "port"

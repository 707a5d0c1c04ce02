"""Parity of the HIP path (through the C ABI, libspx.so) with the CPU oracle and
with the golden vectors produced by running the reference.  Needs an MI355X.

Tolerances (BASELINE.json north_star): identical argmax index for fixed
hyper-parameters; EI within 1e-5 relative (we assert 1e-7: measured ~1e-9)
for every candidate with EI >= 1e-280, NaN positions identical.
"""
import os

import numpy as np
import numpy.random as npr
import pytest

from oracle import gp_ei_oracle as orc
from spearmint_amd import dist as sd
from spearmint_amd.engine import Engine, FLAG_KEEP_MOMENTS, FLAG_PER_SEC
from spearmint_amd.synthetic import synthetic_problem

pytestmark = pytest.mark.gpu

EI_RTOL = 1e-7


@pytest.fixture(scope="module")
def eng():
    from spearmint_amd.engine import Engine
    e = Engine(0)   # raises if libspx.so or the GPU is missing -- no fallback
    yield e
    e.close()


def assert_ei_close(got, ref, rtol=EI_RTOL):
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    big = np.isfinite(ref) & (ref >= 1e-280)
    if big.any():
        assert np.max(np.abs(got[big] - ref[big]) / ref[big]) <= rtol
    small = np.isfinite(ref) & (ref < 1e-280)
    assert np.all(got[small] <= 1e-270)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---- golden vectors (outputs of the reference itself) -------------------------
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_golden_ei(eng, golden_dir, case):
    g = _g(golden_dir, "ei_small_%s.npz" % case)
    idx, val, mean, draws = eng.ei_grid(g["comp"], g["vals"], g["cand"], g["hypers"], want_draws=True)
    assert_ei_close(draws, g["ei"])
    assert idx == int(g["best"])
    assert np.array_equal(mean, np.mean(draws, axis=1))      # numpy summation order on the device
    assert val == mean[idx]


@pytest.mark.parametrize("H", [129, 300, 1100])
def test_mean_over_more_than_128_draws_is_numpys(eng, H):
    """np.mean(overall_ei, axis=1) beyond 128 draws: numpy's pairwise halving, walked without recursion on the device
    (csrc/np_sum.h; the same header is compiled for the host in tests/test_host_logic.py)."""
    comp, cand, vals, hypers = synthetic_problem(24, 700, 3, H, 1000 + H)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert np.array_equal(mean, np.mean(draws, axis=1))
    assert idx == int(np.argmax(mean)) and val == mean[idx]
    sub = np.arange(0, 700, 7)
    ref = orc.ei_over_hypers(comp, cand[sub], vals, hypers[:40])
    assert_ei_close(draws[sub, :40], ref)


def test_golden_stage_arrays(eng, golden_dir):
    g = _g(golden_dir, "ei_small_a.npz")
    eng.set_observations(g["comp"], g["vals"]); eng.set_candidates(g["cand"]); eng.set_hypers(g["hypers"])
    eng.factor()
    for h in range(g["hypers"].shape[0]):
        K, L, alpha = eng.get_factor(h)
        assert np.allclose(K, g["K"][h], rtol=1e-13, atol=1e-15)
        assert np.allclose(L @ L.T, g["K"][h], rtol=1e-12, atol=1e-14)
        assert np.allclose(eng.get_cross_cov(h), g["Kstar"][h], rtol=1e-12, atol=1e-15)


def test_golden_branin_c1(eng, golden_dir):
    g = _g(golden_dir, "branin_c1.npz")
    grid, values = g["grid"], g["values"]
    comp, cand, vals = grid[g["complete"]], grid[g["candidates"]], values[g["complete"]]
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, g["hypers"], want_draws=True)
    assert_ei_close(draws, g["ei"])
    assert int(g["candidates"][idx]) == int(g["job"])


def test_golden_persec(eng, golden_dir):
    g = _g(golden_dir, "ei_persec.npz")
    idx, _, mean, draws = eng.ei_per_sec_grid(g["comp"], g["vals"], g["log_durs"], g["cand"],
                                              g["hypers"], g["time_hypers"], want_draws=True)
    assert_ei_close(draws, g["ei"])
    assert idx == int(np.argmax(np.mean(g["ei"], axis=1)))
    # the reference's literal (early-return) behaviour = draw 0 only
    idx0, _, _, d0 = eng.ei_per_sec_grid(g["comp"], g["vals"], g["log_durs"], g["cand"],
                                         g["hypers"][:1], g["time_hypers"][:1], want_draws=True)
    assert_ei_close(d0[:, 0], g["literal"][:, 0])
    assert idx0 == int(np.argmax(np.mean(g["literal"], axis=1)))


# ---- oracle on seeded inputs, per stage ----------------------------------------
def test_plain_c_client_of_the_abi(eng, tmp_path):
    """tests/c/abi_client.c -- C99, gcc, no Python, no C++ -- drives libspx through include/spx.h and gets, bit for
    bit, what the ctypes binding gets: the boundary is a C ABI, not a Python extension."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "spearmint_amd")
    exe = str(tmp_path / "abi_client")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_client.c"), "-o", exe,
                           "-L" + libdir, "-lspx", "-Wl,-rpath," + libdir])
    comp, cand, vals, hypers = synthetic_problem(300, 7000, 6, 4, 80)
    with open(str(tmp_path / "in.bin"), "wb") as fh:
        np.array([300, 6, 7000, 4], dtype=np.int64).tofile(fh)
        for a in (comp, vals, cand, hypers):
            np.ascontiguousarray(a, dtype=np.float64).tofile(fh)
    out = subprocess.check_output([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(str(tmp_path / "out.bin"), "rb").read()
    best_idx = int(np.frombuffer(raw, dtype=np.int64, count=1)[0])
    rest = np.frombuffer(raw, dtype=np.float64, offset=8)
    best_val, mean, draws, lp = rest[0], rest[1:7001], rest[7001:7001 + 28000].reshape(7000, 4), rest[7001 + 28000:]
    idx, val, m, d = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    eng.set_hypers(hypers)
    assert best_idx == idx and best_val == val and np.array_equal(mean, m) and np.array_equal(draws, d)
    assert np.array_equal(lp, eng.gp_logprob())
    assert out.decode().startswith("best %d " % idx)


@pytest.mark.parametrize("N,M,D,H,seed", [
    (2, 5, 1, 1, 1), (3, 127, 2, 2, 2), (127, 128, 3, 3, 3), (128, 129, 4, 7, 4), (129, 1000, 5, 8, 5),
    (200, 513, 9, 9, 6), (300, 700, 16, 2, 7), (257, 300, 17, 2, 8), (256, 256, 33, 2, 9),
    (512, 2000, 64, 2, 10), (700, 900, 100, 1, 11), (1000, 3000, 32, 3, 12),
])
def test_oracle_stages(eng, N, M, D, H, seed):
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed, near=min(10, M))
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    eng.factor()
    from spearmint_amd.engine import FLAG_KEEP_MOMENTS
    eng.ei_run(FLAG_KEEP_MOMENTS)
    draws = eng.ei_draws()
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(draws, ref)
    assert eng.best()[0] == orc.choose(ref)
    st = {}
    orc.compute_ei(comp, cand, vals, hypers[0], stages=st)
    K, L, alpha = eng.get_factor(0)
    assert np.allclose(K, st["K"], rtol=1e-12, atol=1e-14)
    assert np.linalg.norm(L @ L.T - st["K"]) / np.linalg.norm(st["K"]) <= 1e-13
    assert np.allclose(alpha, st["alpha"], rtol=1e-8, atol=1e-10 * np.abs(st["alpha"]).max())
    m, v = eng.get_moments(0)
    assert np.allclose(m, st["func_m"], rtol=1e-9, atol=1e-10)
    assert np.allclose(v, st["func_v"], rtol=1e-9, atol=1e-13)     # rel <= 1e-9


def test_c2_full_size_every_value(eng):
    """BASELINE config 2 at full size: 256 obs, 20 000 candidates, 8-D, 10 draws."""
    comp, cand, vals, hypers = synthetic_problem(256, 20000, 8, 10, 2000)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(draws, ref)
    assert idx == orc.choose(ref)
    assert np.array_equal(mean, np.mean(draws, axis=1))


def test_c2_noiseless_hypers(eng):
    comp, cand, vals, hypers = synthetic_problem(256, 5000, 8, 4, 2001)
    hypers[:, 1] = 1e-3          # noise pinned as in GPEIChooser.py:270
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(draws, ref)
    assert idx == orc.choose(ref)


# ---- edge cases -------------------------------------------------------------------
def test_single_candidate_and_minimal_sizes(eng):
    comp, cand, vals, hypers = synthetic_problem(2, 1, 1, 1, 21, near=0)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert idx == 0 and draws.shape == (1, 1)
    assert_ei_close(draws, orc.ei_over_hypers(comp, cand, vals, hypers))


def test_ties_go_to_the_first_index(eng):
    comp, cand, vals, hypers = synthetic_problem(40, 600, 3, 3, 22)
    cand[:] = cand[17]                       # every candidate identical -> every EI identical
    idx, _, mean, _ = eng.ei_grid(comp, vals, cand, hypers)
    assert np.all(mean == mean[0]) and idx == 0
    cand = np.random.RandomState(1).rand(600, 3)
    cand[400] = cand[30]                     # a duplicated row
    idx2, _, mean2, _ = eng.ei_grid(comp, vals, cand, hypers)
    assert mean2[400] == mean2[30] and idx2 == int(np.argmax(mean2))


def test_nan_candidate_wins_like_numpy_argmax(eng):
    """NaN policy.  The reference cannot get this far with NaN inputs (scipy's
    check_finite raises ValueError in solve_triangular), so there is no oracle
    value to compare with; what is pinned is the documented device behaviour:
    a NaN input poisons only its own candidate, the other EI values are
    untouched, and the argmax follows numpy (first NaN wins)."""
    comp, cand, vals, hypers = synthetic_problem(30, 500, 2, 2, 23)
    clean = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)[3]
    cand[321, 0] = np.nan
    cand[400, 1] = np.nan
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    bad = np.zeros(500, bool); bad[[321, 400]] = True
    assert np.isnan(draws[bad]).all() and np.array_equal(draws[~bad], clean[~bad])
    assert idx == int(np.argmax(mean)) == 321 and np.isnan(val)
    with pytest.raises(ValueError):
        orc.ei_over_hypers(comp, cand, vals, hypers)


def test_not_positive_definite_raises_linalgerror(eng):
    comp, cand, vals, hypers = synthetic_problem(50, 200, 3, 3, 24)
    hypers[1, 2] = -1.0                      # negative amplitude -> K not PD at draw 1
    with pytest.raises(np.linalg.LinAlgError):
        eng.ei_grid(comp, vals, cand, hypers)
    draw, pivot = eng.not_pd_info()
    assert draw == 1 and pivot == 0
    with pytest.raises(np.linalg.LinAlgError):   # the oracle (scipy) raises at the same place
        orc.ei_over_hypers(comp, cand, vals, hypers)
    # the handle stays usable
    hypers[1, 2] = 1.0
    idx, _, _, _ = eng.ei_grid(comp, vals, cand, hypers)
    assert idx == orc.choose(orc.ei_over_hypers(comp, cand, vals, hypers))


def test_near_duplicate_observations(eng):
    comp, cand, vals, hypers = synthetic_problem(120, 800, 4, 3, 25)
    comp[60:] = comp[:60] + 1e-9             # pairs of (almost) identical observations
    vals[60:] = vals[:60]
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(draws, ref, rtol=1e-5)
    assert idx == orc.choose(ref)


def test_tail_values_keep_relative_accuracy(eng):
    """EI spans hundreds of decades; Phi must come from erfc in the tail."""
    comp, cand, vals, hypers = synthetic_problem(64, 2000, 2, 2, 26)
    vals = vals - 40.0 * (np.arange(64) == 5)       # one very low observation -> best far below the mean
    idx, _, _, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert ref[np.isfinite(ref)].min() < 1e-100
    assert_ei_close(draws, ref, rtol=1e-6)
    assert idx == orc.choose(ref)


def test_chunking_and_draw_grouping_do_not_change_bits(eng):
    comp, cand, vals, hypers = synthetic_problem(300, 3000, 6, 5, 27)
    base = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    try:
        eng.set_option("kstar_budget_bytes", 384 * 128 * 8)      # one 128-column tile, one draw at a time
        small = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    finally:
        eng.set_option("kstar_budget_bytes", 0)
    assert small[0] == base[0] and np.array_equal(small[3], base[3]) and np.array_equal(small[2], base[2])


def test_sharding_is_exact(eng):
    """Two contiguous shards + the argmax rule of the collective == the single-GPU answer, bit for bit."""
    comp, cand, vals, hypers = synthetic_problem(150, 2501, 5, 4, 28)
    idx, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    recs, parts = [], []
    eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.factor()
    for r in range(2):
        lo, hi = sd.shard_bounds(cand.shape[0], 2, r)
        eng.set_candidates(cand[lo:hi], index_base=lo)
        eng.ei_run()
        recs.append(list(eng.best()[::-1]))
        parts.append(eng.ei_draws())
    assert np.array_equal(np.vstack(parts), draws)
    assert sd.pick_best(recs) == (idx, val)


def test_gp_logprob(eng):
    comp, cand, vals, hypers = synthetic_problem(180, 10, 4, 5, 29)
    hypers[3, 2] = -2.0                      # not PD -> -inf, others finite
    eng.set_observations(comp, vals); eng.set_hypers(hypers)
    lp = eng.gp_logprob()
    for h in range(5):
        if h == 3:
            assert lp[h] == -np.inf
        else:
            ref = orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:])
            assert np.isclose(lp[h], ref, rtol=1e-10)


def test_timings_are_reported(eng):
    comp, cand, vals, hypers = synthetic_problem(256, 4000, 8, 3, 30)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    eng.set_option("timing", 1)
    eng.factor(); eng.ei_run()
    tm = eng.timings()
    eng.set_option("timing", 0)
    assert tm["predict_gemm"][1] >= 1 and tm["predict_gemm"][0] > 0
    # (the factorisation is one data-flow launch at this batch size -- stage "chol_diag" -- or one per block column)
    assert tm["chol_diag"][1] in (1, 256 // 64) and tm["factor_total"][1] == 1


# ---- the plugin API end to end on the GPU --------------------------------------------
def test_gpei_chooser_next_on_gpu_matches_reference(golden_dir, tmp_path):
    """examples/braninpy through GPEIChooser.next: same seeded hypers, same job
    as the reference's own run (tests/golden/branin_c1.npz)."""
    from spearmint_amd.chooser import GPEIChooser
    g = _g(golden_dir, "branin_c1.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=10")
    npr.seed(int(g["seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert job == int(g["job"])
    assert_ei_close(ch.last_overall_ei, g["ei"])


def test_opt_and_persec_chooser_next_on_gpu(golden_dir, tmp_path):
    from spearmint_amd.chooser import GPEIOptChooser, GPEIperSecChooser
    g = _g(golden_dir, "chooser_next.npz")
    args = (g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    ch = GPEIOptChooser.init(str(tmp_path / "a"), "mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0")
    os.makedirs(str(tmp_path / "a"), exist_ok=True)
    npr.seed(int(g["opt_seed"]))
    job = ch.next(*args)
    assert isinstance(job, tuple) == bool(int(g["opt_is_new"]))
    if isinstance(job, tuple):
        assert job[0] == int(g["opt_index"]) and np.allclose(job[1], g["opt_point"], atol=1e-5)
    os.makedirs(str(tmp_path / "b"), exist_ok=True)
    ps = GPEIperSecChooser.init(str(tmp_path / "b"), "mcmc_iters=3,burnin=4,grid_subset=4,ref_compat=1")
    npr.seed(int(g["ps_seed"]))
    job = ps.next(*args)
    if isinstance(job, tuple):
        assert job[0] == int(g["ps_index"]) and np.allclose(job[1], g["ps_point"], atol=1e-5)
    else:
        assert job == int(g["ps_index"])


def test_sampler_with_gpu_loglikelihood_matches_reference(golden_dir, tmp_path):
    """the slice sampler's log-likelihood on the GPU
    (spx_gp_logprob: K build + Cholesky + forward solve per call).  Same seeded
    RNG stream, so the hyper draws and the proposal must equal the reference's."""
    from spearmint_amd.chooser import GPEIChooser
    g = _g(golden_dir, "branin_c1.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=10,gpu_logprob=1")
    npr.seed(int(g["seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert job == int(g["job"])
    assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g["hypers"][-1], rtol=1e-6)
    assert_ei_close(ch.last_overall_ei, g["ei"], rtol=1e-5)


def test_gpu_loglikelihood_large_n_and_not_pd(eng):
    comp, cand, vals, hypers = synthetic_problem(700, 10, 6, 3, 31)
    eng.set_observations(comp, vals); eng.set_hypers(hypers)
    lp = eng.gp_logprob()
    for h in range(3):
        ref = orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:])
        assert np.isclose(lp[h], ref, rtol=1e-11)
    hypers[0, 2] = -1.0
    eng.set_hypers(hypers)
    with pytest.raises(np.linalg.LinAlgError):
        eng.gp_logprob(raise_not_pd=True)
    # a logprob call must not leave a stale factorisation behind for the EI path
    eng.set_candidates(cand)
    with pytest.raises(ValueError):
        eng.ei_run()


def test_gpu_loglikelihood_batch_size_does_not_change_bits(eng):
    """Up to 32 draws the log-likelihood path factors right-looking, beyond that left-looking with
    the right-hand side as an extra row block; both apply a tile's updates in the same order, so a
    draw's value must not depend on how many others share the batch (the speculative slice
    sampler relies on that)."""
    comp, cand, vals, hypers = synthetic_problem(330, 10, 5, 40, 37)
    hypers[7, 2] = -1.0                                  # one non-PD draw in the batch
    eng.set_observations(comp, vals)
    eng.set_hypers(hypers)
    big = eng.gp_logprob()                               # 40 draws: left-looking
    assert big[7] == -np.inf and np.all(np.isfinite(np.delete(big, 7)))
    for lo, hi in ((0, 1), (1, 6), (6, 38), (20, 40)):   # 1, 5, 32 and 20 draws: right-looking
        eng.set_hypers(hypers[lo:hi])
        assert np.array_equal(eng.gp_logprob(), big[lo:hi])
    for h in (0, 19, 39):
        ref = orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:])
        assert np.isclose(big[h], ref, rtol=1e-11)
    # ... nor on whether the factorisation is ONE data-flow launch (k_lean_flow, the default) or one launch per block column
    try:
        # (cu: one workgroup per CU or two, by default chosen from the size; cov: K(X,X) built tile by tile inside the
        # launch, the default, or by k_cov before it)
        # ... yl: with two per CU, a workgroup yields while its neighbour factors a diagonal block, or does not)
        for flow, cu, cov, yl in ((1, 1, 1, -1), (1, 0, 1, 1), (1, 0, 1, 0), (1, -1, 0, -1), (0, -1, -1, -1)):
            eng.set_option("lean_flow", flow)
            eng.set_option("lean_flow_cu", cu)
            eng.set_option("lean_flow_cov", cov)
            eng.set_option("lean_flow_yield", yl)
            for lo, hi in ((0, 1), (1, 6), (6, 38), (20, 40)):
                eng.set_hypers(hypers[lo:hi])
                assert np.array_equal(eng.gp_logprob(), big[lo:hi])
        # (what follows: the forms with one launch or two per block column)
        # ... nor on whether the trailing updates are applied one or two block columns at a time (option lean_lazy;
        # by default chosen from the batch's size)
        for lazy in (0, 1):
            eng.set_option("lean_lazy", lazy)
            for lo, hi in ((0, 1), (1, 6), (20, 40)):
                eng.set_hypers(hypers[lo:hi])
                assert np.array_equal(eng.gp_logprob(), big[lo:hi])
        eng.set_option("lean_lazy", -1)
        # ... nor on whether the panel solve of a block column is a launch of its own (option lean_ps=0) or runs inside the
        # update launch, its workgroups handed the inverse of the diagonal block row by row behind the pivots
        # (k_lean_step_ps)
        for ps in (0, 1):
            eng.set_option("lean_ps", ps)
            for lo, hi in ((0, 1), (1, 6), (20, 40)):
                eng.set_hypers(hypers[lo:hi])
                assert np.array_equal(eng.gp_logprob(), big[lo:hi])
    finally:
        eng.set_option("lean_lazy", -1)
        eng.set_option("lean_ps", -1)
        eng.set_option("lean_flow", -1)
        eng.set_option("lean_flow_cu", -1)
        eng.set_option("lean_flow_cov", -1)
        eng.set_option("lean_flow_yield", -1)


def _lean_form(eng, form):
    """The three forms of the log-likelihood factorisation: "flow" one data-flow launch (default), "ps" one launch per
    block column with the panel solve inside, "two" two launches per block column."""
    eng.set_option("lean_flow", {"flow": 1, "ps": 0, "two": 0, None: -1}[form])
    eng.set_option("lean_ps", {"flow": -1, "ps": 1, "two": 0, None: -1}[form])


def test_gpu_loglikelihood_in_launch_panel_solve_at_2048(eng):
    """The in-launch hand-off paths -- k_lean_flow (the whole factorisation one launch: tiles and diagonal inverses handed
    from workgroup to workgroup) and k_lean_step_ps (one launch per block column, the diagonal inverse handed to the panel
    workgroups) -- at the size they were built for, 32 block columns, the right-hand-side rows solved like any panel
    tile: equal to the two-launch path bit for bit, to LAPACK 1e-11; a 28-draw batch repeated 25 times (far more
    workgroups than the chip holds, every CU busy with other draws' tiles while workgroups wait) keeps returning the same
    bits; N not a multiple of 64; one and two block columns."""
    comp, cand, vals, hypers = synthetic_problem(2048, 10, 32, 3, 53)
    eng.set_observations(comp, vals)
    try:
        got = {}
        for name in ("flow", "ps", "two"):
            _lean_form(eng, name)
            eng.set_hypers(hypers)
            got[name] = eng.gp_logprob()
        assert np.array_equal(got["ps"], got["two"])
        assert np.array_equal(got["flow"], got["two"])
        ref = orc.gp_logprob(comp, vals, hypers[1, 0], hypers[1, 2], hypers[1, 1], hypers[1, 3:])
        assert np.isclose(got["flow"][1], ref, rtol=1e-11)
        comp3, _, vals3, hyp3 = synthetic_problem(1100, 10, 9, 28, 55)
        eng.set_observations(comp3, vals3)
        _lean_form(eng, "two")
        eng.set_hypers(hyp3); base = eng.gp_logprob()
        for name in ("ps", "flow"):
            _lean_form(eng, name)
            for _ in range(25):
                eng.set_hypers(hyp3)
                assert np.array_equal(eng.gp_logprob(), base), name
        lp = {}
        for name in ("flow", "ps"):
            _lean_form(eng, name)
            comp2, _, vals2, hyp2 = synthetic_problem(1000, 10, 7, 2, 54)      # N not a multiple of 64
            eng.set_observations(comp2, vals2)
            eng.set_hypers(hyp2)
            lp[name] = eng.gp_logprob()
            for n1, seed in ((70, 56), (40, 57)):                              # two block columns; one
                comp1, _, vals1, hyp1 = synthetic_problem(n1, 10, 3, 2, seed)
                eng.set_observations(comp1, vals1); eng.set_hypers(hyp1)
                lp1 = eng.gp_logprob()
                for h in range(2):
                    ref = orc.gp_logprob(comp1, vals1, hyp1[h, 0], hyp1[h, 2], hyp1[h, 1], hyp1[h, 3:])
                    assert np.isclose(lp1[h], ref, rtol=1e-11)
    finally:
        _lean_form(eng, None)
    assert np.array_equal(lp["flow"], lp["ps"])
    for h in range(2):
        ref = orc.gp_logprob(comp2, vals2, hyp2[h, 0], hyp2[h, 2], hyp2[h, 1], hyp2[h, 3:])
        assert np.isclose(lp["flow"][h], ref, rtol=1e-11)
        ref1 = orc.gp_logprob(comp1, vals1, hyp1[h, 0], hyp1[h, 2], hyp1[h, 1], hyp1[h, 3:])
        assert np.isclose(lp1[h], ref1, rtol=1e-11)


# ---- pending experiments ("next" row 2): fantasies on the GPU --------------------------
def test_golden_pending_fantasies(eng, golden_dir):
    from spearmint_amd import hostgp
    g = _g(golden_dir, "ei_pending.npz")
    comp, pend, cand, vals, hypers = g["comp"], g["pend"], g["cand"], g["vals"], g["hypers"]
    H, N, P, S = hypers.shape[0], len(comp), len(pend), g["randn"].shape[2]
    comp_pend = np.concatenate((comp, pend))
    eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    fant = np.empty((H, N + P, S)); bests = np.empty((H, S))
    for h in range(H):
        chol = eng.get_factor(h, want_K=False, want_alpha=False)[1]
        fant[h], bests[h] = hostgp.fantasize_pending(comp, pend, vals, hypers[h], chol[:N, :N], g["randn"][h])
    eng.set_fantasies(fant, bests)
    eng.ei_run()
    draws = eng.ei_draws()
    assert_ei_close(draws, g["ei"], rtol=1e-6)
    assert eng.best()[0] == int(np.argmax(np.mean(g["ei"], axis=1)))
    # clearing the fantasies gives back the plain path on the same resident data
    eng.set_fantasies(None, None)
    eng.ei_run()
    plain = orc.ei_over_hypers(comp_pend, cand, np.concatenate((vals, np.zeros(P))), hypers)
    assert_ei_close(eng.ei_draws(), plain, rtol=1e-6)


@pytest.mark.parametrize("N,P,M,D,H,S,seed", [(130, 5, 700, 4, 3, 100, 41), (300, 2, 1500, 9, 2, 8, 42),
                                              (64, 1, 300, 2, 2, 7, 43), (500, 7, 900, 6, 2, 128, 44)])
def test_fantasies_against_oracle(eng, N, P, M, D, H, S, seed):
    from spearmint_amd import hostgp
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
    rs = np.random.RandomState(seed)
    pend = rs.rand(P, D)
    comp_pend = np.concatenate((comp, pend))
    eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(P))))
    eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    fant = np.empty((H, N + P, S)); bests = np.empty((H, S)); ref = np.empty((M, H))
    for h in range(H):
        chol = eng.get_factor(h, want_K=False, want_alpha=False)[1]
        fant[h], bests[h] = hostgp.fantasize_pending(comp, pend, vals, hypers[h], chol[:N, :N], rs.randn(P, S))
        ref[:, h] = orc.compute_ei_fantasies(comp_pend, cand, hypers[h], fant[h], bests[h])
    eng.set_fantasies(fant, bests)
    eng.ei_run()
    assert_ei_close(eng.ei_draws(), ref, rtol=1e-6)
    assert eng.best()[0] == orc.choose(ref)
    assert np.array_equal(eng.ei_mean(), np.mean(eng.ei_draws(), axis=1))


def test_choosers_with_pending_on_gpu_match_reference(golden_dir, tmp_path):
    from spearmint_amd.chooser import GPEIChooser, GPEIOptChooser
    g = _g(golden_dir, "chooser_next_pending.npz")
    args = (g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=3,pending_samples=9")
    npr.seed(int(g["g_seed"]))
    assert ch.next(*args) == int(g["g_job"])
    assert_ei_close(ch.last_overall_ei, g["g_ei"], rtol=1e-6)
    os.makedirs(str(tmp_path / "o"), exist_ok=True)
    op = GPEIOptChooser.init(str(tmp_path / "o"), "mcmc_iters=3,burnin=4,grid_subset=3,pending_samples=8,use_multiprocessing=0")
    npr.seed(int(g["o_seed"]))
    job = op.next(*args)
    if int(g["o_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["o_index"]) and np.allclose(job[1], g["o_point"], atol=1e-5)
    else:
        assert job == int(g["o_index"])
    os.makedirs(str(tmp_path / "p"), exist_ok=True)
    from spearmint_amd.chooser import GPEIperSecChooser
    ps = GPEIperSecChooser.init(str(tmp_path / "p"), "mcmc_iters=2,burnin=3,grid_subset=3,pending_samples=6,ref_compat=1")
    npr.seed(int(g["p_seed"]))
    job = ps.next(*args)
    if int(g["p_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["p_index"]) and np.allclose(job[1], g["p_point"], atol=1e-5)
    else:
        assert job == int(g["p_index"])


def test_opt_chooser_with_gpu_refinement_matches_reference(golden_dir, tmp_path):
    from spearmint_amd.chooser import GPEIOptChooser
    g = _g(golden_dir, "chooser_next.npz")
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0,"
                                            "gpu_refine=1,gpu_logprob=1")
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert isinstance(job, tuple) and job[0] == int(g["opt_index"])
    assert np.allclose(job[1], g["opt_point"], atol=1e-5)


def test_chooser_with_ndev(golden_dir, tmp_path):
    from spearmint_amd.chooser import GPEIChooser
    g = _g(golden_dir, "branin_c1.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=10,ndev=1,device=0")
    npr.seed(int(g["seed"]))
    assert ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"]) == int(g["job"])


def test_two_stream_mode_is_bit_identical(eng):
    """Option streams=2 (K(X*,X) of item i+1 produced on a second stream while item i is
    consumed) must not change a single bit."""
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(260, 5000, 7, 5, 71, per_sec=True)
    try:
        eng.set_option("kstar_budget_bytes", 384 * 1024 * 8)     # several chunks and items
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        ap = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        eng.set_option("streams", 2)
        b = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        bp = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    finally:
        eng.set_option("streams", 1)
        eng.set_option("kstar_budget_bytes", 0)
    assert a[0] == b[0] and np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
    assert ap[0] == bp[0] and np.array_equal(ap[3], bp[3])


def test_two_stream_step_waits_for_the_pending_factorisation(eng):
    """spx_ei_step queues the factorisation on the main stream and, with option streams=2, the
    producer stream's scaling / K(X*,X) / duration-GP launches had no order against it -- they could read x / ls, the hyper
    table and alpha of the PREVIOUS problem.  The earlier two-stream test could not see that (the preceding one-stream
    call had left identical data behind).  Here every call changes observations AND hyper draws, and a two-stream handle
    is compared with a one-stream handle call by call."""
    two = Engine(0)
    try:
        two.set_option("streams", 2)
        for e in (eng, two):
            e.set_option("kstar_budget_bytes", 384 * 1024 * 8)      # several chunks and work items
        for rep in range(6):
            N = 260 + 130 * (rep % 3)                                  # (the padded size changes too)
            comp, cand, vals, hypers, log_durs, th = synthetic_problem(N, 5000, 7, 5, 710 + rep, per_sec=True)
            a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
            b = two.ei_grid(comp, vals, cand, hypers, want_draws=True)
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[3], b[3]), rep
            ap = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
            bp = two.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
            assert ap[0] == bp[0] and ap[1] == bp[1] and np.array_equal(ap[3], bp[3]), rep
    finally:
        two.close()
        eng.set_option("kstar_budget_bytes", 0)


@pytest.mark.parametrize("N", [1000, 900])      # (900: together with the skipped padding -- K* rows from 912 on unwritten)
def test_flat_covariance_launch_is_bit_identical(eng, tmp_path, N):
    """K(X*,X) launches of several residency rounds run as k_cov_flat (equal contiguous shares of the
    launch for a whole multiple of the chip's places) -- the same arithmetic per element: EI of every (candidate, draw) equals,
    bit for bit, the 3-D grid's (handle option cov_flat = 0; an environment variable until round 6)."""
    comp, cand, vals, hypers = synthetic_problem(N, 20000, 9, 4, 79)
    a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert eng.stat("last_step_skipped_padding") == (1 if N == 900 else 0)
    eng.set_option("cov_flat", 0)
    try:
        b = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    finally:
        eng.set_option("cov_flat", -1)
    assert b[0] == a[0] and b[1] == a[1] and np.array_equal(b[3], a[3])
    ref = orc.ei_over_hypers(comp, cand[:3000], vals, hypers)         # ... and both equal the oracle
    assert_ei_close(a[3][:3000], ref)


@pytest.mark.parametrize("N,D,H,expect_skip", [(129, 3, 4, True), (150, 8, 3, True), (200, 5, 2, True), (230, 4, 2, False),
                                                 (250, 4, 2, False), (257, 6, 3, True), (300, 8, 10, True), (400, 16, 2, True),
                                                 (900, 9, 2, True), (1000, 9, 2, False), (1300, 12, 2, True), (2000, 32, 2, False),
                                                 (256, 8, 2, False)])
def test_padding_of_the_observation_count_is_skipped_bit_identically(eng, N, D, H, expect_skip):
    """Round 5: N is padded to the GEMM's 128-row tiles; when at most six of the last row block's eight 16-row tiles hold
    observations (and that is at least 12 % of the pass: predict_gemm_padding_plan), that block goes to k_predict_gemm_tail (K steps and row tiles of the padding are not computed) and K(X*,X)
    leaves the pad rows unwritten (option gemm_partial, default on).  Every EI value, the moments, the mean and the winner
    equal the padded computation bit for bit -- and the oracle, as before."""
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(N, 4100, D, H, 900 + N, per_sec=True)
    try:
        eng.set_option("gemm_partial", 0)
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True, flags=2)
        ma = [eng.get_moments(h) for h in range(H)]
        assert eng.stat("last_step_skipped_padding") == 0
        ap = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        eng.set_option("gemm_partial", 1)
        # (stale rows in the K* staging buffer must not matter: poison it with another problem's values first)
        eng.ei_grid(comp[::-1].copy(), vals, 1.0 - cand, hypers)
        b = eng.ei_grid(comp, vals, cand, hypers, want_draws=True, flags=2)
        assert eng.stat("last_step_skipped_padding") == (1 if expect_skip else 0)
        mb = [eng.get_moments(h) for h in range(H)]
        bp = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
    finally:
        eng.set_option("gemm_partial", -1)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    for (m0, v0), (m1, v1) in zip(ma, mb):
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    assert ap[0] == bp[0] and np.array_equal(ap[3], bp[3])
    if N <= 1000:
        assert_ei_close(b[3], orc.ei_over_hypers(comp, cand, vals, hypers))
    else:
        assert_ei_close(b[3][:600], orc.ei_over_hypers(comp, cand[:600], vals, hypers), rtol=1e-6)


def test_padding_skip_with_pending_fantasies_and_chunks(eng):
    """... the same through the pending branch (the tail kernel's per-fantasy epilogue) and with several chunks and draw
    groups per pass."""
    comp, cand, vals, hypers = synthetic_problem(300, 6000, 5, 4, 907)
    rs = np.random.RandomState(3)
    fant = np.repeat(vals[None, :, None], 4, axis=0) + 0.05 * rs.randn(4, 300, 7)
    bests = fant.min(axis=1)
    out = []
    try:
        eng.set_option("kstar_budget_bytes", 384 * 1024 * 8)
        for on in (0, 1):
            eng.set_option("gemm_partial", on)
            eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.set_candidates(cand)
            eng.factor(); eng.set_fantasies(fant, bests); eng.ei_run()
            out.append((eng.best(), eng.ei_draws(), eng.stat("last_step_skipped_padding")))
    finally:
        eng.set_option("gemm_partial", -1)
        eng.set_option("kstar_budget_bytes", 0)
    assert out[0][2] == 0 and out[1][2] == 1
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    ref = np.stack([orc.compute_ei_fantasies(comp, cand[:500], hypers[h], fant[h], bests[h]) for h in range(4)], axis=1)
    assert_ei_close(out[1][1][:500], ref)


@pytest.mark.parametrize("N,D,H", [(5, 2, 1), (64, 3, 6), (65, 8, 3), (200, 5, 12), (700, 9, 6), (2048, 16, 4), (300, 4, 40)])
def test_merged_loglikelihood_prologue_is_bit_identical(eng, N, D, H):
    """Round 5: the log-likelihood call's two prologue launches (observation scaling; right-hand-side rows + flag clearing) are
    ONE (k_lean_prologue, option lean_merge).  Same values, bit for bit, as the two launches -- with a not-PD draw in the
    batch, after a call with other sizes (stale right-hand sides / flags), and equal to the oracle."""
    comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 4400 + N)
    if H >= 3:
        hypers[H // 2, 2] = -1.0                       # one draw that is not positive definite
    other = synthetic_problem(N + 37, 16, D, max(1, H - 1), 11)
    out = []
    try:
        for on in (0, 1):
            eng.set_option("lean_merge", on)
            eng.set_observations(other[0], other[2]); eng.set_hypers(other[3]); eng.gp_logprob()
            eng.set_observations(comp, vals); eng.set_hypers(hypers)
            out.append((eng.gp_logprob(), eng.not_pd_info()))
    finally:
        eng.set_option("lean_merge", -1)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    ref = np.array([orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:]) if hypers[h, 2] > 0 else -np.inf
                    for h in range(H)])
    ok = np.isfinite(ref)
    assert np.array_equal(np.isneginf(out[1][0]), ~ok)
    assert np.allclose(out[1][0][ok], ref[ok], rtol=1e-9, atol=1e-9 * N)


@pytest.mark.parametrize("N,D,H,covar", [(2, 1, 1, "Matern52"), (5, 2, 3, "Matern52"), (20, 2, 7, "Matern52"), (64, 3, 6, "Matern52"),
                                         (65, 8, 3, "Matern52"), (128, 8, 32, "Matern52"), (200, 5, 12, "Matern32"), (256, 8, 22, "Matern52"),
                                         (500, 17, 5, "ARDSE"), (700, 33, 6, "Matern52"), (1024, 16, 8, "Matern52"),
                                         (2048, 32, 4, "Matern52"), (300, 4, 40, "Matern52"), (130, 100, 3, "Matern52"), (90, 6, 4, "SE")])
def test_fused_loglikelihood_call_is_bit_identical(N, D, H, covar):
    """Round 6: spx_gp_logprob as ONE launch (option lean_one, default on) -- k_lean_flow's items scale
    the observation rows they need into LDS, generate the right-hand-side rows, and the last item of each draw reduces
    -sum log diag L - 0.5 |y|^2 into pinned host memory; no prologue launch, no reduction launch.  Same values bit for bit
    as the three-launch form (lean_one = 0): with a not-PD draw in the batch, after a call with other sizes, called
    repeatedly (the not-PD flags are left clean by the launch itself), for every covariance function, beyond 32 draws and
    for D > 64 (where the call keeps the three-launch form) -- and equal to the oracle."""
    from spearmint_amd.engine import Engine
    eng = Engine(0)
    try:
        eng.set_covar(covar)
        comp, cand, vals, hypers = synthetic_problem(N, 16, D, H, 5400 + N)
        if H >= 3:
            hypers[H // 2, 2] = -1.0                       # one draw that is not positive definite
        other = synthetic_problem(N + 37, 16, D, max(1, H - 1), 11)
        out = []
        for on in (0, 1):
            eng.set_option("lean_one", on)
            eng.set_observations(other[0], other[2]); eng.set_hypers(other[3]); eng.gp_logprob()
            eng.set_observations(comp, vals); eng.set_hypers(hypers)
            first = (eng.gp_logprob(), eng.not_pd_info())
            for _ in range(3):                             # ... and again: nothing stale
                eng.set_hypers(hypers)
                again = (eng.gp_logprob(), eng.not_pd_info())
                assert np.array_equal(again[0], first[0]) and again[1] == first[1]
            good = np.delete(hypers, H // 2, axis=0) if H >= 3 else hypers
            eng.set_hypers(good)                           # the failed draw's flag does not leak into the next call
            clean = eng.gp_logprob()
            assert np.isfinite(clean).all() and eng.not_pd_info()[0] < 0
            out.append((first, clean))
        assert np.array_equal(out[0][0][0], out[1][0][0]) and out[0][0][1] == out[1][0][1]
        assert np.array_equal(out[0][1], out[1][1])
        if covar == "Matern52":
            ref = np.array([orc.gp_logprob(comp, vals, hypers[h, 0], hypers[h, 2], hypers[h, 1], hypers[h, 3:]) if hypers[h, 2] > 0 else -np.inf
                            for h in range(H)])
            ok = np.isfinite(ref)
            assert np.array_equal(np.isneginf(out[1][0][0]), ~ok)
            assert np.allclose(out[1][0][0][ok], ref[ok], rtol=1e-9, atol=1e-9 * N)
    finally:
        eng.close()


def test_fused_loglikelihood_flags_survive_a_growing_batch():
    """The one-launch log-likelihood call keeps its not-PD flags clean by itself (no clearing launch): they are zeroed once per
    ALLOCATION.  A flag buffer that grows may be handed the address the smaller one had -- its new words are not zeros
    (round 6: a fresh engine whose first call has 1 row and whose second has 17 reported a spurious not-PD draw, one run in
    a few).  Fresh engines, batches that grow and shrink: every value equals the three-launch form's, no draw is reported."""
    from spearmint_amd.engine import Engine
    comp, cand, vals, hypers = synthetic_problem(40, 16, 3, 32, 97)
    ref = Engine(0)
    ref.set_option("lean_one", 0)
    ref.set_observations(comp, vals)
    want = {}
    for H in (1, 17, 3, 32, 8, 24):
        ref.set_hypers(hypers[:H])
        want[H] = ref.gp_logprob()
        assert np.isfinite(want[H]).all()
    ref.close()
    for trial in range(25):
        eng = Engine(0)
        eng.set_observations(comp, vals)
        for H in ((1, 17, 3, 32, 8, 24) if trial % 2 == 0 else (3, 8, 17, 24, 32, 1)):
            eng.set_hypers(hypers[:H])
            got = eng.gp_logprob()
            assert np.array_equal(got, want[H]), (trial, H)
            assert eng.not_pd_info()[0] < 0
        eng.close()


@pytest.mark.parametrize("name,N,D,args", [
    ("GPEIOptChooser", 40, 3, "mcmc_iters=4,burnin=6,grid_subset=3"),
    ("GPEIOptChooser", 300, 8, "mcmc_iters=5,burnin=5,grid_subset=4"),
    ("GPEIChooser", 130, 5, "mcmc_iters=6"),
    ("GPEIOptChooser", 200, 6, "mcmc_iters=4,burnin=4,grid_subset=3,noiseless=1"),
    ("GPEIperSecChooser", 120, 4, "mcmc_iters=3,burnin=4,grid_subset=3"),
])
def test_native_sampler_on_libspx_is_the_python_sampler_chain(tmp_path, name, N, D, args):
    """Round 6: `sampler=native` (spx_sample_hypers: the slice sampler's control flow, priors, speculation and numpy's
    random stream as C++ inside libspx; the default) and `sampler=python` (util.slice_sample_batched around
    Engine.gp_logprob, round 5) on the SAME log-likelihood kernels: the same hyper samples bit for bit, the same proposal,
    the same generator state afterwards -- at the measured depth (auto) and at explicit depths, for the three choosers,
    noiseless, and with a handle over three device slots."""
    import importlib
    from spearmint_amd.engine import MultiEngine
    mod = importlib.import_module("spearmint_amd.chooser." + name)
    rs = np.random.RandomState(N)
    G = 400
    grid = rs.rand(N + G, D)
    values = np.full(N + G, np.nan)
    values[:N] = np.sin(3 * grid[:N]).sum(axis=1) + 0.05 * rs.randn(N)
    durations = np.full(N + G, np.nan)
    durations[:N] = 1.0 + 3.0 * grid[:N, 0] + np.sin(5 * grid[:N, 1]) ** 2
    complete, candidates, pending = np.arange(N), np.arange(N, N + G), np.array([], dtype=int)
    got = {}
    for tag, extra, multi in (("python", "sampler=python,lookahead=6,follow=0:0", False), ("native", "sampler=native", False),
                              ("native-deep", "sampler=native,lookahead=8,follow=6:3", False),
                              ("native-multi", "sampler=native,lookahead=5,follow=3:2", True)):
        d = tmp_path / tag
        d.mkdir()
        ch = mod.init(str(d), args + ",use_multiprocessing=0," + extra)
        if multi:
            ch._eng = MultiEngine([0, 0, 0])
        npr.seed(77)
        job = ch.next(grid, values, durations, candidates, pending, complete)
        got[tag] = (job, [np.concatenate(([h[0], h[1], h[2]], h[3])) for h in getattr(ch, "hyper_samples", [])] or [ch.current_hyper_row()],
                    npr.get_state(), dict(ch.sampler_stats))
        ch.engine().close()
    ref = got["python"]
    assert ref[3]["calls"] == 0
    for tag in ("native", "native-deep", "native-multi"):
        g = got[tag]
        assert np.array_equal(np.array(g[1]), np.array(ref[1])), tag                      # the chain, bit for bit
        assert np.array_equal(g[2][1], ref[2][1]) and g[2][2:] == ref[2][2:], tag          # the generator, draw for draw
        if isinstance(ref[0], tuple):
            assert g[0][0] == ref[0][0] and np.array_equal(g[0][1], ref[0][1]), tag
        else:
            assert g[0] == ref[0], tag
        assert g[3]["calls"] > 0 and g[3]["moves"] > 0
    assert got["native-deep"][3]["free_moves"] > 0


def test_sampler_argument_errors(eng):
    from spearmint_amd.engine import SamplerCfg, RngState
    comp, cand, vals, hypers = synthetic_problem(50, 16, 4, 1, 5)
    eng.set_observations(comp, vals)
    cfg = SamplerCfg(D=5, n_iter=1, noiseless=0, check_mean=1, amp2_prior_on_sqrt=1, lookahead=4, follow_props=0, follow_hyps=0,
                     max_rows=32, noise_scale=0.1, amp2_scale=1.0, max_ls=2.0, vals_min=float(vals.min()), vals_max=float(vals.max()))
    with pytest.raises(ValueError):          # the observations have D = 4
        eng.sample_hypers(cfg, np.concatenate((hypers[0], [1.0])), np.zeros(12), rng_state=RngState.from_numpy())
    cfg.D = 4
    cfg.lookahead = 0
    with pytest.raises(ValueError):
        eng.sample_hypers(cfg, hypers[0].copy(), np.zeros(12), rng_state=RngState.from_numpy())
    cfg.lookahead = 4
    rows, st = eng.sample_hypers(cfg, hypers[0].copy(), np.zeros(12), rng_state=RngState.from_numpy())
    assert rows.shape == (1, 7) and st["iterations"] == 1 and st["moves"] == 5


def test_staged_host_copies_change_nothing(eng):
    """Round 6: callers' small host buffers travel through the handle's page-locked staging buffer (option stage_copies, default
    on) instead of the runtime's path for pageable memory: the same results bit for bit -- EI grid with mean and draws, the
    log-likelihood, the refinement objective, pending fantasies -- and buffers above the staging limit (8 MB) still arrive."""
    comp, cand, vals, hypers, log_durs, th = synthetic_problem(150, 9000, 6, 3, 41, per_sec=True)
    big = synthetic_problem(150, 200000, 6, 1, 42)[1]          # 9.6 MB of candidates: above the limit
    out = []
    for on in (0, 1):
        eng.set_option("stage_copies", on)
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        p = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, th, want_draws=True)
        eng.set_observations(comp, vals); eng.set_hypers(hypers)
        lp = eng.gp_logprob()
        eng.set_candidates(cand[:4000]); eng.factor()
        f, g = eng.ei_grad_batch(cand[:7])
        rs = np.random.RandomState(1)
        fant = vals[None, :, None] + 0.1 * rs.randn(3, 150, 5)
        eng.set_fantasies(fant, fant.min(axis=1)); eng.ei_run()
        fm = eng.ei_mean()
        b = eng.ei_grid(comp, vals, big, hypers[:1], want_mean=True)
        out.append((a, p, lp, f, g, fm, b))
    eng.set_option("stage_copies", -1)
    x, y = out
    assert x[0][0] == y[0][0] and np.array_equal(x[0][2], y[0][2]) and np.array_equal(x[0][3], y[0][3])
    assert x[1][0] == y[1][0] and np.array_equal(x[1][3], y[1][3])
    assert np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]) and np.array_equal(x[4], y[4]) and np.array_equal(x[5], y[5])
    assert x[6][0] == y[6][0] and np.array_equal(x[6][2], y[6][2])


def test_step_argument_errors_leave_nothing_queued(eng):
    """spx_ei_step checks its flags BEFORE it queues the factorisation, and any later error exit of a pending
    step returns with the streams idle and without an unchecked factor."""
    comp, cand, vals, hypers = synthetic_problem(300, 2000, 5, 4, 77)
    eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.set_candidates(cand)
    with pytest.raises(ValueError):
        eng.ei_step(FLAG_PER_SEC)                 # no time model
    with pytest.raises(ValueError):
        eng.get_factor(0)                         # nothing was factored by the refused call
    with pytest.raises(ValueError):
        eng.ei_step(8)                            # SPX_FLAG_TIME_ONLY alone
    eng.ei_step(0)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert eng.best()[0] == orc.choose(ref)


def test_injected_handoff_timeout_falls_back_warns_and_rearms():
    """a hand-off time-out of the data-flow factorisation (injected: one poll instead of 2^20) makes
    THAT call fall back to one launch per block column -- same bits, SPX_OK, a warning in spx_last_error -- keeps the handle
    there for `flow_rearm_after` clean factorisations and then returns it to the data-flow launch by itself."""
    comp, cand, vals, hypers = synthetic_problem(700, 3000, 6, 3, 78)
    e = Engine(0)
    try:
        e.set_observations(comp, vals); e.set_hypers(hypers); e.set_candidates(cand)
        lp = e.gp_logprob()
        e.ei_step(0)
        best, draws = e.best(), e.ei_draws()
        assert e.stat("flow_enabled") == 1 and e.stat("flow_fallbacks") == 0 and e.last_warning() is None
        e.set_option("flow_rearm_after", 3)
        e.set_option("flow_spin_limit", 1)         # the first hand-off that is not there yet gives up
        lp1 = e.gp_logprob()
        assert np.array_equal(lp1, lp)             # the repeated call's result
        assert e.stat("flow_fallbacks") == 1 and e.stat("flow_enabled") == 0
        w = e.last_warning()
        assert w and w.startswith("warning:") and "timed out" in w
        e.set_option("flow_spin_limit", 0)
        for k in range(3):                         # clean calls on the fallback form ...
            assert e.stat("flow_enabled") == 0
            assert np.array_equal(e.gp_logprob(), lp)
        assert e.stat("flow_enabled") == 1 and e.stat("flow_rearms") == 1       # ... and the handle is back
        assert np.array_equal(e.gp_logprob(), lp) and e.stat("flow_fallbacks") == 1
        # the same through the EI step (factorisation pending behind the pass) and spx_factor
        e.set_option("flow_spin_limit", 1)
        e.ei_step(0)
        assert e.stat("flow_fallbacks") == 2 and e.best() == best and np.array_equal(e.ei_draws(), draws)
        e.set_option("lean_flow", 1)               # asking again re-arms at once
        assert e.stat("flow_enabled") == 1
        e.factor()
        assert e.stat("flow_fallbacks") == 3
        e.set_option("flow_spin_limit", 0)
        e.set_option("lean_flow", 1)
        e.ei_step(0)
        assert e.stat("flow_fallbacks") == 3 and e.best() == best and np.array_equal(e.ei_draws(), draws)
    finally:
        e.close()


def test_gpei_chooser_ml2_hypers_on_gpu(golden_dir, tmp_path):
    """mcmc_iters=0 (ML-II point estimate, gp.py:181-292): same hypers and proposal as the reference.
    The optimiser drives the length scales to its lower bound exp(-10), an extreme the K build must survive."""
    from spearmint_amd.chooser import GPEIChooser
    g = _g(golden_dir, "chooser_next_ml2.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=0")
    npr.seed(5)
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert job == int(g["job"])
    assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g["hyper"], rtol=1e-6)
    comp, cand = g["grid"][g["complete"]], g["grid"][g["candidates"]]
    ref = orc.ei_over_hypers(comp, cand, g["values"][g["complete"]], g["hyper"][None, :])
    assert_ei_close(ch.last_overall_ei, ref, rtol=1e-6)


def test_noiseless_choosers_on_gpu_match_reference(golden_dir, tmp_path):
    """The same three seeded noiseless=1 runs as tests/test_host_logic.py, on the real engine."""
    from tests.test_host_logic import _noiseless_runs
    _noiseless_runs(golden_dir, tmp_path, lambda: None)


def test_restart_between_calls_on_gpu_matches_reference(golden_dir, tmp_path):
    """next(), a new chooser object on the same expt_dir, next(): the reference's two proposals, on the real engine."""
    from tests.test_host_logic import _two_call_runs
    _two_call_runs(golden_dir, tmp_path, lambda: None)


@pytest.mark.parametrize("extra", ["", ",gpu_logprob=1,gpu_refine=1"])
def test_whole_branin_runs_on_gpu_match_reference(golden_dir, tmp_path, extra):
    """The reference's 24 + 14 consecutive Branin proposals (tests/golden/branin_trajectory.npz), reproduced with
    the real engine -- by default (host log-likelihood / refinement below their size thresholds) and with the
    device log-likelihood and the batched device refinement forced on."""
    from tests.test_host_logic import _trajectory_runs
    _trajectory_runs(golden_dir, tmp_path, lambda: None, extra)


@pytest.mark.parametrize("npend,multi", [(0, False), (3, False), (0, True), (2, True)])
def test_opt_chooser_rescore_grid_does_not_change_the_proposal(tmp_path, npend, multi):
    """with rescore_grid=0 (default) the second EI pass of GPEIOptChooser keeps the pass-1 values of
    the grid rows and scores only the refined points; rescore_grid=1 runs the reference's literal second pass over
    [grid; refined].  They agree only while a candidate's EI is bit-independent of which other candidates (and which
    chunk / tile / device shard) share the call -- asserted here: same proposal and the same mean-EI vector, without
    and with pending jobs (fantasies), on one engine and on a multi-device handle (three engines on the one GPU)."""
    from spearmint_amd.chooser import GPEIOptChooser
    from spearmint_amd.engine import MultiEngine
    comp, cand, vals, _ = synthetic_problem(150, 2600, 4, 1, 77)
    rs = np.random.RandomState(5)
    pend = rs.rand(npend, 4)
    grid = np.vstack((comp, cand, pend))
    n, m = comp.shape[0], cand.shape[0]
    values = np.concatenate((vals, np.full(m + npend, np.nan)))
    durations = np.ones(grid.shape[0])
    complete, candidates, pending = np.arange(n), np.arange(n, n + m), np.arange(n + m, n + m + npend)
    out = []
    for rescore in (0, 1):
        d = tmp_path / ("r%d" % rescore)
        d.mkdir()
        ch = GPEIOptChooser.init(str(d), "mcmc_iters=3,burnin=1,grid_subset=4,use_multiprocessing=0,gpu_refine=1,"
                                         "gpu_logprob=1,rescore_grid=%d" % rescore)
        if multi:
            ch._eng = MultiEngine([0, 0, 0])
            ch._eng.set_covar(ch.covar)
            ch._eng.set_option("kstar_budget_bytes", 160 * 1024 * 8)      # several chunks per shard as well
        npr.seed(19)
        job = ch.next(grid, values, durations, candidates, pending, complete)
        out.append((job, np.array(ch.last_ei_mean)))
        ch.engine().close()
    (j0, m0), (j1, m1) = out
    # rescore_grid=1 leaves the means of [grid; refined]; rescore_grid=0 those of the refined points only
    assert np.array_equal(m0, m1[m:])
    if isinstance(j0, tuple):
        assert isinstance(j1, tuple) and j0[0] == j1[0] and np.array_equal(j0[1], j1[1])
    else:
        assert j0 == j1


def test_in_launch_handoff_under_concurrent_load(eng):
    """k_lean_flow and k_lean_step_ps hand tiles of L and the inverse of every diagonal block from workgroup to workgroup
    inside a launch (write-through stores, drained, then a flag; readers past their L1).  A stale read would show as different bits -- looked for here
    under UNEVEN load: a second engine keeps the GPU busy with EI grids from another host thread while random sizes and
    batch sizes go through the hand-off repeatedly; the two-launch path is the reference, bit for bit."""
    import threading
    from spearmint_amd.engine import Engine
    stop = []

    def noise():
        e2 = Engine(0)
        comp, cand, vals, hyp = synthetic_problem(700, 20000, 9, 5, 199)
        while not stop:
            e2.ei_grid(comp, vals, cand, hyp, want_mean=False)
        e2.close()
    th = threading.Thread(target=noise)
    th.start()
    rs = np.random.RandomState(17)
    try:
        for _ in range(40):
            N = int(rs.choice([260, 330, 700, 1000, 1500, 2048]))
            H = int(rs.randint(1, 25))
            comp, cand, vals, hyp = synthetic_problem(N, 10, int(rs.choice([2, 8, 32])), H, int(rs.randint(1 << 30)))
            if rs.rand() < 0.25:
                hyp[rs.randint(H), 2] = -1.0             # a non-PD draw: its workgroups must not hold the others up
            eng.set_observations(comp, vals)
            _lean_form(eng, "two")
            eng.set_hypers(hyp)
            ref = eng.gp_logprob()
            for name in ("flow", "ps"):
                _lean_form(eng, name)
                for rep in range(4):
                    eng.set_hypers(hyp)
                    assert np.array_equal(eng.gp_logprob(), ref), (name, N, H, rep)
        # no hand-off timed out on the way (a time-out would have switched the handle to one launch per block column, silently
        # but for stderr): the counter a caller can poll says so
        assert eng.stat("flow_fallbacks") == 0
    finally:
        stop.append(1)
        th.join()
        _lean_form(eng, None)


@pytest.mark.parametrize("N,D,H,per_sec", [(256, 8, 10, False), (1000, 7, 12, False), (70, 3, 5, False),
                                           (40, 2, 3, False), (520, 6, 20, True), (330, 5, 9, True), (200, 4, 70, False)])
def test_ei_path_factor_through_the_data_flow_launch(eng, N, D, H, per_sec):
    """spx_factor takes the log-likelihood path's one-launch factorisation (k_lean_flow: no right-hand-side rows, W from
    the tile-major factor; option ei_flow, default on -- the per-second cases here run 2 H = 40 and 18 draws through it,
    the last one 70).  The factor is the same bit for bit, so EI, its mean, the
    argmax, L and alpha must all be exactly what the left-looking launches give; a not-PD draw is reported alike."""
    prob = synthetic_problem(N, 3000, D, H, 77 + N, per_sec=per_sec)
    res = {}
    try:
        for flow in (0, 1):
            eng.set_option("ei_flow", flow)
            if per_sec:
                comp, cand, vals, hyp, ld, th = prob
                out = eng.ei_per_sec_grid(comp, vals, ld, cand, hyp, th, want_draws=True)
            else:
                comp, cand, vals, hyp = prob
                out = eng.ei_grid(comp, vals, cand, hyp, want_draws=True)
            fac = [eng.get_factor(d, want_K=False) for d in (0, H - 1)]
            res[flow] = (out, fac)
        (o0, f0), (o1, f1) = res[0], res[1]
        assert o0[0] == o1[0] and o0[1] == o1[1]
        assert np.array_equal(o0[2], o1[2]) and np.array_equal(o0[3], o1[3])
        for a, b in zip(f0, f1):
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        if not per_sec:
            bad = hyp.copy()
            bad[H // 2, 2] = -1.0                       # a not-PD draw: same report from both forms
            msgs = []
            for flow in (0, 1):
                eng.set_option("ei_flow", flow)
                with pytest.raises(Exception) as ei:
                    eng.ei_grid(comp, vals, cand, bad)
                msgs.append(str(ei.value))
            assert msgs[0] == msgs[1] and "positive definite" in msgs[0]
    finally:
        eng.set_option("ei_flow", -1)


# ---- the fused small-N EI pass (fused_kernels.hip: K* -> beta -> EI in one kernel, N <= 128) ------------------------
@pytest.mark.parametrize("N,M,D,H,seed", [(20, 1000, 2, 10, 901), (64, 5000, 8, 10, 902), (128, 20000, 8, 10, 903),
                                          (97, 3001, 5, 7, 904), (3, 200, 1, 2, 905), (128, 4099, 33, 3, 906),
                                          (17, 777, 16, 4, 907)])
def test_fused_small_n_equals_the_general_path_and_the_oracle(eng, N, M, D, H, seed):
    """N <= 128 (where Spearmint lives: tens of observations, S/main.py:83-85): the one-kernel EI pass gives, bit for
    bit, what the three-stage path (k_cov -> k_predict_gemm_tri -> k_ei_finalize) gives -- EI per draw, predictive
    moments, mean, winner -- and both match the oracle to the path's tolerance."""
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
    try:
        eng.set_option("ei_fused", 0)
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True, flags=2)   # SPX_FLAG_KEEP_MOMENTS
        ma = [eng.get_moments(h) for h in range(H)]
        eng.set_option("ei_fused", 1)
        b = eng.ei_grid(comp, vals, cand, hypers, want_draws=True, flags=2)
        mb = [eng.get_moments(h) for h in range(H)]
    finally:
        eng.set_option("ei_fused", -1)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    for (m0, v0), (m1, v1) in zip(ma, mb):
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(b[3], ref)
    assert b[0] == orc.choose(ref)


@pytest.mark.parametrize("kname", ["Matern32", "ARDSE"])
def test_fused_small_n_other_covariances_and_per_second(eng, kname):
    comp, cand, vals, hypers, ld, th = synthetic_problem(90, 6000, 6, 5, 911, per_sec=True)
    try:
        eng.set_covar(kname)
        eng.set_option("ei_fused", 0)
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        pa = eng.ei_per_sec_grid(comp, vals, ld, cand, hypers, th, want_draws=True)
        eng.set_option("ei_fused", 1)
        b = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        pb = eng.ei_per_sec_grid(comp, vals, ld, cand, hypers, th, want_draws=True)
    finally:
        eng.set_option("ei_fused", -1)
        eng.set_covar("Matern52")
    assert a[0] == b[0] and np.array_equal(a[3], b[3])
    assert pa[0] == pb[0] and np.array_equal(pa[3], pb[3]) and np.array_equal(pa[2], pb[2])
    with orc.covar(kname):
        ref = orc.ei_over_hypers(comp, cand, vals, hypers)
    assert_ei_close(b[3], ref)


def test_fused_small_n_nan_candidate_and_chunks(eng):
    """A NaN candidate wins like numpy's argmax in the fused pass too; chunked candidates (a small staging budget) do not
    change a bit."""
    comp, cand, vals, hypers = synthetic_problem(50, 9000, 4, 6, 921)
    one = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    try:
        eng.set_option("kstar_budget_bytes", 8 * 128 * 2048)      # chunks of 2048 candidates
        many = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    finally:
        eng.set_option("kstar_budget_bytes", 0)
    assert one[0] == many[0] and np.array_equal(one[3], many[3])
    bad = cand.copy(); bad[4321, 2] = np.nan
    idx, val, mean, draws = eng.ei_grid(comp, vals, bad, hypers, want_draws=True)
    assert idx == 4321 and np.isnan(val) and np.all(np.isnan(draws[4321]))
    keep = np.ones(len(cand), bool); keep[4321] = False
    assert np.array_equal(draws[keep], one[3][keep])


@pytest.mark.parametrize("N,M,D,H,per_sec", [(40, 3000, 3, 5, False), (128, 9000, 8, 10, True), (700, 5000, 6, 4, False)])
def test_ei_step_is_factor_plus_run_with_one_synchronisation(eng, N, M, D, H, per_sec):
    """spx_ei_step (what bench.py times and spx_ei_grid runs): the same results as spx_factor + spx_ei_run, the not-PD
    error reported as spx_factor reports it (LinAlgError, draw / pivot), and the handle fine afterwards."""
    from numpy.linalg import LinAlgError
    comp, cand, vals, hypers, ld, th = synthetic_problem(N, M, D, H, 940 + N, per_sec=True)
    fl = 1 if per_sec else 0
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    if per_sec:
        eng.set_time_model(ld, th)
    eng.factor(); eng.ei_run(fl)
    two = (eng.best(), eng.ei_draws(), eng.ei_mean())
    eng.set_hypers(hypers)
    if per_sec:
        eng.set_time_model(ld, th)
    eng.ei_step(fl)
    one = (eng.best(), eng.ei_draws(), eng.ei_mean())
    assert one[0] == two[0] and np.array_equal(one[1], two[1]) and np.array_equal(one[2], two[2])
    bad = hypers.copy(); bad[H // 2, 2] = -1.0                 # a negative amplitude: not positive definite
    eng.set_hypers(bad)
    if per_sec:
        eng.set_time_model(ld, th)                             # (spx_set_hypers drops the time model)
    with pytest.raises(LinAlgError):
        eng.ei_step(fl)
    assert eng.not_pd_info()[0] == H // 2
    with pytest.raises(ValueError):
        eng.best()                                             # no results after a failed step
    eng.set_hypers(hypers)
    if per_sec:
        eng.set_time_model(ld, th)
    eng.ei_step(fl)
    assert eng.best() == two[0] and np.array_equal(eng.ei_draws(), two[1])


def test_handle_statistics(eng):
    comp, cand, vals, hypers = synthetic_problem(40, 2000, 3, 4, 951)
    eng.ei_grid(comp, vals, cand, hypers)
    assert eng.stat("flow_enabled") == 1
    assert eng.stat("last_step_fused") == 1 and eng.stat("n_cu") >= 64 and eng.stat("flow_fallbacks") == 0
    comp, cand, vals, hypers = synthetic_problem(300, 2000, 3, 4, 952)
    eng.ei_grid(comp, vals, cand, hypers)
    assert eng.stat("last_step_fused") == 0
    with pytest.raises(ValueError):
        eng.stat("no_such_counter")


@pytest.mark.parametrize("N", [2, 33, 64, 65, 128, 129, 191, 192, 193, 300])
def test_loglikelihood_pads_to_64_not_128(eng, N):
    """The log-likelihood path pads the observations to whole 64 x 64 blocks (the EI path to the predict GEMM's 128): one
    diagonal block for N <= 64, three for 129..192.  Same values as the oracle, batch-size independent bits, and the EI
    path right after it (which pads to 128 again) is not disturbed."""
    comp, cand, vals, hypers = synthetic_problem(N, 300, 4, 5, 980 + N)
    eng.set_observations(comp, vals)
    eng.set_hypers(hypers); all5 = eng.gp_logprob()
    for h in range(5):
        eng.set_hypers(hypers[h:h + 1])
        assert eng.gp_logprob()[0] == all5[h]
    ref = np.array([orc.gp_logprob(comp, vals, h[0], h[2], h[1], h[3:]) for h in hypers])
    assert np.allclose(all5, ref, rtol=1e-10, atol=1e-9)
    got = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert_ei_close(got[3], orc.ei_over_hypers(comp, cand, vals, hypers))
    eng.set_hypers(hypers)
    assert np.array_equal(eng.gp_logprob(), all5)


@pytest.mark.parametrize("N,M,D,H,per_sec", [(64, 4000, 5, 6, False), (256, 20000, 8, 10, False), (128, 6000, 4, 5, True),
                                             (700, 9000, 6, 4, True)])
def test_step_overlap_does_not_change_bits(eng, N, M, D, H, per_sec):
    """spx_ei_step starts the candidate side of the pass (scaling, the first K(X*,X)) on the second stream beside the
    factorisation (option step_overlap, default on): same winner, means, per-draw EI and predicted durations as with
    everything on one stream, fused and general path, with and without a time model."""
    prob = synthetic_problem(N, M, D, H, 990 + N, per_sec=per_sec)
    comp, cand, vals, hypers = prob[:4]
    fl = 3 if per_sec else 2            # (PER_SEC |) KEEP_MOMENTS
    out = {}
    try:
        for ov in (0, 1):
            eng.set_option("step_overlap", ov)
            eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
            if per_sec:
                eng.set_time_model(prob[4], prob[5])
            for _ in range(3):          # repeated: the two streams have no fixed relative timing
                eng.ei_step(fl)
            out[ov] = (eng.best(), eng.ei_mean(), eng.ei_draws(), eng.get_moments(H - 1),
                       eng.get_time_mean(0) if per_sec else None)
    finally:
        eng.set_option("step_overlap", -1)
    a, b = out[0], out[1]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1])
    if per_sec:
        assert np.array_equal(a[4], b[4])


@pytest.mark.parametrize("N,P,D,H,flow", [(130, 3, 4, 3, 1), (300, 5, 6, 2, 1), (70, 2, 3, 2, 0), (1000, 4, 8, 2, 1)])
def test_factor_rows_are_the_rows_of_the_factor(eng, N, P, D, H, flow):
    """spx_get_factor_rows (bottom P rows of L and gamma: what the pending branch brings home instead of the whole factor)
    against spx_get_factor and the oracle, tile-major (ei_flow=1) and row-major factor storage."""
    import scipy.linalg as spla
    comp, cand, vals, hypers = synthetic_problem(N + P, 200, D, H, 995 + N)
    try:
        eng.set_option("ei_flow", flow)
        eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
        for h in range(H):
            L = eng.get_factor(h, want_K=False, want_alpha=False)[1]
            rows, gam = eng.get_factor_rows(h, N, P)
            assert np.array_equal(rows, L[N:, :])
            mid, _ = eng.get_factor_rows(h, 60, 9, want_gamma=False)       # a range that straddles a 64-row block
            assert np.array_equal(mid, L[60:69, :])
            ref = spla.solve_triangular(orc.posterior(comp, vals, hypers[h])[1], vals - hypers[h][0], lower=True)
            assert np.allclose(gam, ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
        with pytest.raises(ValueError):
            eng.get_factor_rows(0, N, P + 1)
    finally:
        eng.set_option("ei_flow", -1)


def test_time_only_pass_gives_the_predicted_durations(eng):
    """SPX_FLAG_TIME_ONLY (with PER_SEC | KEEP_MOMENTS): the predicted durations of the full per-second pass, bit for bit,
    without the EI work; EI getters refuse afterwards; a change of inputs invalidates them."""
    comp, cand, vals, hypers, ld, th = synthetic_problem(150, 5000, 5, 4, 997, per_sec=True)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.set_time_model(ld, th)
    eng.ei_step(3)
    full = [eng.get_time_mean(h) for h in range(4)]
    eng.set_hypers(hypers); eng.set_time_model(ld, th)
    eng.ei_step(3 | 8)
    only = [eng.get_time_mean(h) for h in range(4)]
    assert all(np.array_equal(a, b) for a, b in zip(full, only))
    with pytest.raises(ValueError):
        eng.best()
    with pytest.raises(ValueError):
        eng.ei_step(8)                      # needs PER_SEC | KEEP_MOMENTS
    eng.set_candidates(cand[:100])
    with pytest.raises(ValueError):
        eng.get_time_mean(0)

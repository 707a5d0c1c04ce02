"""Host-side logic of the choosers on a CPU-only box: argument parsing, the
slice sampler's RNG order, state pickles, and -- with the test-only
OracleEngine injected in place of the GPU -- that a seeded .next() proposes
exactly what the REFERENCE's own .next() proposed (tests/golden, produced by
running the reference)."""
import os
import pickle

import numpy as np
import numpy.random as npr
import pytest

from spearmint_amd import hostgp, util
from spearmint_amd.chooser import GPEIChooser, GPEIOptChooser, GPEIperSecChooser
from tests.helpers import OracleEngine, np_mean_device_order


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_unpack_args():
    assert util.unpack_args("mcmc_iters=20, noiseless = 1") == {"mcmc_iters": "20", "noiseless": "1"}
    assert util.unpack_args("") == {}
    assert util.unpack_args("x") == {}


def test_slice_sampler_rng_order_matches_reference(golden_dir):
    g = _g(golden_dir, "slice_sampler.npz")
    comp, vals = g["comp"], g["vals"]

    def lp_ls(ls):
        if np.any(ls < 0) or np.any(ls > 2):
            return -np.inf
        return hostgp.data_logprob(comp, vals, 0.1, 1.3, 1e-3, ls)

    assert np.isclose(lp_ls(np.ones(3)), float(g["lp_at_ones"]), rtol=1e-12)
    npr.seed(77)
    x = g["compwise"][0]
    for k in range(1, g["compwise"].shape[0]):
        x = util.slice_sample(x, lp_ls, compwise=True)
        assert np.allclose(x, g["compwise"][k], rtol=1e-9)
    npr.seed(78)
    y = g["joint"][0]
    for k in range(1, g["joint"].shape[0]):
        y = util.slice_sample(y, lambda v: -0.5 * np.sum((v - 0.2) ** 2) / 0.3, compwise=False)
        assert np.allclose(y, g["joint"][k], rtol=1e-9)


def test_early_out_touches_nothing(tmp_path):
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=3")
    grid = np.random.rand(10, 2)
    assert ch.next(grid, np.zeros(10), np.zeros(10), np.arange(1, 10), np.array([], int), np.array([0])) == 1
    assert ch._eng is None and ch.D == -1


def test_gpei_next_matches_reference_c1(golden_dir, tmp_path):
    """BASELINE config 1 through the plugin API: same hypers, same proposal."""
    g = _g(golden_dir, "branin_c1.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=10")
    eng = OracleEngine(); ch._eng = eng
    npr.seed(int(g["seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert isinstance(job, int) and job == int(g["job"])
    assert eng.calls == [("ei_grid", 980, 10)]
    assert np.allclose(ch.last_overall_ei, g["ei"], rtol=1e-9, atol=1e-300)
    # state pickle: same keys as the reference's (GPEIChooser.py:70-76)
    ch.save_state()
    st = pickle.load(open(ch.state_pkl, "rb"))
    assert sorted(st) == ["amp2", "dims", "ls", "mean", "noise"]
    assert np.allclose(np.concatenate(([st["mean"], st["noise"], st["amp2"]], st["ls"])), g["hypers"][-1], rtol=1e-9)
    assert os.path.basename(ch.state_pkl).endswith("GPEIChooser.pkl")


def test_gpei_restart_reloads_state(golden_dir, tmp_path):
    g = _g(golden_dir, "branin_c1.npz")
    a = GPEIChooser.init(str(tmp_path), "mcmc_iters=2")
    a._eng = OracleEngine(); npr.seed(3)
    a.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    a.save_state()
    b = GPEIChooser.init(str(tmp_path), "mcmc_iters=2")
    b._real_init(2, g["values"][g["complete"]])
    assert np.array_equal(b.ls, a.ls) and b.amp2 == a.amp2 and b.noise == a.noise and b.mean == a.mean


def test_opt_next_matches_reference(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next.npz")
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0")
    eng = OracleEngine(); ch._eng = eng
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert np.allclose(ch.hyper_rows(), g["opt_hypers"], rtol=1e-9)
    assert [c[1] for c in eng.calls] == [len(g["candidates"]) + 10, 5]     # second pass: the refined points only
    if int(g["opt_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["opt_index"])
        assert np.allclose(job[1], g["opt_point"], atol=1e-6)
    else:
        assert job == int(g["opt_index"])
    assert os.path.exists(ch.stats_file) and os.path.exists(ch.state_pkl)
    assert "lsdata" in ch.generate_stats_html()
    # pickling the chooser (multiprocessing / fork safety) drops the engine handle
    clone = pickle.loads(pickle.dumps(ch))
    assert clone._eng is None


def test_persec_next_matches_reference_bug_compatible(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next.npz")
    ch = GPEIperSecChooser.init(str(tmp_path), "mcmc_iters=3,burnin=4,grid_subset=4,ref_compat=1")
    ch._eng = OracleEngine()
    npr.seed(int(g["ps_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert np.allclose(ch._rows(ch.hyper_samples), g["ps_hypers"], rtol=1e-9)
    assert np.allclose(ch._rows(ch.time_hyper_samples), g["ps_time_hypers"], rtol=1e-9)  # never cleared (:199)
    if int(g["ps_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["ps_index"])
        assert np.allclose(job[1], g["ps_point"], atol=1e-6)
    else:
        assert job == int(g["ps_index"])


def test_persec_default_semantics_evaluate_all_draws(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next.npz")
    ch = GPEIperSecChooser.init(str(tmp_path), "mcmc_iters=3,burnin=2,grid_subset=3")
    eng = OracleEngine(); ch._eng = eng
    npr.seed(1)
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert all(c[2] == 3 for c in eng.calls) and len(ch.time_hyper_samples) == 3
    assert isinstance(job, (int, tuple))


def test_gpei_next_with_pending_matches_reference(golden_dir, tmp_path):
    """Fantasy branch through the plugin API (three jobs still running)."""
    g = _g(golden_dir, "chooser_next_pending.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=3,pending_samples=9")
    eng = OracleEngine(); ch._eng = eng
    npr.seed(int(g["g_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert job == int(g["g_job"])
    assert eng.calls == [("ei_run", len(g["candidates"]), 3, 9)]
    assert np.allclose(ch.last_overall_ei, g["g_ei"], rtol=1e-8, atol=1e-300)


def test_opt_next_with_pending_matches_reference(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next_pending.npz")
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=3,burnin=4,grid_subset=3,pending_samples=8,use_multiprocessing=0")
    ch._eng = OracleEngine()
    npr.seed(int(g["o_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    if int(g["o_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["o_index"])
        assert np.allclose(job[1], g["o_point"], atol=1e-6)
    else:
        assert job == int(g["o_index"])


def test_persec_next_with_pending_matches_reference(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next_pending.npz")
    ch = GPEIperSecChooser.init(str(tmp_path), "mcmc_iters=2,burnin=3,grid_subset=3,pending_samples=6,ref_compat=1")
    ch._eng = OracleEngine()
    npr.seed(int(g["p_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    if int(g["p_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["p_index"])
        assert np.allclose(job[1], g["p_point"], atol=1e-6)
    else:
        assert job == int(g["p_index"])


def test_unknown_covariance_rejected(tmp_path):
    with pytest.raises(AttributeError):            # the reference: getattr(gp, covar)
        GPEIChooser.init(str(tmp_path), "covar=Periodic")


def _covar_runs(golden_dir, tmp_path, kname, make_engine, extra=""):
    g = _g(golden_dir, "covar_%s.npz" % kname)
    args = (g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    specs = [("g", GPEIChooser, "mcmc_iters=3")]
    if kname != "SE":
        specs += [("o", GPEIOptChooser, "mcmc_iters=3,burnin=4,grid_subset=3,use_multiprocessing=0"),
                  ("p", GPEIperSecChooser, "mcmc_iters=2,burnin=3,grid_subset=3,ref_compat=1")]
    for tag, mod, arg in specs:
        d = tmp_path / (kname + tag)
        d.mkdir()
        ch = mod.init(str(d), arg + ",covar=" + kname + extra)
        eng = make_engine(kname)
        if eng is not None:
            ch._eng = eng
        npr.seed(int(g[tag + "_seed"]))
        job = ch.next(*args)
        assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g[tag + "_hyper"], rtol=1e-6)
        if int(g[tag + "_is_new"]):
            assert isinstance(job, tuple) and job[0] == int(g[tag + "_index"])
            assert np.allclose(job[1], g[tag + "_point"], atol=1e-5)
        else:
            assert job == int(g[tag + "_index"])
    # mcmc_iters=0: the ML-II point estimate under this covariance
    d = tmp_path / (kname + "ml2")
    d.mkdir()
    ch = GPEIChooser.init(str(d), "mcmc_iters=0,covar=" + kname + extra)
    eng = make_engine(kname)
    if eng is not None:
        ch._eng = eng
    npr.seed(5)
    assert ch.next(*args) == int(g["ml2_job"])
    assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g["ml2_hyper"], rtol=1e-6)
    if kname == "SE":      # no grad_SE in gp.py: the reference's refinement raises, and so does ours
        for mod, arg in ((GPEIOptChooser, "mcmc_iters=2,burnin=2,grid_subset=2,use_multiprocessing=0"),
                         (GPEIperSecChooser, "mcmc_iters=2,burnin=2,grid_subset=2")):
            d = tmp_path / (kname + mod.__name__.split(".")[-1])
            d.mkdir()
            ch = mod.init(str(d), arg + ",covar=SE" + extra)
            eng = make_engine(kname)
            if eng is not None:
                ch._eng = eng
            npr.seed(1)
            with pytest.raises(AttributeError):
                ch.next(*args)


@pytest.mark.parametrize("kname", ["Matern32", "ARDSE", "SE"])
def test_choosers_with_other_covariances_match_reference(golden_dir, tmp_path, kname):
    """covar=Matern32 / ARDSE / SE: the reference's own seeded next() calls (hyper draws through the slice
    sampler, EI grid, refinement) reproduced by our choosers."""
    _covar_runs(golden_dir, tmp_path, kname, lambda k: OracleEngine(k))


@pytest.mark.parametrize("H", [1, 3, 7, 8, 9, 10, 16, 20, 31, 128, 129, 300])
def test_device_mean_order_is_numpys(H):
    rs = np.random.RandomState(H)
    a = rs.rand(50, H) * 10.0 ** rs.randint(-20, 3, size=(50, H))
    want = np.mean(a, axis=1)
    got = np.array([np_mean_device_order(r) for r in a])
    assert np.array_equal(want, got)


def test_device_pairwise_sum_source_is_numpys(tmp_path):
    """The summation the kernels run (spearmint_amd/csrc/np_sum.h: numpy's pairwise order restated WITHOUT recursion, so
    no kernel needs a private segment) compiled for the host from that very header and compared with numpy bit for
    bit: every length up to 1200, the power-of-two edges, strided (column) access as k_mean_over_draws reads it."""
    import ctypes
    import subprocess
    src = tmp_path / "pw.cpp"
    src.write_text('#define SPX_HD static inline\n#include "np_sum.h"\n'
                   'extern "C" double pw(const double* a, long stride, int n) { return np_pairwise(a, stride, n); }\n')
    so = str(tmp_path / "libpw.so")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spearmint_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I" + inc, str(src), "-o", so])
    lib = ctypes.CDLL(so)
    lib.pw.restype = ctypes.c_double
    lib.pw.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    rs = np.random.RandomState(5)
    for n in list(range(1, 1201)) + [2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192]:
        a = rs.randn(n) * 10.0 ** rs.uniform(-6, 6, n)
        assert lib.pw(a.ctypes.data, 1, n) == np.sum(a), n
    for H in (20, 129, 300, 1000, 4096):
        m = np.ascontiguousarray((rs.rand(7, H) * 10.0 ** rs.randint(-20, 3, size=(7, H))))
        t = np.ascontiguousarray(m.T)                      # [H][M]: the device layout, stride M
        got = np.array([(0.0 + lib.pw(t.ctypes.data + 8 * c, 7, H)) / H for c in range(7)])
        assert np.array_equal(got, np.mean(m, axis=1)), H


def test_speculative_sampler_is_the_same_markov_chain(golden_dir):
    """util.slice_sample_batched (proposals evaluated in speculative batches, RNG rewound)
    must reproduce util.slice_sample bit for bit -- values and RNG state -- also when the
    log-probability raises beyond the point the sequential sampler would have reached."""
    g = _g(golden_dir, "slice_sampler.npz")
    comp, vals = g["comp"], g["vals"]

    def lp_ls(ls):
        if np.any(ls < 0) or np.any(ls > 2):
            return -np.inf
        return hostgp.data_logprob(comp, vals, 0.1, 1.3, 1e-3, ls)

    def many(xs):
        vals_, errs = [], []
        for x in xs:
            try:
                vals_.append(lp_ls(x)); errs.append(None)
            except Exception as e:          # surfaces only if the sampler consumes this entry
                vals_.append(np.nan); errs.append(e)
        return util._LazyValues(vals_, errs)

    for compwise, sigma, la in [(True, 1.0, 4), (True, 0.2, 2), (False, 1.0, 3), (False, 0.5, 1)]:
        npr.seed(11); a = [np.ones(3)]
        for _ in range(40):
            a.append(util.slice_sample(a[-1], lp_ls, sigma=sigma, compwise=compwise))
        sa = npr.get_state()
        npr.seed(11); b = [np.ones(3)]
        for _ in range(40):
            b.append(util.slice_sample_batched(b[-1], many, sigma=sigma, compwise=compwise, lookahead=la))
        sb = npr.get_state()
        assert np.array_equal(np.array(a), np.array(b))
        assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]
    # With the priors' support known a priori (`admissible`) and a record of how the bracket's ends behaved in earlier
    # moves (`history`), the sampler also speculates on an end stepping out to the edge of the support: still the same
    # chain and RNG state -- and fewer batches where that is what the moves do (sigma well above the support's width)
    batches = {}

    def counting(xs):
        batches[counting.tag] = batches.get(counting.tag, 0) + 1
        return many(xs)
    for compwise, sigma, la in [(True, 1.0, 6), (True, 1.0, 2), (True, 0.4, 4), (False, 1.0, 6), (True, 3.0, 6), (True, 3.0, 3)]:
        npr.seed(12); a = [np.ones(3)]
        for _ in range(40):
            a.append(util.slice_sample(a[-1], lp_ls, sigma=sigma, compwise=compwise))
        sa = npr.get_state()
        for with_adm in (False, True):
            counting.tag = (sigma, la, with_adm)
            for attr in ("admissible", "history"):
                if hasattr(counting, attr):
                    delattr(counting, attr)
            if with_adm:
                counting.admissible = lambda x: not (np.any(x < 0) or np.any(x > 2))
                counting.history = {}
            npr.seed(12); b = [np.ones(3)]
            for _ in range(40):
                b.append(util.slice_sample_batched(b[-1], counting, sigma=sigma, compwise=compwise, lookahead=la))
            sb = npr.get_state()
            assert np.array_equal(np.array(a), np.array(b))
            assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]
    # a log-probability that pushes against the upper edge of the support (what the length scales of a 32-D problem
    # do under their top-hat prior): the upper end of the bracket steps out to the edge in most moves, the planner
    # learns it, and the chain -- unchanged -- needs fewer batches
    def lp_push(x):
        return -np.inf if (np.any(x < 0) or np.any(x > 2)) else 4.0 * float(np.sum(x))

    def many_push(xs):
        many_push.n += 1
        return util._LazyValues([lp_push(x) for x in xs], [None] * len(xs))
    npr.seed(5); a = [np.ones(4)]
    for _ in range(60):
        a.append(util.slice_sample(a[-1], lp_push, compwise=True))
    sa = npr.get_state()
    count = {}
    for with_adm in (False, True):
        for attr in ("admissible", "history"):
            if hasattr(many_push, attr):
                delattr(many_push, attr)
        if with_adm:
            many_push.admissible = lambda x: not (np.any(x < 0) or np.any(x > 2))
            many_push.history = {}
        many_push.n = 0
        npr.seed(5); b = [np.ones(4)]
        for _ in range(60):
            b.append(util.slice_sample_batched(b[-1], many_push, compwise=True, lookahead=6))
        assert np.array_equal(np.array(a), np.array(b)) and np.array_equal(sa[1], npr.get_state()[1])
        count[with_adm] = many_push.n
    assert count[True] < 0.85 * count[False], count
    # golden trace of the REFERENCE's sampler through the batched code path
    npr.seed(77)
    x = g["compwise"][0]
    for k in range(1, g["compwise"].shape[0]):
        x = util.slice_sample_batched(x, many, compwise=True)
        assert np.allclose(x, g["compwise"][k], rtol=1e-9)


def test_cross_move_speculation_is_the_same_markov_chain_with_fewer_calls(golden_dir):
    """util.slice_sample_batched(follow=(P, H)): the batch of one coordinate move also carries what the NEXT coordinate's
    move will ask for if this one accepts its 1st ... H-th proposal (edges, ladder, first P proposals).  The evaluator is
    lazy (`submit`): a move whose points are all known makes no call.  Same chain, same RNG state as util.slice_sample,
    at every depth -- and fewer evaluator calls."""
    g = _g(golden_dir, "slice_sampler.npz")
    comp, vals = g["comp"], g["vals"]

    def lp_ls(ls):
        if np.any(ls < 0) or np.any(ls > 2):
            return -np.inf
        return hostgp.data_logprob(comp, vals, 0.1, 1.3, 1e-3, ls)

    def lazy(lp, count):
        memo = {}

        def submit(xs, extras=()):
            values, errors, missing = [None] * len(xs), [None] * len(xs), set()
            for k, x in enumerate(xs):
                key = np.asarray(x, dtype=float).tobytes()
                if key in memo:
                    values[k] = memo[key]
                else:
                    missing.add(k)
            if not missing:
                return util._LazyValues(values, errors)

            def fill(lv):
                count[0] += 1
                count[1] += len(missing)
                for k in missing:
                    values[k] = memo[np.asarray(xs[k], dtype=float).tobytes()] = lp(xs[k])
                for x in extras:
                    key = np.asarray(x, dtype=float).tobytes()
                    if key not in memo:
                        memo[key] = lp(x)
                        count[1] += 1
                lv.missing = ()
            return util._LazyValues(values, errors, missing, fill)

        def many(xs):
            lv = submit(xs)
            if lv.missing:
                lv.fill(lv)
            return lv
        many.submit = submit
        many.admissible = lambda x: not (np.any(x < 0) or np.any(x > 2))
        many.history = {}
        return many

    for sigma, la in ((1.0, 6), (1.0, 8), (0.4, 4), (3.0, 6)):
        npr.seed(21); a = [np.ones(3)]
        for _ in range(40):
            a.append(util.slice_sample(a[-1], lp_ls, sigma=sigma, compwise=True))
        sa = npr.get_state()
        calls = {}
        for follow in ((0, 0), (4, 2), (3, 1), (6, 3)):
            count = [0, 0]
            ev = lazy(lp_ls, count)
            npr.seed(21); b = [np.ones(3)]
            for _ in range(40):
                b.append(util.slice_sample_batched(b[-1], ev, sigma=sigma, compwise=True, lookahead=la, follow=follow))
            sb = npr.get_state()
            assert np.array_equal(np.array(a), np.array(b)), (sigma, la, follow)
            assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]
            calls[follow] = count[0]
        assert calls[(4, 2)] < 0.9 * calls[(0, 0)] and calls[(6, 3)] < 0.8 * calls[(0, 0)], calls
        assert calls[(6, 3)] <= calls[(3, 1)] <= calls[(0, 0)], calls
    # the joint (random-direction) move ignores `follow`: nothing to follow into
    ev = lazy(lp_ls, [0, 0])
    npr.seed(4); x1 = util.slice_sample(np.ones(3), lp_ls, compwise=False); s1 = npr.get_state()
    npr.seed(4); x2 = util.slice_sample_batched(np.ones(3), ev, compwise=False, lookahead=6, follow=(4, 2)); s2 = npr.get_state()
    assert np.array_equal(x1, x2) and np.array_equal(s1[1], s2[1]) and s1[2] == s2[2]


def test_chooser_adapter_serves_a_following_move_from_the_memo(tmp_path):
    """chooser._speculative_logprob().submit: a batch with `extras` evaluates them in the same engine call (up to 32 rows);
    a later batch made only of known rows makes no call; rows a priori outside the support never reach the engine; the
    eager form many(xs) still evaluates at once.  Values equal the eager adapter's, bit for bit."""
    from spearmint_amd.chooser import GPEIChooser
    from tests.helpers import OracleEngine
    rs = np.random.RandomState(5)
    comp = rs.rand(25, 3); vals = rs.randn(25)
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=2,gpu_logprob=1")
    ch._eng = OracleEngine()
    calls = []
    orig = ch.data_logprob_many

    def counted(c, v, rows):
        calls.append(len(rows))
        return orig(c, v, rows)
    ch.data_logprob_many = counted

    def inside(x):
        return not ((x < 0).any() or (x > 2.0).any())

    def to_row(x):
        return np.concatenate(([0.1, 1e-3, 1.2], x)) if inside(x) else None
    xs = [rs.rand(3) * 1.5 for _ in range(4)] + [np.array([0.5, -0.1, 1.0])]
    nxt = [rs.rand(3) * 1.5 for _ in range(3)] + [np.array([2.5, 0.1, 1.0])]
    ev = ch._speculative_logprob(comp, vals, to_row, lambda x, lp: lp, admissible=inside)
    lv = ev.submit(xs, nxt)
    assert calls == [] and lv.missing == {0, 1, 2, 3}
    assert lv.get(4) == -np.inf and calls == []              # a priori: no call
    v0 = lv.get(0)
    assert calls == [7] and not lv.missing                   # 4 of the batch + 3 admissible extras, one call
    lv2 = ev.submit(nxt)
    assert not lv2.missing and [lv2.get(k) for k in range(4)][3] == -np.inf and calls == [7]
    eager = ch._speculative_logprob(comp, vals, to_row, lambda x, lp: lp, admissible=inside)
    ve = eager(xs + nxt)
    assert calls == [7, 7]
    assert [lv.get(k) for k in range(5)] + [lv2.get(k) for k in range(4)] == list(ve.values) and v0 == ve.values[0]
    # the row budget: 32 per call, the batch's own points first
    many_x = [rs.rand(3) * 1.5 for _ in range(30)]
    many_e = [rs.rand(3) * 1.5 for _ in range(10)]
    lv3 = ev.submit(many_x, many_e)
    lv3.get(0)
    assert calls[-1] == 32


# ---- batched local refinement (spearmint_amd/refine.py) ------------------------------------
def test_lbfgs_many_equals_serial_scipy():
    """Every L-BFGS-B instance fed from batched objective calls returns exactly what a serial
    fmin_l_bfgs_b run on the same objective returns."""
    import scipy.optimize as spo
    from spearmint_amd import refine
    rs = np.random.RandomState(0)
    A = rs.randn(5, 5); A = A @ A.T + np.eye(5)
    c = rs.rand(5)
    batches = []

    def f_one(x):
        r = x - c
        return float(r @ A @ r + np.sum(np.cos(3 * x))), 2 * A @ r - 3 * np.sin(3 * x)

    def f_many(X):
        batches.append(len(X))
        fs, gs = zip(*[f_one(x) for x in X])
        return np.array(fs), np.array(gs)

    pts = rs.rand(7, 5)
    bounds = [(0, 1)] * 5
    got = refine.lbfgs_many(f_many, pts, bounds)
    for i in range(7):
        ref = spo.fmin_l_bfgs_b(f_one, pts[i].copy(), bounds=bounds, disp=0)[0]
        assert np.array_equal(got[i], ref)
    assert batches[0] == 7 and min(batches) >= 1 and len(batches) < sum(batches)   # really batched
    assert refine.lbfgs_many(f_many, np.zeros((0, 5)), bounds).shape == (0, 5)
    # the serial fallback (serial=True / SPX_REFINE_SERIAL=1, for a scipy whose fmin_l_bfgs_b is not re-entrant)
    assert np.array_equal(refine.lbfgs_many(f_many, pts, bounds, serial=True), got)


def test_lockstep_refinement_equals_threads_and_serial_scipy(monkeypatch):
    """The lock-step driver (round 6: scipy's own `_minimize_lbfgsb` loop run for all instances in one thread, default
    where `lockstep_ok()`), the threaded form (SPX_REFINE_THREADS=1) and serial fmin_l_bfgs_b: the same optima bit for
    bit, the same number of batched objective calls, also from starting points outside the bounds (scipy clips them), with
    one-sided / missing bounds, and with an objective that fails."""
    import threading
    import scipy.optimize as spo
    from spearmint_amd import refine
    assert refine.lockstep_ok()          # scipy 1.15.3 here; a scipy with another setulb gets the threaded form
    rs = np.random.RandomState(2)
    D = 6
    A = rs.randn(D, D); A = A @ A.T + np.eye(D)
    c = rs.rand(D) * 1.4 - 0.2
    n_batches = {}

    def f_one(x):
        r = x - c
        return float(0.5 * r @ A @ r + 0.2 * np.sum(np.cos(4 * x))), A @ r - 0.8 * np.sin(4 * x)

    def f_many(X):
        n_batches[f_many.tag] = n_batches.get(f_many.tag, 0) + 1
        fs, gs = zip(*[f_one(x) for x in X])
        return np.array(fs), np.array(gs)

    pts = rs.rand(20, D) * 1.3 - 0.15
    for bounds in ([(0, 1)] * D, [(0, None), (None, 1), (None, None), (0.2, 0.9), (0, 1), (-1, 2)]):
        threads_before = threading.active_count()
        monkeypatch.setenv("SPX_REFINE_THREADS", "0")
        f_many.tag = "lockstep"
        a = refine.lbfgs_many(f_many, pts, bounds)
        assert threading.active_count() == threads_before
        monkeypatch.setenv("SPX_REFINE_THREADS", "1")
        f_many.tag = "threads"
        b = refine.lbfgs_many(f_many, pts, bounds)
        ref = np.array([spo.fmin_l_bfgs_b(f_one, p.copy(), bounds=bounds, disp=0)[0] for p in pts])
        assert np.array_equal(a, ref) and np.array_equal(b, ref)
        assert n_batches["lockstep"] == n_batches["threads"]
        n_batches.clear()
    monkeypatch.setenv("SPX_REFINE_THREADS", "0")

    def boom(X):
        raise RuntimeError("objective failed")
    with pytest.raises(RuntimeError):
        refine.lbfgs_many(boom, pts, [(0, 1)] * D)


def test_lbfgs_many_propagates_errors():
    from spearmint_amd import refine

    def boom(X):
        raise RuntimeError("objective failed")
    with pytest.raises(RuntimeError):
        refine.lbfgs_many(boom, np.random.rand(3, 2), [(0, 1)] * 2)


def test_lbfgs_many_abort_releases_its_threads():
    """Leaving the dispatch loop abnormally (here: KeyboardInterrupt out of the objective callback) must not strand
    the L-BFGS-B instances on the condition variable: they are aborted and joined, the interrupt propagates."""
    import threading
    from spearmint_amd import refine
    before = threading.active_count()
    calls = []

    def interrupt(X):
        calls.append(len(X))
        if len(calls) == 2:
            raise KeyboardInterrupt()
        return np.sum(X ** 2, axis=1), 2 * X
    with pytest.raises(KeyboardInterrupt):
        refine.lbfgs_many(interrupt, np.random.RandomState(1).rand(4, 3) + 0.2, [(0, 2)] * 3)
    assert threading.active_count() == before


def test_opt_next_with_batched_refinement_matches_reference(golden_dir, tmp_path):
    """gpu_refine=1 routes the grid_subset L-BFGS-B problems through refine.lbfgs_many +
    engine.ei_grad_batch (here the oracle's restatement of grad_optimize_ei_over_hypers): the
    proposal must still be the reference's."""
    g = _g(golden_dir, "chooser_next.npz")
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0,gpu_refine=1")
    ch._eng = OracleEngine()
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert isinstance(job, tuple) == bool(int(g["opt_is_new"]))
    if isinstance(job, tuple):
        assert job[0] == int(g["opt_index"]) and np.allclose(job[1], g["opt_point"], atol=1e-6)


def test_opt_next_pending_with_batched_refinement_matches_reference(golden_dir, tmp_path):
    g = _g(golden_dir, "chooser_next_pending.npz")
    ch = GPEIOptChooser.init(str(tmp_path), "mcmc_iters=3,burnin=4,grid_subset=3,pending_samples=8,"
                                            "use_multiprocessing=0,gpu_refine=1")
    ch._eng = OracleEngine()
    npr.seed(int(g["o_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    if int(g["o_is_new"]):
        assert isinstance(job, tuple) and job[0] == int(g["o_index"]) and np.allclose(job[1], g["o_point"], atol=1e-6)
    else:
        assert job == int(g["o_index"])


def test_pending_samples_range_is_checked(tmp_path):
    with pytest.raises(ValueError):
        GPEIChooser.init(str(tmp_path), "pending_samples=5000")
    GPEIChooser.init(str(tmp_path), "pending_samples=300")     # > 128 is fine now


def test_unpickle_reads_python2_state(tmp_path):
    """A state file written by the Python-2 reference holds numpy arrays as byte strings."""
    from spearmint_amd.helpers import unpickle
    # protocol-2 pickle of {'ls': ndarray} as Python 2 / numpy 1.x writes it (bytes payload, 'latin1' only)
    py2 = (b"\x80\x02}q\x00U\x02lsq\x01cnumpy.core.multiarray\n_reconstruct\nq\x02cnumpy\nndarray\nq\x03K\x00\x85q\x04U\x01b"
           b"q\x05\x87q\x06Rq\x07(K\x01K\x02\x85q\x08cnumpy\ndtype\nq\tU\x02f8q\nK\x00K\x01\x87q\x0bRq\x0c(K\x03U\x01<q\rNNNJ"
           b"\xff\xff\xff\xffJ\xff\xff\xff\xffK\x00tq\x0eb\x89U\x10\x00\x00\x00\x00\x00\x00\xf0?\x9a\x99\x99\x99\x99\x99\xb9?"
           b"q\x0ftq\x10bs.")
    p = tmp_path / "state.pkl"
    p.write_bytes(py2)
    st = unpickle(str(p))
    assert np.allclose(st["ls"], [1.0, 0.1])


def test_gpei_next_ml2_hypers_matches_reference(golden_dir, tmp_path):
    """mcmc_iters=0: ML-II hyper optimisation on the host (gp.py:181-292), one-row EI grid."""
    g = _g(golden_dir, "chooser_next_ml2.npz")
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=0")
    eng = OracleEngine(); ch._eng = eng
    npr.seed(5)
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    assert job == int(g["job"])
    assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g["hyper"], rtol=1e-6)
    assert eng.calls == [("ei_grid", len(g["candidates"]), 1)]
    for mod in (GPEIOptChooser, GPEIperSecChooser):      # the reference's own branches raise there
        c2 = mod.init(str(tmp_path), "mcmc_iters=0"); c2._eng = OracleEngine()
        with pytest.raises(NotImplementedError):
            c2.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])


def _noiseless_runs(golden_dir, tmp_path, make_engine):
    g = _g(golden_dir, "chooser_next_noiseless.npz")
    args = (g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    specs = (("g", GPEIChooser, "mcmc_iters=3,noiseless=1"),
             ("o", GPEIOptChooser, "mcmc_iters=3,burnin=4,grid_subset=3,noiseless=1,use_multiprocessing=0"),
             ("p", GPEIperSecChooser, "mcmc_iters=2,burnin=3,grid_subset=3,noiseless=1,ref_compat=1"))
    for tag, mod, arg in specs:
        d = tmp_path / tag
        d.mkdir()
        ch = mod.init(str(d), arg)
        eng = make_engine()
        if eng is not None:
            ch._eng = eng
        npr.seed(int(g[tag + "_seed"]))
        job = ch.next(*args)
        assert ch.noise == 1e-3
        assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g[tag + "_hyper"], rtol=1e-6)
        if int(g[tag + "_is_new"]):
            assert isinstance(job, tuple) and job[0] == int(g[tag + "_index"])
            assert np.allclose(job[1], g[tag + "_point"], atol=1e-5)
        else:
            assert job == int(g[tag + "_index"])


def _two_call_runs(golden_dir, tmp_path, make_engine):
    g = _g(golden_dir, "chooser_two_calls.npz")
    values, durations = g["values"], g["durations"]
    early = np.arange(len(values)) < 12
    first = (g["grid"], np.where(early, values, np.nan), np.where(early, durations, np.nan),
             np.arange(12, len(values)), g["pending"], np.arange(12))
    second = (g["grid"], values, durations, g["candidates"], g["pending"], g["complete"])
    specs = (("g", GPEIChooser, "mcmc_iters=3"),
             ("o", GPEIOptChooser, "mcmc_iters=3,burnin=4,grid_subset=3,use_multiprocessing=0"),
             ("p", GPEIperSecChooser, "mcmc_iters=2,burnin=3,grid_subset=3,ref_compat=1"))

    def check(job, ch, k):
        assert np.allclose(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls)), g["%s_h%d" % (tag, k)], rtol=1e-6)
        if int(g["%s_new%d" % (tag, k)]):
            assert isinstance(job, tuple) and job[0] == int(g["%s_idx%d" % (tag, k)])
            assert np.allclose(job[1], g["%s_pt%d" % (tag, k)], atol=1e-5)
        else:
            assert job == int(g["%s_idx%d" % (tag, k)])

    for tag, mod, arg in specs:
        d = tmp_path / tag
        d.mkdir()
        ch = mod.init(str(d), arg)
        eng = make_engine()
        if eng is not None:
            ch._eng = eng
        seed = int(g[tag + "_seed"])
        npr.seed(seed)
        check(ch.next(*first), ch, 1)
        if mod is GPEIChooser:
            ch.__del__()
        ch2 = mod.init(str(d), arg)
        eng = make_engine()
        if eng is not None:
            ch2._eng = eng
        npr.seed(seed + 7)
        check(ch2.next(*second), ch2, 2)


def test_restart_between_calls_matches_reference(golden_dir, tmp_path):
    """spearmint-lite builds a fresh chooser per invocation: the second next() of a NEW object in the same
    expt_dir starts from the pickled state (hypers, burn-in already done) exactly as the reference's does."""
    _two_call_runs(golden_dir, tmp_path, OracleEngine)


def _trajectory_runs(golden_dir, tmp_path, make_engine, extra=""):
    from tests.helpers import run_trajectory
    g = _g(golden_dir, "branin_trajectory.npz")
    for tag, mod, arg in (("g", GPEIChooser, "mcmc_iters=4"),
                          ("o", GPEIOptChooser, "mcmc_iters=3,burnin=5,grid_subset=4,use_multiprocessing=0"),
                          ("p", GPEIperSecChooser, "mcmc_iters=2,burnin=4,grid_subset=3,ref_compat=1")):
        d = tmp_path / tag
        d.mkdir()

        def make():
            ch = mod.init(str(d), arg + extra)
            eng = make_engine()
            if eng is not None:
                ch._eng = eng
            return ch
        props = run_trajectory(make, g["grid"], int(g[tag + "_iters"]), int(g[tag + "_seed"]))
        assert [p[0] for p in props] == list(g[tag + "_new"])
        assert [p[1] for p in props] == list(g[tag + "_idx"])
        # refined points come out of L-BFGS runs (atol as in the single-call goldens), and each feeds the next call
        assert np.allclose(np.array([p[2] for p in props]), g[tag + "_pts"], atol=2e-5)


def test_whole_branin_runs_match_reference(golden_dir, tmp_path):
    """24 / 14 / 12 consecutive proposals of the three choosers under a spearmint-lite-style loop (fresh
    chooser per call, state pickle in between, new points appended): the same experiments, in the same order,
    as the reference's own choosers produced with the same seeds."""
    _trajectory_runs(golden_dir, tmp_path, OracleEngine)


def test_noiseless_choosers_match_reference(golden_dir, tmp_path):
    """noiseless=1: noise pinned to 1e-3, joint slice move over [mean, amp2] only (GPEIChooser.py:268-270,
    :316-346): same hyper draws and proposals as the reference's own seeded runs."""
    _noiseless_runs(golden_dir, tmp_path, OracleEngine)


def test_host_code_parses_with_the_python2_grammar():
    """The reference driver is Python 2.7 (README.md:25): the chooser package, its helpers and the drop-in shims stay
    in the 2/3 common subset -- here: every file parses with lib2to3's Python 2 grammar."""
    import glob
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lib2to3 = pytest.importorskip("lib2to3")
        from lib2to3 import pygram, pytree
        from lib2to3.pgen2 import driver
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = driver.Driver(pygram.python_grammar, convert=pytree.convert)
    files = (glob.glob(os.path.join(root, "spearmint_amd", "*.py")) + glob.glob(os.path.join(root, "spearmint_amd", "chooser", "*.py"))
             + glob.glob(os.path.join(root, "dropin", "chooser", "*.py")))
    assert len(files) > 10
    for f in files:
        drv.parse_string(open(f).read() + "\n")


def test_uniform_stream_lookahead_leaves_the_generator_where_plain_draws_would():
    """util._Uniforms: peek() shows numbers ahead of their use, take() consumes them in order, close() leaves numpy's
    global generator exactly where as many plain npr.rand() calls as were taken would have left it -- with numbers
    peeked but never taken, with more taken than ever peeked, and with nothing peeked at all."""
    for peeks, takes in [((4, 2), 3), ((6,), 9), ((), 5), ((3, 8), 0), ((2,), 2)]:
        npr.seed(123)
        plain = [npr.rand() for _ in range(takes)]
        want = npr.get_state()
        npr.seed(123)
        u = util._Uniforms()
        seen = []
        for n in peeks:
            seen = u.peek(n)
            assert len(seen) == n
        got = [u.take() for _ in range(takes)]
        u.close()
        have = npr.get_state()
        assert got == plain
        assert list(seen[:takes]) == plain[:min(takes, len(seen))][:len(seen)]
        assert np.array_equal(want[1], have[1]) and want[2] == have[2]


def test_batched_row_builder_equals_the_per_point_one(tmp_path):
    """chooser._speculative_logprob with `to_rows` (the hyper rows and a-priori rejections of a whole speculative batch in
    a few array operations) gives the sampler exactly what the per-point `to_row` path gives: same values, same -inf
    entries, same memo behaviour -- checked through the length-scale sampler's adapter on the stand-in engine."""
    from spearmint_amd.chooser import GPEIChooser
    from tests.helpers import OracleEngine
    rs = np.random.RandomState(3)
    comp = rs.rand(30, 3); vals = rs.randn(30)
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=2,gpu_logprob=1")
    ch._eng = OracleEngine()
    max_ls = 2.0
    mean, noise, amp2 = 0.1, 1e-3, 1.2

    def inside(x):
        return not ((x < 0).any() or (x > max_ls).any())

    def to_row(x):
        return np.concatenate(([mean, noise, amp2], x)) if inside(x) else None

    def to_rows(X):
        ok = ~((X < 0) | (X > max_ls)).any(axis=1)
        R = np.empty((X.shape[0], 3 + X.shape[1]))
        R[:, 0] = mean; R[:, 1] = noise; R[:, 2] = amp2; R[:, 3:] = X
        return [R[k] if ok[k] else None for k in range(X.shape[0])]
    xs = [rs.rand(3) * 1.5 for _ in range(5)] + [np.array([0.5, -0.1, 1.0]), np.array([0.5, 2.5, 1.0])]
    a = ch._speculative_logprob(comp, vals, to_row, lambda x, lp: lp, admissible=inside)
    b = ch._speculative_logprob(comp, vals, to_row, lambda x, lp: lp, admissible=inside, to_rows=to_rows)
    for batch in (xs, xs[2:6] + xs[:1], xs[5:7], xs[:1]):
        va, vb = a(batch), b(batch)
        assert len(va.values) == len(vb.values)
        for k in range(len(batch)):
            assert (va.errors[k] is None) == (vb.errors[k] is None)
            assert va.values[k] == vb.values[k] or (np.isnan(va.values[k]) and np.isnan(vb.values[k]))
        assert [np.isneginf(v) for v in va.values] == [np.isneginf(v) for v in vb.values]


def test_fantasies_from_the_bottom_rows_of_the_factor_equal_the_reference_form():
    """hostgp.fantasize_from_factor_rows (what the GPU path uses: the bottom P rows of chol(cov([comp; pend]) + noise I)
    and gamma) against hostgp.fantasize_pending (GPEIChooser.py:219-249 restated: the N x N sub-Cholesky and two solves
    against it): the same fantasies and bests to rounding, also with a pending point 1e-4 away from an observation."""
    import scipy.linalg as spla
    rs = np.random.RandomState(4)
    for n, p, d, kname in ((30, 3, 2, "Matern52"), (120, 5, 6, "Matern52"), (65, 1, 3, "ARDSE"), (40, 4, 2, "Matern32")):
        comp, pend = rs.rand(n, d), rs.rand(p, d)
        pend[0] = comp[3] + 1e-4
        vals = np.sin(3 * comp).sum(axis=1) + 0.01 * rs.randn(n)
        row = np.concatenate(([vals.mean(), 10.0 ** rs.uniform(-4, -2), np.exp(0.5 * rs.randn())], rs.uniform(0.3, 2.0, d)))
        cp = np.concatenate((comp, pend))
        chol = spla.cholesky(hostgp.obs_cov(row[2], row[1], row[3:], cp, kname), lower=True)
        gamma = spla.solve_triangular(chol, np.concatenate((vals, np.zeros(p))) - row[0], lower=True)
        z = rs.randn(p, 50)
        f0, b0 = hostgp.fantasize_pending(comp, pend, vals, row, chol[:n, :n], z, kname)
        f1, b1 = hostgp.fantasize_from_factor_rows(vals, row, chol[n:, :], gamma, z)
        scale = np.abs(f0).max()
        assert np.allclose(f1, f0, rtol=0, atol=1e-9 * scale) and np.allclose(b1, b0, rtol=0, atol=1e-9 * scale)


def test_fantasies_from_factor_rows_with_large_noise_and_nearly_duplicate_pending_points():
    """fantasize_from_factor_rows forms pend_K = L_S L_S^T - noise I (the noise is added by the factorisation
    and taken off again), the reference pend_kappa - cross^T beta (GPEIChooser.py:229-236): equal in exact arithmetic,
    different by O(eps (noise + amp2)) in floating point.  Where that matters most -- noise as large as the amplitude,
    pending points 1e-5 apart (pend_K within the 1e-6 amp2 jitter of singular) -- both forms stay positive definite and
    the fantasies agree to 1e-6 of their scale (the Cholesky factor of a matrix this close to singular amplifies the
    1e-16 difference of its entries by up to 1 / sqrt(1e-6 jitter))."""
    import scipy.linalg as spla
    rs = np.random.RandomState(11)
    for n, p, d, noise in ((60, 4, 3, 1.0), (200, 6, 5, 0.3), (90, 3, 2, 2.0)):
        comp, pend = rs.rand(n, d), rs.rand(p, d)
        pend[1] = pend[0] + 1e-5                 # nearly duplicate pending points
        pend[2] = comp[7] + 1e-6                 # ... and one on top of an observation
        vals = np.sin(3 * comp).sum(axis=1) + 0.01 * rs.randn(n)
        row = np.concatenate(([vals.mean(), noise, 1.0], rs.uniform(0.5, 1.5, d)))
        cp = np.concatenate((comp, pend))
        chol = spla.cholesky(hostgp.obs_cov(row[2], row[1], row[3:], cp, "Matern52"), lower=True)
        gamma = spla.solve_triangular(chol, np.concatenate((vals, np.zeros(p))) - row[0], lower=True)
        z = rs.randn(p, 100)
        f0, b0 = hostgp.fantasize_pending(comp, pend, vals, row, chol[:n, :n], z, "Matern52")      # raises if not PD
        f1, b1 = hostgp.fantasize_from_factor_rows(vals, row, chol[n:, :], gamma, z)                # raises if not PD
        scale = np.abs(f0).max()
        assert np.allclose(f1, f0, rtol=0, atol=1e-6 * scale) and np.allclose(b1, b0, rtol=0, atol=1e-6 * scale)

"""The slice sampler inside libspx (csrc/spx_sampler.hip: spx_sample_hypers) against the reference's chain -- on the CPU.

`spx_sample_hypers_with` is the library's own sampler (the C++ control flow, priors, speculative batching and numpy's
legacy random stream that the GPU path runs) with the log-likelihood supplied by a callback; here the callback is the
host restatement of the reference's data term (hostgp.data_logprob, GPEIChooser.py:281-285).  What it must reproduce:

  * numpy.random.RandomState draw for draw (rand / randn with the cached second value / shuffle), any start state;
  * the Markov chain of the reference's sample_hypers -- util.slice_sample (pinned to the reference's golden trace in
    tests/test_host_logic.py) driven by the choosers' closures -- for GPEIChooser / GPEIOptChooser priors, noisy and
    noiseless, at every speculation depth: the same hyper rows BIT FOR BIT and the same generator state;
  * the reference's errors, at the reference's point of the stream.
The choosers on the test-only oracle engine (tests/helpers.py) run this same sampler, so every chooser golden of
tests/test_host_logic.py also passes through it; tests/test_gpu_a_parity.py does the same on the GPU."""
import ctypes
import tempfile

import numpy as np
import numpy.random as npr
import pytest

from spearmint_amd import engine as E
from tests import helpers as H
from spearmint_amd import hostgp, util
from spearmint_amd.chooser import GPEIChooser, GPEIOptChooser


def test_legacy_numpy_generator_draw_for_draw():
    lib = E.load_library()
    for seed in range(60):
        npr.seed(seed)
        for _ in range(seed % 7):
            npr.rand()
        if seed % 3 == 0:
            npr.randn()                       # leaves a cached gaussian in the state
        if seed % 11 == 0:
            npr.rand(623 - (seed % 5))        # close to the end of the key block: regeneration inside the draws
        st = npr.get_state()
        want_u = [npr.rand() for _ in range(700)]
        want_n = npr.randn(9)
        order = list(range(11 + seed % 4))
        npr.shuffle(order)
        after = npr.get_state()
        rs = E.RngState.from_numpy(st)
        u, n, o = np.empty(700), np.empty(9), np.empty(len(order), dtype=np.int32)
        assert lib.spx_rng_draw(ctypes.byref(rs), 700, E._dp(u), 9, E._dp(n), len(order), o.ctypes.data_as(E._c_int32_p)) == 0
        got = rs.to_numpy()
        assert np.array_equal(u, want_u) and np.array_equal(n, want_n) and list(o) == order
        assert np.array_equal(got[1], after[1]) and got[2:] == after[2:]


def _problem(rs, N, D):
    comp = rs.rand(N, D)
    vals = np.sin(3 * comp).sum(axis=1) + 0.1 * rs.randn(N)
    return comp, vals


def _cfg(ch, D, vals, n_iter, noiseless, la, fo, max_rows=32):
    return E.SamplerCfg(D=D, n_iter=n_iter, noiseless=noiseless, check_mean=int((not noiseless) or ch.noiseless_checks_mean),
                        amp2_prior_on_sqrt=int(ch.amp2_prior_on_sqrt), lookahead=la, follow_props=fo[0], follow_hyps=fo[1],
                        max_rows=max_rows, noise_scale=ch.noise_scale, amp2_scale=ch.amp2_scale, max_ls=ch.max_ls,
                        vals_min=float(np.min(vals)), vals_max=float(np.max(vals)))


def _rows_lp(comp, vals, sizes=None):
    def f(rows):
        if sizes is not None:
            sizes.append(len(rows))
        out = np.empty(len(rows))
        for i, r in enumerate(rows):
            try:
                out[i] = hostgp.data_logprob(comp, vals, r[0], r[2], r[1], r[3:], "Matern52")
            except np.linalg.LinAlgError:
                out[i] = -np.inf
        return out
    return f


@pytest.mark.parametrize("mod,D,N,noiseless,la,fo", [
    (GPEIOptChooser, 4, 30, 0, 6, (0, 0)), (GPEIOptChooser, 8, 40, 0, 8, (4, 2)), (GPEIChooser, 3, 25, 0, 6, (3, 1)),
    (GPEIOptChooser, 3, 25, 1, 4, (6, 3)), (GPEIChooser, 5, 25, 1, 8, (4, 2)), (GPEIChooser, 1, 12, 0, 1, (2, 2)),
    (GPEIOptChooser, 2, 20, 0, 2, (1, 1))])
def test_native_sampler_is_the_reference_chain(mod, D, N, noiseless, la, fo):
    comp, vals = _problem(np.random.RandomState(D * 100 + N), N, D)
    n_iter = 10
    ref = mod.init(tempfile.mkdtemp(), "mcmc_iters=3,gpu_logprob=0,noiseless=%d" % noiseless)   # the host's serial slice_sample
    ref._real_init(D, vals)
    npr.seed(5)
    want = []
    for _ in range(n_iter):
        if hasattr(ref, "hyper_samples"):
            ref.hyper_samples = []
        ref.sample_hypers(comp, vals)
        want.append(ref.current_hyper_row().copy())
    s_want = npr.get_state()
    ch = mod.init(tempfile.mkdtemp(), "mcmc_iters=3,noiseless=%d" % noiseless)
    ch._real_init(D, vals)
    hyper, hist, sizes = ch.current_hyper_row().copy(), np.zeros(12), []
    npr.seed(5)
    rows, st = H.sample_hypers_with(_rows_lp(comp, vals, sizes), _cfg(ch, D, vals, n_iter, noiseless, la, fo), hyper, hist)
    s_got = npr.get_state()
    assert np.array_equal(rows, np.array(want))                    # bit for bit
    assert np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:]
    assert np.array_equal(hyper, want[-1])
    assert st["iterations"] == n_iter and st["moves"] == n_iter * (1 + D) and st["calls"] == len(sizes)
    assert max(sizes) <= 32 and st["rows"] == sum(sizes)
    if fo[0] > 0 and D >= 3:
        assert st["free_moves"] > 0                                # some coordinate moves were served by the previous call
    assert hist.sum() > 0
    # one call of n_iter iterations == n_iter calls of one (the generator state carries over)
    ch2 = mod.init(tempfile.mkdtemp(), "mcmc_iters=3,noiseless=%d" % noiseless)
    ch2._real_init(D, vals)
    hyper2, hist2 = ch2.current_hyper_row().copy(), np.zeros(12)
    npr.seed(5)
    for k in range(n_iter):
        r1, _ = H.sample_hypers_with(_rows_lp(comp, vals), _cfg(ch2, D, vals, 1, noiseless, la, fo), hyper2, hist2)
        assert np.array_equal(r1[0], want[k])
    assert np.array_equal(npr.get_state()[1], s_want[1])


def test_cross_move_speculation_saves_calls_not_accuracy():
    comp, vals = _problem(np.random.RandomState(77), 40, 8)
    ch = GPEIOptChooser.init(tempfile.mkdtemp(), "mcmc_iters=3")
    ch._real_init(8, vals)
    out = {}
    for fo in ((0, 0), (4, 2), (6, 3)):
        hyper, hist = ch.current_hyper_row().copy(), np.zeros(12)
        npr.seed(9)
        rows, st = H.sample_hypers_with(_rows_lp(comp, vals), _cfg(ch, 8, vals, 20, 0, 8, fo), hyper, hist)
        out[fo] = (rows, st, npr.get_state())
    for fo in ((4, 2), (6, 3)):
        assert np.array_equal(out[fo][0], out[(0, 0)][0]) and np.array_equal(out[fo][2][1], out[(0, 0)][2][1])
    assert out[(0, 0)][1]["free_moves"] == 0
    assert out[(4, 2)][1]["calls"] < 0.85 * out[(0, 0)][1]["calls"], {k: v[1] for k, v in out.items()}
    assert out[(6, 3)][1]["calls"] < 0.80 * out[(0, 0)][1]["calls"], {k: v[1] for k, v in out.items()}


def test_native_sampler_raises_what_the_reference_raises():
    comp, vals = _problem(np.random.RandomState(3), 20, 3)
    ch = GPEIOptChooser.init(tempfile.mkdtemp(), "mcmc_iters=3")
    ch._real_init(3, vals)
    cfg = _cfg(ch, 3, vals, 6, 0, 6, (4, 2))
    good = _rows_lp(comp, vals)
    # reference: util.slice_sample with a log-probability that turns NaN / fails after `k` evaluations
    for kind, exc in (("nan", util.SliceSamplerError), ("notpd", np.linalg.LinAlgError)):
        ref = GPEIOptChooser.init(tempfile.mkdtemp(), "mcmc_iters=3,gpu_logprob=0")
        ref._real_init(3, vals)
        poison = {}

        def bad_point(r):      # a region of hyper space, so that serial and batched evaluation agree on WHICH points fail
            return r[3] > (1.5 if kind == "nan" else 1.7)

        orig = hostgp.data_logprob

        def host_lp(c, v, mean, amp2, noise, ls, covar):
            r = np.concatenate(([mean, noise, amp2], ls))
            if bad_point(r):
                if kind == "nan":
                    return np.nan
                raise np.linalg.LinAlgError("not PD")
            return orig(c, v, mean, amp2, noise, ls, covar)
        hostgp.data_logprob = host_lp
        try:
            npr.seed(13)
            done = []
            with pytest.raises(exc):
                for _ in range(6):
                    ref.hyper_samples = []
                    ref.sample_hypers(comp, vals)
                    done.append(ref.current_hyper_row().copy())
            s_want = npr.get_state()
            at_error = ref.current_hyper_row().copy()
        finally:
            hostgp.data_logprob = orig

        def rows_lp(rows):
            out = good(rows)
            for i, r in enumerate(rows):
                if bad_point(r):
                    out[i] = np.nan if kind == "nan" else -np.inf
            return out
        hyper, hist = ch.current_hyper_row().copy(), np.zeros(12)
        npr.seed(13)
        with pytest.raises(exc) as info:
            H.sample_hypers_with(rows_lp, cfg, hyper, hist)
        assert 0 < len(done) < 6                                         # the failure really came mid-run
        assert np.array_equal(info.value.rows_done, np.array(done))       # the iterations before it are delivered
        assert np.array_equal(hyper, at_error)                            # a finished joint move applied, the sweep not
        s_got = npr.get_state()
        assert np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:]
    # a failing callback is an error, not a crash
    def boom(rows):
        raise RuntimeError("evaluator failed")
    with pytest.raises(RuntimeError):
        H.sample_hypers_with(boom, cfg, ch.current_hyper_row().copy(), np.zeros(12))
    with pytest.raises(ValueError):
        H.sample_hypers_with(good, _cfg(ch, 3, vals, 2, 0, 0, (0, 0)), ch.current_hyper_row().copy(), np.zeros(12))


def test_chooser_default_is_the_native_sampler_and_equals_the_python_one(tmp_path):
    """GPEIOptChooser / GPEIChooser.next() on the oracle engine: sampler=native (default; one library call per loop of
    iterations) and sampler=python (util.slice_sample_batched, round 5) propose the same point from the same hyper samples
    and leave the generator in the same state."""
    from tests.helpers import OracleEngine
    rs = np.random.RandomState(8)
    grid = rs.rand(300, 3)
    values = np.full(300, np.nan)
    values[:24] = np.sin(4 * grid[:24]).sum(axis=1)
    complete, candidates, pending = np.arange(24), np.arange(24, 300), np.array([], dtype=int)
    for mod, args in ((GPEIOptChooser, "mcmc_iters=4,burnin=5,grid_subset=3,use_multiprocessing=0"), (GPEIChooser, "mcmc_iters=5")):
        got = {}
        for sampler in ("native", "python"):
            d = tmp_path / (mod.__name__.split(".")[-1] + sampler)
            d.mkdir()
            ch = mod.init(str(d), args + ",sampler=" + sampler + ",lookahead=6,follow=4:2")
            eng = OracleEngine(); ch._eng = eng
            npr.seed(21)
            job = ch.next(grid, values, np.ones(300), candidates, pending, complete)
            got[sampler] = (job, ch.current_hyper_row(), npr.get_state(), len(eng.native_calls), dict(ch.sampler_stats))
        a, b = got["native"], got["python"]
        assert (a[0] == b[0]) if isinstance(a[0], int) else (a[0][0] == b[0][0] and np.array_equal(a[0][1], b[0][1]))
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2][1], b[2][1]) and a[2][2:] == b[2][2:]
        assert a[3] > 0 and b[3] == 0 and a[4]["moves"] > 0 and a[4]["calls"] == a[3]

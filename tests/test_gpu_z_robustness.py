"""Resource hygiene and robustness of the handle (collected LAST: a failure here must not hide the parity tests):
device memory comes back after spx_destroy, an impossible allocation is an error code, every error leaves the handle
usable, and the timing-only ablation kernels are not in the shipped library."""
import os
import time

import numpy as np
import pytest

from oracle import gp_ei_oracle as orc
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def test_ablation_variants_are_not_in_the_shipped_library(eng):
    with pytest.raises(ValueError):
        eng.set_option("gemm_waves", 41)
    with pytest.raises(ValueError):
        eng.set_option("gemm_waves", 5)
    comp, cand, vals, hypers = synthetic_problem(300, 2000, 6, 3, 73)
    try:
        eng.set_option("gemm_waves", 8)                 # a real variant (8 waves: other summation order in the epilogue)
        a = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
        other = Engine(0)                               # the option is per handle, not per process
        b = other.ei_grid(comp, vals, cand, hypers, want_draws=True)
        other.close()
    finally:
        eng.set_option("gemm_waves", 0)
    c = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert np.array_equal(b[3], c[3]) and b[0] == c[0]
    big = c[3] >= 1e-280          # the far EI tail amplifies rounding differences (u^2 ~ 1e3)
    assert a[0] == c[0] and np.max(np.abs(a[3][big] - c[3][big]) / c[3][big]) <= 1e-7


def test_handles_release_their_device_memory():
    """Create / use / destroy handles in a loop (every buffer family: plain EI, per second, fantasies, refinement,
    log-likelihood, Sobol): the device's free memory comes back, so spx_destroy's buffer list is complete."""
    from spearmint_amd import sobol
    comp, cand, vals, hypers, ld, th = synthetic_problem(300, 20000, 6, 3, 77, per_sec=True)
    rs = np.random.RandomState(0)

    def cycle():
        e = Engine(0)
        e.ei_grid(comp, vals, cand, hypers, want_draws=True)
        e.ei_per_sec_grid(comp, vals, ld, cand, hypers, th)
        e.ei_grad_batch(cand[:5])
        e.set_observations(comp, vals); e.set_candidates(cand); e.set_hypers(hypers); e.factor()
        e.set_fantasies(rs.randn(3, 300, 9), rs.randn(3, 9))
        e.ei_run(); e.ei_grad_batch(cand[:3])
        e.set_hypers(hypers); e.gp_logprob()
        e.sobol_grid(sobol.load_dirs("bf40"), 8, 50000, 1)
        e.close()

    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")     # the runtime libspx is linked against: no second framework in this measurement
    # (torch.cuda.mem_get_info needs torch's own device initialisation, which fails -- "No HIP GPUs are available" -- when
    # this test is the first in the process to touch the GPU: the round-4 isolated run of this file)

    def free_now():
        # hipMemGetInfo is not a steady figure on this runtime: scripts/dev/leak_probe.py (profiles/r04_leak_probe.log)
        # shows 0.000 MiB per lifetime for every op, and -- while three kernels still had a private segment -- now and then
        # ONE reading 200+ MiB low that was back a lifetime later (per-queue scratch).  A leak is what does NOT come back:
        # the highest of a few readings, a short wait apart.
        best = 0
        for _ in range(5):
            assert hip.hipDeviceSynchronize() == 0
            free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
            assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
            best = max(best, int(free_b.value))
            time.sleep(0.02)
        return best

    for _ in range(4):          # the runtime keeps some freed blocks for reuse: let that settle first
        cycle()
    free0 = free_now()
    lost = []
    for _ in range(12):
        cycle()
    lost.append(free0 - free_now())
    for _ in range(3):          # a held chunk is back a lifetime later; a leak has grown by then
        if lost[-1] < (16 << 20):
            break
        cycle()
        lost.append(free0 - free_now())
    assert min(lost) < (16 << 20), "leaked %s MiB over 12.. handle lifetimes" % [round(x / 2.0 ** 20, 1) for x in lost]


def test_out_of_device_memory_is_an_error_code_not_a_crash(eng):
    """A K(X*,X) staging budget the device cannot satisfy (315 GB asked of 288 GB): the call returns the HIP error
    through spx_last_error -> SpxError, and the same handle works again once the budget is sane."""
    from spearmint_amd.engine import SpxError
    comp, cand, vals, hypers = synthetic_problem(4096, 600000, 4, 16, 78)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers); eng.factor()
    eng.set_option("kstar_budget_bytes", 1 << 40)
    try:
        with pytest.raises(SpxError) as err:
            eng.ei_run()
        assert "hipMalloc" in str(err.value)
    finally:
        eng.set_option("kstar_budget_bytes", 0)     # back to the default
    eng.ei_run()
    idx, val = eng.best()
    sub = np.r_[idx, 0:200]
    ref = orc.ei_over_hypers(comp, cand[sub], vals, hypers)
    got = eng.ei_mean()[sub]
    assert np.allclose(got, np.mean(ref, axis=1), rtol=1e-6, atol=1e-300)


def test_handle_survives_argument_and_numerical_errors(eng):
    """Every error is a return code; the handle keeps working afterwards, with unchanged results."""
    from numpy.linalg import LinAlgError
    comp, cand, vals, hypers = synthetic_problem(150, 3000, 5, 3, 79)
    ref = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    bad = hypers.copy(); bad[1, 2] = -1.0                      # negative amplitude: not positive definite
    with pytest.raises(LinAlgError):
        eng.ei_grid(comp, vals, cand, bad)
    with pytest.raises(ValueError):
        eng.ei_grid(comp, vals, cand[:, :4], hypers)           # candidates of another dimension
    with pytest.raises(ValueError):
        eng.ei_grid(comp, vals, cand, hypers[:, :6])           # hyper rows too short
    eng.set_observations(comp, vals); eng.set_hypers(hypers)
    with pytest.raises(ValueError):
        eng.ei_run()                                           # nothing factored
    with pytest.raises(ValueError):
        eng.ei_grad_batch(cand[:2])                            # no resident factorisation either
    lp = eng.gp_logprob()                                      # -inf is a value here, not an error
    eng.set_hypers(bad)
    assert np.isneginf(eng.gp_logprob()[1]) and np.array_equal(eng.gp_logprob()[[0, 2]], lp[[0, 2]])
    again = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    assert again[0] == ref[0] and np.array_equal(again[3], ref[3])




def test_step_overlap_under_concurrent_load(eng):
    """spx_ei_step's second stream (candidate side beside the factorisation) while another engine keeps the GPU busy from
    another host thread: random sizes through both forms of the EI pass, every result compared bit for bit with the
    one-stream form.  A missing event dependency between the two streams would show here as different bits."""
    import threading
    stop = []

    def noise():
        e2 = Engine(0)
        comp, cand, vals, hyp = synthetic_problem(500, 30000, 7, 4, 299)
        while not stop:
            e2.ei_grid(comp, vals, cand, hyp, want_mean=False)
        e2.close()
    th = threading.Thread(target=noise)
    th.start()
    rs = np.random.RandomState(23)
    try:
        for _ in range(25):
            N = int(rs.choice([7, 40, 64, 100, 128, 129, 256, 400, 900]))
            M = int(rs.choice([300, 5000, 40000]))
            D = int(rs.choice([1, 4, 9, 33]))
            H = int(rs.randint(1, 12))
            per_sec = bool(rs.rand() < 0.35)
            prob = synthetic_problem(N, M, D, H, int(rs.randint(1 << 30)), per_sec=per_sec)
            comp, cand, vals, hyp = prob[:4]
            eng.set_observations(comp, vals); eng.set_candidates(cand)
            res = {}
            for ov in (0, 1, 1):
                eng.set_option("step_overlap", ov)
                eng.set_hypers(hyp)
                if per_sec:
                    eng.set_time_model(prob[4], prob[5])
                eng.ei_step(1 if per_sec else 0)
                got = (eng.best(), eng.ei_draws())
                if ov in res:
                    assert got[0] == res[ov][0] and np.array_equal(got[1], res[ov][1])
                res[ov] = got
            assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]), (N, M, D, H, per_sec)
    finally:
        stop.append(1)
        th.join()
        eng.set_option("step_overlap", -1)

"""Parity margin on the ILL-CONDITIONED family, asserted.

A6 is built as W = L^-1 times K* (an MFMA GEMM) instead of LAPACK's substitution; both are backward stable,
but their forward errors differ by O(eps * cond(K)), so the GPU-vs-reference EI difference is largest where
cond(K) is: smooth kernels (ARDSE / SE) or a one-dimensional input with more than a thousand observations packed
on the unit interval, small noise.  The random sweeps of round 2 (profiles/r02_fuzz_400.log) found their three
worst cases there (2.2e-7, 3.8e-7, 9.6e-7).  These fixed-seed cases pin that family: EI within 2e-6 relative of the
oracle for every value >= 1e-250 (north star: 1e-5) and the identical argmax."""
import numpy as np
import pytest

from oracle import gp_ei_oracle as orc
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem

pytestmark = pytest.mark.gpu

CASES = [
    # covar,     N,    M,     D, H, per_sec, seed, noise (None = the sampled 1e-4 .. 1e-2)
    ("ARDSE",    1500, 20001, 1, 8, True,  9101, None),
    ("SE",       2049, 500,   1, 2, False, 9102, None),
    ("ARDSE",    2049, 20001, 1, 3, False, 9103, None),
    ("Matern52", 1500, 4097,  1, 4, False, 9104, None),
    ("Matern52", 2049, 4097,  1, 3, False, 9105, 1e-3),      # noiseless=1 pins the noise at 1e-3
    ("ARDSE",    1025, 4097,  2, 3, True,  9106, None),
    ("SE",       1500, 4097,  3, 3, False, 9107, 1e-4),
    ("Matern32", 2049, 1000,  1, 2, False, 9108, 1e-4),
]


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("covar,N,M,D,H,per_sec,seed,noise", CASES)
def test_ill_conditioned_family_stays_inside_2e_6(eng, covar, N, M, D, H, per_sec, seed, noise):
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
    if noise is not None:
        hypers[:, 1] = noise
    rs = np.random.RandomState(seed)
    try:
        eng.set_covar(covar)
        with orc.covar(covar):
            if per_sec:
                log_durs = 0.5 * np.sin(comp.sum(axis=1))
                thyp = hypers.copy(); thyp[:, 0] = 0.1; thyp[:, 3:] = rs.uniform(0.3, 5.0, (H, D))
                best, val, mean, draws = eng.ei_per_sec_grid(comp, vals, log_durs, cand, hypers, thyp, want_draws=True)
                ref = np.stack([orc.compute_ei_per_s(comp, cand, vals, log_durs, hypers[h], thyp[h]) for h in range(H)], axis=1)
            else:
                best, val, mean, draws = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
                ref = np.stack([orc.compute_ei(comp, cand, vals, hypers[h]) for h in range(H)], axis=1)
    finally:
        eng.set_covar("Matern52")
    big = ref >= 1e-250
    assert big.any()
    err = float(np.max(np.abs(draws[big] - ref[big]) / ref[big]))
    print("ill-conditioned %s N=%d D=%d: max EI rel err %.2e" % (covar, N, D, err))
    assert err <= 2e-6, err
    assert np.array_equal(np.isnan(draws), np.isnan(ref))
    assert best == int(np.argmax(np.mean(ref, axis=1)))

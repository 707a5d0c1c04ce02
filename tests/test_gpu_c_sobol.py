"""Sobol candidate grid: oracle vs reference golden vectors (CPU), HIP
generator vs oracle and golden, bit for bit (GPU)."""
import os
import sys

import numpy as np
import pytest

from oracle import sobol_oracle as so
from spearmint_amd import sobol as ssob

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sobol.npz"))
CASES = [k for k in GOLD.files if not k.endswith("_args")]


def table_of(case):
    return "bf40" if case.startswith("bf_") else "jk1111"


# ---- CPU: the oracle is pinned by what the reference itself produced -------------------------
@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_bit_for_bit(case):
    m, n, skip = (int(v) for v in GOLD[case + "_args"])
    r = so.i4_sobol_generate(m, n, skip, ssob.load_dirs(table_of(case)))
    assert r.shape == (m, n) and r.dtype == np.float64
    assert np.array_equal(r, GOLD[case])


def test_direction_tables_are_well_formed():
    for table, dim_max in (("bf40", 40), ("jk1111", 1111)):
        V = ssob.load_dirs(table)
        assert V.shape == (dim_max, 30) and V.dtype == np.uint32
        # V[d][b] = m_b * 2^(29 - b) with m_b odd and < 2^(b + 1): bit 29 - b set, nothing below
        # it, nothing at or above bit 30 -- what makes each coordinate a (0, 1)-sequence in base 2
        assert np.all(V < (1 << 30))
        for b in range(30):
            assert np.all((V[:, b] >> (29 - b)) & 1 == 1)
            assert np.all(V[:, b] & ((1 << (29 - b)) - 1) == 0)
        assert np.all(V[0] == (1 << 29) >> np.arange(30))     # dimension 1 = van der Corput
    a, b = ssob.load_dirs("bf40"), ssob.load_dirs("jk1111")
    assert np.array_equal(a[:20], b[:20]) and not np.array_equal(a, b[:40])


def test_oracle_properties():
    V = ssob.load_dirs("jk1111")
    r = so.i4_sobol_generate(16, 1024, 1, V)
    assert np.all(r[:, 0] == 0.0)                 # skip = 1: the first point is the origin
    for d in range(16):                           # first 2^10 points: a permutation of k / 1024
        assert np.array_equal(np.sort(r[d]), np.arange(1024) / 1024.0)
    # skip <= 0: negative seeds are clamped to 0, so the origin repeats
    r0 = so.i4_sobol_generate(3, 5, -1, V)
    assert np.all(r0[:, :3] == 0.0) and np.array_equal(r0[:, 3:], r[:3, 1:3])
    with pytest.raises(ValueError):
        so.i4_sobol_generate(41, 4, 1, ssob.load_dirs("bf40"))
    with pytest.raises(ValueError):
        so.i4_sobol_generate(2, 4, 1 << 30, V)


# ---- GPU: the HIP generator through the C ABI -------------------------------------------------
@pytest.fixture(scope="module")
def eng():
    from spearmint_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_matches_reference_bit_for_bit(eng, case):
    m, n, skip = (int(v) for v in GOLD[case + "_args"])
    r = ssob.i4_sobol_generate(m, n, skip, table=table_of(case), engine=eng)
    assert r.shape == (m, n)
    assert np.array_equal(r, GOLD[case])
    assert np.transpose(r).flags["C_CONTIGUOUS"]     # ExperimentGrid's transpose is free


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,skip,table", [
    (1, 1, 1, "bf40"), (1, 4097, 5, "jk1111"), (40, 3000, 2, "bf40"), (33, 1001, 777, "jk1111"),
    (600, 257, 1, "jk1111"),          # > 512 dimensions: direction rows read through L2, not LDS
    (1111, 65, 123456, "jk1111"), (7, 100000, 0, "jk1111"), (3, 50, (1 << 30) - 49, "bf40")])
def test_gpu_matches_oracle_bit_for_bit(eng, m, n, skip, table):
    V = ssob.load_dirs(table)
    grid, ms = eng.sobol_grid(V, m, n, skip)
    assert grid.shape == (n, m) and ms >= 0.0
    assert np.array_equal(grid.T, so.i4_sobol_generate(m, n, skip, V))


@pytest.mark.gpu
def test_gpu_full_size_grid_properties_and_checksum(eng):
    """BASELINE config 4's grid: 1M x 32 (2^20 points).  Every coordinate must be a permutation of
    k / 2^20, the first point the origin, and the whole grid equal to the oracle's."""
    V = ssob.load_dirs("jk1111")
    n, m = 1 << 20, 32
    grid, ms = eng.sobol_grid(V, m, n, 1)
    assert np.all(grid[0] == 0.0)
    k = np.rint(grid * n).astype(np.int64)
    assert np.array_equal(k / float(n), grid)
    for d in range(m):
        assert np.array_equal(np.bincount(k[:, d], minlength=n), np.ones(n, dtype=np.int64))
    assert np.array_equal(grid.T, so.i4_sobol_generate(m, n, 1, V))


@pytest.mark.gpu
def test_gpu_grid_as_resident_candidates(eng):
    """as_candidates: the generated grid is the candidate set of the next EI run, no host copy."""
    from spearmint_amd.synthetic import synthetic_problem
    comp, _, vals, hypers = synthetic_problem(96, 16, 6, 3, 11)
    V = ssob.load_dirs("jk1111")
    eng.set_observations(comp, vals)
    grid, _ = eng.sobol_grid(V, 6, 5000, 1, as_candidates=True)
    eng.set_hypers(hypers); eng.factor(); eng.ei_run()
    i1, v1 = eng.best(); e1 = eng.ei_mean()
    eng.set_candidates(grid)
    eng.factor(); eng.ei_run()
    i2, v2 = eng.best()
    assert (i1, v1) == (i2, v2) and np.array_equal(e1, eng.ei_mean())
    with pytest.raises(ValueError):
        eng.sobol_grid(V, 5, 10, 1, as_candidates=True)      # dimension differs from the observations


@pytest.mark.gpu
def test_gpu_argument_errors(eng):
    V = ssob.load_dirs("bf40")
    with pytest.raises(ValueError):
        eng.sobol_grid(V, 41, 4, 1)
    with pytest.raises(ValueError):
        eng.sobol_grid(V, 2, 4, 1 << 30)
    with pytest.raises(ValueError):
        eng.sobol_grid(V[:, :29], 2, 4, 1)


@pytest.mark.gpu
def test_install_rebinds_reference_generator(eng):
    """gpu_sobol=1: a module that did `from sobol_lib import *` gets the GPU generator, and only
    if its own generator agrees with one of the known tables."""
    import types
    V = ssob.load_dirs("bf40")
    calls = []

    def ref_generate(m, n, skip):            # stands in for the reference's function (40-dim table)
        calls.append((m, n, skip))
        return so.i4_sobol_generate(m, n, skip, V)

    def other_generate(m, n, skip):          # some other generator: must be left alone
        return np.full((m, n), 0.25)

    good = types.ModuleType("spx_fake_grid"); good.i4_sobol_generate = ref_generate
    bad = types.ModuleType("spx_fake_other"); bad.i4_sobol_generate = other_generate
    sys.modules["spx_fake_grid"] = good; sys.modules["spx_fake_other"] = bad
    try:
        done = ssob.install(modules=("spx_fake_grid", "spx_fake_other", "spx_not_imported"), engine=eng)
        assert done == ["spx_fake_grid", "spx_fake_other"]
        assert ssob.install(modules=("spx_fake_grid",), engine=eng) == []        # idempotent
        g = good.i4_sobol_generate(5, 300, 1)
        assert calls == [(40, 16, 3)]                                         # only the identification probe
        assert np.array_equal(g, so.i4_sobol_generate(5, 300, 1, V))
        assert np.array_equal(bad.i4_sobol_generate(2, 3, 1), np.full((2, 3), 0.25))
        ssob.uninstall(modules=("spx_fake_grid", "spx_fake_other"))
        assert good.i4_sobol_generate is ref_generate and bad.i4_sobol_generate is other_generate
    finally:
        del sys.modules["spx_fake_grid"], sys.modules["spx_fake_other"]

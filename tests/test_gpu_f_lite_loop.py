""""Driver untouched", on the GPU: the reference's OWN spearmint-lite loop
(`spearmint-lite/spearmint-lite.py:88-215`, `main_controller`) drives our drop-in
chooser modules on the real `libspx.so`.

`__graft_entry__.build()` converts the reference's driver for Python 3 (stdlib
lib2to3 + the import rewrite of SURVEY.md Appendix D, nothing else) into the archive
`oracle/_ref/lite_py3.zip` where /root/reference exists; the archive is git-ignored but
travels to the GPU box with the tree, like a built .so, and is unpacked here into a
temporary directory.  Its unmodified
`main_controller` is run with `dropin/` on the path -- `--method=GPEIChooser |
GPEIOptChooser | GPEIperSecChooser` resolves to OUR modules -- for seven proposals each,
the last two in ONE call so that the second sees the first as a pending job, and the
whole run is repeated with the test-only oracle engine in place of the HIP engine:
the two results files must hold the same proposals.
"""
import importlib
import os
import sys
import types
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LITE_ZIP = os.path.join(ROOT, "oracle", "_ref", "lite_py3.zip")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isfile(LITE_ZIP),
                                 reason="oracle/_ref/lite_py3.zip absent: run __graft_entry__.build() where /root/reference "
                                        "exists (it converts the reference's spearmint-lite driver)")]

_LITE_MODULES = ("spearmint_lite_main", "ExperimentGrid", "sobol_lib", "Locker", "util")


def _branin(x):
    a = x[0] * 15
    b = (x[1] * 15) - 5
    return float(np.square(b - (5.1 / (4 * np.square(np.pi))) * np.square(a) + (5 / np.pi) * a - 6)
                 + 10 * (1 - (1. / (8 * np.pi))) * np.cos(a) + 10)


def _forget_lite_modules():
    for name in [m for m in sys.modules if m == "chooser" or m.startswith("chooser.") or m in _LITE_MODULES]:
        sys.modules.pop(name, None)


def _run_loop(work, method, margs, seed, engine_factory):
    """Seven proposals through the reference's main_controller; returns the proposal lines and the engine
    objects the chooser created."""
    import spearmint_amd.engine as eng
    made = []
    saved_path = list(sys.path)
    saved_engine = eng.Engine
    _forget_lite_modules()
    for p in (ROOT, os.path.join(ROOT, "dropin"), work):      # PYTHONPATH=<lite>:dropin:<repo>
        sys.path.insert(0, p)
    if engine_factory is not None:
        def make(*a, **k):
            made.append(engine_factory())
            return made[-1]
        eng.Engine = make
    else:
        def make(*a, **k):
            made.append(saved_engine(*a, **k))
            return made[-1]
        eng.Engine = make
    try:
        mod = importlib.import_module("spearmint_lite_main")
        expt = os.path.join(work, "braninpy")
        res = os.path.join(expt, "results.dat")
        open(res, "w").close()
        opts = types.SimpleNamespace(num_jobs=1, max_finished_jobs=1000, chooser_module=method, chooser_args=margs,
                                     grid_size=400, grid_seed=1, config_file="config.json", results_file="results.dat")
        np.random.seed(seed)
        proposals = []
        for it in range(5):
            mod.main_controller(opts, [expt])
            lines = open(res).read().strip().split("\n")
            assert lines[-1].startswith("P P ")
            proposals.append(lines[-1])
            x = [float(v) for v in lines[-1].split()[2:]]
            # the job "finishes": value and a duration that depends on the point (the per-second chooser models it)
            lines[-1] = "%f %f %s" % (_branin(x), 1.0 + 3.0 * x[0] + np.sin(5 * x[1]) ** 2,
                                      " ".join(lines[-1].split()[2:]))
            open(res, "w").write("\n".join(lines) + "\n")
        opts.num_jobs = 2           # two proposals in one call: the second one sees the first as PENDING
        mod.main_controller(opts, [expt])
        lines = open(res).read().strip().split("\n")
        assert lines[-1].startswith("P P ") and lines[-2].startswith("P P ")
        proposals += lines[-2:]
        ch_mod = sys.modules["chooser." + method]
        assert "dropin" in ch_mod.__file__            # OUR module was the one the driver loaded, under the reference's name
        assert os.path.exists(os.path.join(expt, "chooser.%s.pkl" % method))
        return proposals, made
    finally:
        eng.Engine = saved_engine
        sys.path[:] = saved_path
        _forget_lite_modules()


@pytest.mark.parametrize("method,margs", [
    ("GPEIChooser", "mcmc_iters=4"),
    ("GPEIOptChooser", "mcmc_iters=3,burnin=3,grid_subset=3,use_multiprocessing=0"),
    ("GPEIperSecChooser", "mcmc_iters=3,burnin=3,grid_subset=3,use_multiprocessing=0"),
])
def test_reference_lite_loop_on_libspx_equals_oracle_engine(tmp_path, method, margs):
    from spearmint_amd.engine import Engine
    from tests.helpers import OracleEngine
    runs = {}
    for tag, factory in (("gpu", None), ("oracle", OracleEngine)):
        work = str(tmp_path / tag)
        with zipfile.ZipFile(LITE_ZIP) as z:
            z.extractall(work)
        runs[tag] = _run_loop(work, method, margs, 11, factory)
    gpu_lines, gpu_engines = runs["gpu"]
    ora_lines, _ = runs["oracle"]
    # the GPU run really went through libspx (a ctypes handle, not the stand-in) ...
    assert gpu_engines and all(isinstance(e, Engine) for e in gpu_engines)
    assert len(gpu_lines) == len(ora_lines) == 7
    assert gpu_lines[-1] != gpu_lines[-2]
    # ... and proposed what the oracle-engine run proposed: grid points print identically, refined
    # off-grid points (GPEIOpt / perSec) agree to the L-BFGS tolerance
    for a, b in zip(gpu_lines, ora_lines):
        xa = np.array([float(v) for v in a.split()[2:]])
        xb = np.array([float(v) for v in b.split()[2:]])
        assert xa.shape == xb.shape == (2,)
        np.testing.assert_allclose(xa, xb, rtol=0, atol=2e-4)
    for e in gpu_engines:
        e.close()

""""Driver untouched" integration: the reference's own spearmint-lite loop
(converted to Python 3 mechanically, at test time, in a tmp dir -- the reference
tree is never copied into the repo) drives OUR chooser through its plugin API.

Runs only where /root/reference exists (the build container).  There is no GPU
there, so the test-only OracleEngine is injected in place of the HIP engine;
what is under test is the plumbing: module discovery by name, init(), next()'s
inputs as spearmint-lite builds them (np.matrix rows, (n,1) durations), the
int / (int, ndarray) return contract, and the state pickle name."""
import importlib
import os
import re
import shutil
import subprocess
import sys
import types

import numpy as np
import pytest

REF = os.environ.get("SPEARMINT_REFERENCE", "/root/reference")
LITE = os.path.join(REF, "spearmint-lite")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(LITE), reason="reference tree not present")


def _branin(x):
    a = x[0] * 15
    b = (x[1] * 15) - 5
    return float(np.square(b - (5.1 / (4 * np.square(np.pi))) * np.square(a) + (5 / np.pi) * a - 6)
                 + 10 * (1 - (1. / (8 * np.pi))) * np.cos(a) + 10)


@pytest.fixture()
def lite(tmp_path, monkeypatch):
    work = tmp_path / "lite"
    shutil.copytree(LITE, str(work))
    subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", str(work)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(str(work)):
        if f.endswith(".py"):
            p = os.path.join(str(work), f)
            src = re.sub(r"from \.(\w+)\s+import", r"from \1 import", open(p).read())
            open(p, "w").write(src)
    shutil.copy(os.path.join(str(work), "spearmint-lite.py"), os.path.join(str(work), "spearmint_lite_main.py"))
    # sys.path as a user would set PYTHONPATH: the lite dir, our dropin shim, our repo
    for p in (ROOT, os.path.join(ROOT, "dropin"), str(work)):
        monkeypatch.syspath_prepend(p)
    for name in [m for m in sys.modules if m == "chooser" or m.startswith("chooser.")]:
        monkeypatch.delitem(sys.modules, name)
    # no GPU in this container: inject the test-only oracle engine
    from tests.helpers import OracleEngine
    import spearmint_amd.engine as eng
    monkeypatch.setattr(eng, "Engine", lambda *a, **k: OracleEngine())
    mod = importlib.import_module("spearmint_lite_main")
    yield mod, work
    for name in [m for m in sys.modules if m == "chooser" or m.startswith("chooser.")
                 or m in ("spearmint_lite_main", "ExperimentGrid", "sobol_lib", "Locker", "util")]:
        sys.modules.pop(name, None)


@pytest.mark.parametrize("method,margs", [
    ("GPEIChooser", "mcmc_iters=3"),
    ("GPEIOptChooser", "mcmc_iters=2,burnin=2,grid_subset=2,use_multiprocessing=0"),
])
def test_unmodified_lite_loop_drives_our_chooser(lite, method, margs):
    mod, work = lite
    expt = os.path.join(str(work), "braninpy")
    res = os.path.join(expt, "results.dat")
    open(res, "w").close()
    opts = types.SimpleNamespace(num_jobs=1, max_finished_jobs=1000, chooser_module=method,
                                 chooser_args=margs, grid_size=200, grid_seed=1,
                                 config_file="config.json", results_file="results.dat")
    np.random.seed(7)
    for it in range(5):
        mod.main_controller(opts, [expt])
        lines = open(res).read().strip().split("\n")
        assert lines[-1].startswith("P P ")
        x = [float(v) for v in lines[-1].split()[2:]]
        assert len(x) == 2 and all(0.0 <= v <= 1.0 for v in x)
        lines[-1] = "%f 1.5 %s" % (_branin(x), " ".join(lines[-1].split()[2:]))   # job "finished"
        open(res, "w").write("\n".join(lines) + "\n")
    # two proposals in one call: the second one sees the first as PENDING (fantasy branch)
    opts.num_jobs = 2
    mod.main_controller(opts, [expt])
    lines = open(res).read().strip().split("\n")
    assert lines[-1].startswith("P P ") and lines[-2].startswith("P P ") and lines[-1] != lines[-2]
    # our module was the one loaded, under the reference's name, and it persisted state under that name
    ch = sys.modules["chooser." + method]
    assert "dropin" in ch.__file__
    if method == "GPEIOptChooser":
        assert os.path.exists(os.path.join(expt, "chooser.GPEIOptChooser.pkl"))
        assert os.path.exists(os.path.join(expt, "chooser.GPEIOptChooser_hyperparameters.txt"))


def test_gpu_sobol_option_rebinds_the_grid_generator(lite, monkeypatch):
    """--method-args gpu_sobol=1: spearmint-lite's ExperimentGrid builds its candidate grid with
    our generator (here the test-only oracle engine stands in for the GPU), and the proposals are
    the ones the reference's own pure-Python generator leads to."""
    mod, work = lite
    from spearmint_amd import sobol as ssob
    from tests.helpers import OracleEngine
    fake = OracleEngine()
    monkeypatch.setattr(ssob, "_engine", fake)
    expt = os.path.join(str(work), "braninpy")
    res = os.path.join(expt, "results.dat")

    def run(margs):
        open(res, "w").close()
        for f in os.listdir(expt):
            if f.endswith(".pkl"):
                os.remove(os.path.join(expt, f))
        opts = types.SimpleNamespace(num_jobs=1, max_finished_jobs=1000, chooser_module="GPEIChooser",
                                     chooser_args=margs, grid_size=150, grid_seed=1,
                                     config_file="config.json", results_file="results.dat")
        np.random.seed(11)
        out = []
        for it in range(4):
            mod.main_controller(opts, [expt])
            lines = open(res).read().strip().split("\n")
            out.append(lines[-1])
            x = [float(v) for v in lines[-1].split()[2:]]
            lines[-1] = "%f 1.5 %s" % (_branin(x), " ".join(lines[-1].split()[2:]))
            open(res, "w").write("\n".join(lines) + "\n")
        return out

    try:
        plain = run("mcmc_iters=2")
        assert not any(c[0] == "sobol_grid" for c in fake.calls)
        patched = run("mcmc_iters=2,gpu_sobol=1")
        gen = [c for c in fake.calls if c[0] == "sobol_grid"]
        assert gen[0] == ("sobol_grid", 40, 16, 3)          # the table-identification probe
        assert len(gen) >= 5 and all(c[1] == 2 and c[2] == 150 for c in gen[1:])
        assert patched == plain
        assert getattr(sys.modules["ExperimentGrid"].i4_sobol_generate, "_spx_gpu", False)
    finally:
        ssob.uninstall()

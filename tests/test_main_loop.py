"""The reference's PRIMARY driver, untouched, on our drop-in choosers (CPU box: test-only oracle engine).

`spearmint/spearmint/main.py` -- `main()` (main.py:147-180) and `attempt_dispatch` (:187-284) with everything they
pull in: `ExperimentGrid` (its pickle reloaded on every dispatch, ExperimentGrid.py:43-196), the Job protobuf
(`spearmint_pb2.py`, `helpers.save_job`), `driver/local.py` forking `runner.job_runner`, which imports and runs
`examples/braninpy/branin.py` -- is converted for Python 3 by `oracle/ref_py3.convert_main_tree` (lib2to3 + the
mechanical edits listed there) into a scratch tree, never into the repository.  The three chooser files of that tree are
replaced by `dropin/chooser/*.py` and `tests/run_reference_main.py` runs the driver in a
process of its own.  What must come out is what the reference's own choosers produced under the same seed
(`tests/golden/main_loop.npz`, made by `oracle/make_golden.py gen_main_loop` with `--engine reference`):

  g  BASELINE configs[0] literally: the `main()` loop, `--method=GPEIChooser --method-args=mcmc_iters=10
     --grid-size=1000 --grid-seed=1`, until eight jobs have finished -- the same eight job ids.
  o  `attempt_dispatch` eight times with `--method=GPEIOptChooser --max-concurrent=2` and jobs held RUNNING across
     dispatches: the pending (fantasy) branch three times, the "(id, candidate)" return -> `add_to_grid` five times, the
     "maximum number of jobs pending" early return once -- the same ids, the same new points (L-BFGS tolerance).

The GPU box runs the same two scenarios on libspx.so: tests/test_gpu_i_main_loop.py."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import make_golden as mg
from oracle import ref_py3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "main_loop.npz")


def tree_source():
    """The packed driver (build output of __graft_entry__.build()) or, in the build container, a fresh conversion."""
    if os.path.isfile(ref_py3.MAIN_ZIP):
        return ref_py3.MAIN_ZIP
    if os.path.isdir(os.path.join(ref_py3.REF_ROOT, "spearmint")):
        return None
    pytest.skip("neither oracle/_ref/main_py3.zip nor the reference tree is present")


def check_run(tag, rec, engine_marker):
    g = np.load(GOLDEN)
    method = "GPEIChooser" if tag == "g" else "GPEIOptChooser"
    # OUR module was the one main.py loaded, under the reference's name, from the reference's chooser directory
    assert rec["chooser_file"] == os.path.join(rec["tree"], "spearmint", "chooser", method + ".py")
    assert "spearmint_amd" in open(rec["chooser_file"]).read()
    assert rec["engines_made"] and all(e == engine_marker for e in rec["engines_made"]), rec["engines_made"]
    # the sequence of experiments: job ids identical, grid points identical, refined points to the L-BFGS tolerance
    assert rec["order"] == list(g[tag + "_order"]), (rec["order"], g[tag + "_order"])
    pts = np.array([rec["points"][str(j)] for j in rec["order"]])
    on_grid = g[tag + "_order"] < 1000
    assert np.array_equal(pts[on_grid], g[tag + "_points"][on_grid])
    assert np.allclose(pts, g[tag + "_points"], rtol=0, atol=2e-5)
    vals = np.array([rec["values"][str(j)] for j in rec["order"]])
    assert np.allclose(vals, g[tag + "_values"], rtol=1e-3, atol=1e-3)
    assert rec["grid_rows"] == int(g[tag + "_grid_rows"])
    assert all(rec["status"][str(j)] == 3 for j in rec["order"])          # COMPLETE_STATE, set by the forked job_runner
    if tag == "o":
        steps = rec["steps"]
        assert [(-1 if s["job"] is None else s["job"]) for s in steps] == list(g["o_step_job"])
        assert [len(s["pending_before"]) for s in steps] == list(g["o_step_npending"])
        assert [s["complete_before"] for s in steps] == list(g["o_step_ncomplete"])
        assert all(s["returned"] for s in steps)
        assert "Maximum number of jobs (2) pending" in rec["stderr"]                    # main.py:240-242
        assert rec["stderr"].count("selected job") == 7
    else:
        assert rec["steps"]["held_back"] >= 0 and rec["steps"]["dispatches"] >= 9       # 8 dispatches + the final look
        assert "Maximum number of finished jobs (8) reached" in rec["stderr"]           # main.py:231-234
    # files under the reference's names ...
    expt = os.path.join(rec["tree"], "examples", "braninpy")
    want = {"expt-grid.pkl", "chooser.%s.pkl" % method, "trace.csv", "best_job_and_result.txt", "jobs", "output"}
    if tag == "o":
        want.add("chooser.GPEIOptChooser_hyperparameters.txt")
    assert want <= set(rec["files"]), rec["files"]
    assert len(os.listdir(os.path.join(expt, "jobs"))) == len(rec["order"])
    assert len(os.listdir(os.path.join(expt, "output"))) == len(rec["order"])
    # ... which the reference's own bin/cleanup removes (its glob: *GP*Chooser*.pkl *Chooser*hyperparameters.txt ...)
    subprocess.check_call(["bash", os.path.join(rec["tree"], "bin", "cleanup"), expt])
    left = set(os.listdir(expt)) - {"__pycache__"}
    assert left == {"branin.py", "config.pb", "jobs", "output"}, left
    assert os.listdir(os.path.join(expt, "jobs")) == [] and os.listdir(os.path.join(expt, "output")) == []


@pytest.mark.parametrize("tag", ["g", "o"])
def test_reference_main_loop_drives_our_chooser_like_its_own(tag):
    src = tree_source()
    with tempfile.TemporaryDirectory(prefix="spx_main_loop_") as work:
        rec = mg.run_main_loop(tag, "oracle", work, zip_path=src)
        check_run(tag, rec, "oracle")

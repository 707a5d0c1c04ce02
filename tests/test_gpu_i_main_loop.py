"""The reference's PRIMARY driver, untouched, on libspx.so: `main.py`'s `main()` loop (BASELINE configs[0]: examples/braninpy,
GPEIChooser, grid 1000, mcmc_iters=10) and its `attempt_dispatch` with `--max-concurrent=2` (GPEIOptChooser: pending
branch, tuple return -> `add_to_grid`), exactly as tests/test_main_loop.py describes, but with the HIP engine: the job ids
and new points must be the ones the reference's own choosers produced (tests/golden/main_loop.npz).

The driver arrives as `oracle/_ref/main_py3.zip` (built by `__graft_entry__.build()` where /root/reference exists;
git-ignored, shipped with the tree like a built .so).  The forked job processes (driver/local.py:16) inherit the parent's
HIP engine object and never touch it (process model)."""
import os
import tempfile

import pytest

from oracle import make_golden as mg
from oracle import ref_py3
from tests.test_main_loop import check_run

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isfile(ref_py3.MAIN_ZIP),
                                 reason="oracle/_ref/main_py3.zip absent: run __graft_entry__.build() where /root/reference exists")]


@pytest.mark.parametrize("tag", ["g", "o"])
def test_reference_main_loop_on_libspx_reproduces_the_reference_choosers_jobs(tag):
    with tempfile.TemporaryDirectory(prefix="spx_main_loop_gpu_") as work:
        rec = mg.run_main_loop(tag, "hip", work, zip_path=ref_py3.MAIN_ZIP)
        check_run(tag, rec, "spearmint_amd.engine.Engine")

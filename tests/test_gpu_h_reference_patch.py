"""INTEGRATION.md section 2, executed: the ctypes stub a Spearmint maintainer would paste into the REFERENCE's own
chooser/GPEIOptChooser.py is taken from the document as it is written there (the fenced python block), bound to the built
libspx.so, and patched onto the reference's own (lib2to3-converted) GPEIOptChooser class -- whose sampler, priors, state
pickles and L-BFGS refinement stay the reference's.  A seeded next() of that patched reference chooser must give the proposal
and the hyper-parameter samples of the unpatched reference (tests/golden/chooser_next.npz): the drop-in boundary is the C ABI,
and it fits the reference's own call site (S/chooser/GPEIOptChooser.py:331-341, called at :269 and :293).

The converted reference comes from oracle/_ref/chooser_py3.zip (built by __graft_entry__.build() where /root/reference exists,
shipped with the tree) -- test infrastructure, like every use of oracle/."""
import os
import re

import numpy as np
import numpy.random as npr
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def integration_stub_source():
    """The first ```python block of INTEGRATION.md section 2, with the library path filled in."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The ctypes stub"):]
    block = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "spx_ei_grid" in block and "class GPEIOptChooser" in block
    return block.replace("/path/to/libspx.so", os.path.join(ROOT, "spearmint_amd", "libspx.so"))


def integration_sampler_stub_source():
    """The second ```python block of INTEGRATION.md section 2 (the sampler behind the boundary), appended to the first."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The ctypes stub"):]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) >= 2 and "spx_sample_hypers" in blocks[1] and "class GPEIOptChooserSampler" in blocks[1]
    return blocks[1]


def test_the_documented_sampler_stub_on_the_reference_chooser_reproduces_the_reference(golden_dir, tmp_path):
    """Round 6: the reference's own GPEIOptChooser with BOTH documented methods patched in -- `ei_over_hypers` on spx_ei_grid and
    `sample_hypers` on spx_sample_hypers (the library's native slice sampler, fed by numpy's global generator) -- gives the
    reference's golden hyper samples and proposal, and leaves numpy's generator where the unpatched reference leaves it."""
    from oracle import ref_py3
    mods = ref_py3.load() if ref_py3.available() else ref_py3.load_shipped()
    if mods is None:
        pytest.skip("oracle/_ref/chooser_py3.zip not built (run __graft_entry__.build() where /root/reference exists)")
    ns = {}
    exec(compile(integration_stub_source(), "INTEGRATION.md#2", "exec"), ns)
    exec(compile(integration_sampler_stub_source(), "INTEGRATION.md#2b", "exec"), ns)
    ref_mod = mods["GPEIOptChooser"]

    class Patched(ref_mod.GPEIOptChooser):
        _spx_handle = None
        _spx_hist = None
        ei_over_hypers = ns["GPEIOptChooser"].ei_over_hypers
        __getstate__ = ns["GPEIOptChooser"].__getstate__
        sample_hypers = ns["GPEIOptChooserSampler"].sample_hypers

    g = np.load(os.path.join(golden_dir, "chooser_next.npz"), allow_pickle=True)
    args = {"mcmc_iters": "4", "burnin": "6", "grid_subset": "5", "use_multiprocessing": "0"}
    (tmp_path / "patched").mkdir()
    ch = Patched(str(tmp_path / "patched"), **args)
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    state_patched = npr.get_state()
    hypers = np.array([np.hstack(h) for h in ch.hyper_samples])
    assert np.allclose(hypers, g["opt_hypers"], rtol=1e-9)                 # the library's sampler: the reference's chain
    assert int(job[0] if isinstance(job, tuple) else job) == int(g["opt_index"])
    if int(g["opt_is_new"]):
        assert np.allclose(job[1], g["opt_point"], atol=1e-6)
    # the unpatched reference from the same seed: same generator state afterwards
    (tmp_path / "plain").mkdir()
    ref = ref_mod.GPEIOptChooser(str(tmp_path / "plain"), **args)
    npr.seed(int(g["opt_seed"]))
    ref.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    state_ref = npr.get_state()
    assert np.array_equal(state_patched[1], state_ref[1]) and state_patched[2:] == state_ref[2:]


def test_the_documented_stub_on_the_reference_chooser_reproduces_the_reference(golden_dir, tmp_path):
    from oracle import ref_py3
    mods = ref_py3.load() if ref_py3.available() else ref_py3.load_shipped()
    if mods is None:
        pytest.skip("oracle/_ref/chooser_py3.zip not built (run __graft_entry__.build() where /root/reference exists)")
    ns = {}
    exec(compile(integration_stub_source(), "INTEGRATION.md#2", "exec"), ns)
    stub = ns["GPEIOptChooser"]
    ref_mod = mods["GPEIOptChooser"]

    class Patched(ref_mod.GPEIOptChooser):            # the reference's class, two methods replaced by the document's
        _spx_handle = None
        ei_over_hypers = stub.ei_over_hypers
        __getstate__ = stub.__getstate__

    g = np.load(os.path.join(golden_dir, "chooser_next.npz"), allow_pickle=True)
    args = ref_mod.util.unpack_args("mcmc_iters=4,burnin=6,grid_subset=5,use_multiprocessing=0") \
        if hasattr(ref_mod, "util") else {"mcmc_iters": "4", "burnin": "6", "grid_subset": "5", "use_multiprocessing": "0"}
    ch = Patched(str(tmp_path), **args)
    npr.seed(int(g["opt_seed"]))
    job = ch.next(g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    hypers = np.array([np.hstack(h) for h in ch.hyper_samples])
    assert np.allclose(hypers, g["opt_hypers"], rtol=1e-9)                 # the reference's own sampler, untouched
    assert ch._spx_handle is not None and ch._spx_handle.value            # the GPU path really ran
    if int(g["opt_is_new"]):
        assert isinstance(job, tuple) and int(job[0]) == int(g["opt_index"])
        assert np.allclose(job[1], g["opt_point"], atol=1e-6)
    else:
        assert int(job) == int(g["opt_index"])
    # EI of the grid through the stub == the reference's own numpy/scipy ei_over_hypers on the same state
    comp = g["grid"][g["complete"]]
    cand = g["grid"][g["candidates"]]
    vals = g["values"][g["complete"]]
    pend = g["grid"][g["pending"]]
    got = ch.ei_over_hypers(comp, pend, cand, vals)
    want = ref_mod.GPEIOptChooser.ei_over_hypers(ch, comp, pend, cand, vals)
    big = want > 1e-280
    assert got.shape == want.shape and np.max(np.abs(got[big] - want[big]) / want[big]) < 1e-7
    assert int(np.argmax(np.mean(got, axis=1))) == int(np.argmax(np.mean(want, axis=1)))
    state = ch.__getstate__()                                              # the stub's __getstate__: what a Pool worker receives
    assert state["_spx_handle"] is None and "hyper_samples" in state

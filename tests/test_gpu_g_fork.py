"""Fork-after-init safety.  The reference's drivers fork from the long-lived main loop AFTER the chooser has
run -- job processes (`spearmint/driver/local.py:9-44`: multiprocessing.Process around the job runner) and the
status web server (`main.py:126-141`).  A forked child inherits the chooser object, engine handle included,
but no usable HIP context: it must be able to run a no-GPU job and exit without disturbing the parent, a
child that does touch the engine must get a clear error (not a hang or a crash of the parent), and the
parent's next proposal must be unaffected."""
import multiprocessing
import os

import numpy as np
import numpy.random as npr
import pytest

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=True)


def _job_without_gpu(q):
    # what a Spearmint job process does: run user code, report, exit (interpreter teardown runs Engine.__del__
    # on the inherited copy, which must not destroy the parent's handle)
    q.put(("job", float(np.sum(np.arange(10.0)))))


def _job_touching_engine(ch, q):
    from spearmint_amd.engine import SpxError
    try:
        ch.engine().set_option("timing", 0)
        q.put(("touch", "no error"))
    except SpxError as e:
        q.put(("touch", "SpxError: %s" % e))
    except Exception as e:     # pragma: no cover
        q.put(("touch", "%s: %s" % (type(e).__name__, e)))


def test_fork_after_next_leaves_the_parent_intact(golden_dir, tmp_path):
    from spearmint_amd.chooser import GPEIChooser
    g = _g(golden_dir, "branin_c1.npz")
    args = (g["grid"], g["values"], g["durations"], g["candidates"], g["pending"], g["complete"])
    ch = GPEIChooser.init(str(tmp_path), "mcmc_iters=10")
    npr.seed(int(g["seed"]))
    assert ch.next(*args) == int(g["job"])          # HIP context, streams and buffers exist from here on
    assert ch.engine().owned_by_this_process()
    ctx = multiprocessing.get_context("fork")
    q = ctx.Queue()
    p1 = ctx.Process(target=_job_without_gpu, args=(q,))
    p1.start()
    p2 = ctx.Process(target=_job_touching_engine, args=(ch, q))
    p2.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    p1.join(120)
    p2.join(120)
    assert p1.exitcode == 0 and p2.exitcode == 0
    assert got["job"] == 45.0
    assert got["touch"].startswith("SpxError") and "pid" in got["touch"]
    # the parent goes on as if nothing had happened: same state file, same chain, same proposal as a fresh run
    os.makedirs(str(tmp_path / "again"))
    ch2 = GPEIChooser.init(str(tmp_path / "again"), "mcmc_iters=10")
    npr.seed(int(g["seed"]))
    assert ch2.next(*args) == int(g["job"])
    # and the ORIGINAL chooser's engine still works after the children came and went
    eng = ch.engine()
    comp = g["grid"][g["complete"]]
    eng.set_observations(comp, g["values"][g["complete"]])
    eng.set_hypers(np.concatenate(([ch.mean, ch.noise, ch.amp2], ch.ls))[None, :])
    assert np.isfinite(eng.gp_logprob()[0])

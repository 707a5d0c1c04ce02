"""Local refinement of the best grid points: ``grid_subset`` independent L-BFGS-B problems
(GPEIOptChooser.py:265-291) whose objective evaluations are served by ONE batched GPU call.

The reference runs the problems one after the other (or in a fork-based Pool, which cannot share a
HIP context); each objective evaluation there is a full per-draw factorisation.  Here every problem
is an unmodified ``scipy.optimize.fmin_l_bfgs_b`` instance on its own host thread; whenever all
instances that are still running wait for an objective value, the waiting points go to the GPU in
one ``spx_ei_grad_batch`` call.  A point's value does not depend on the other points of a call, so
each instance sees exactly the numbers a serial run would give it and returns the same optimum.

Thread model: the instances never run concurrently with each other or with the dispatcher in any way that
matters -- an instance only executes scipy code between two objective calls, and the dispatcher only acts when
every live instance waits -- but several fmin_l_bfgs_b calls are in flight at once, one per thread.  That needs a
re-entrant fmin_l_bfgs_b: true for scipy >= 1.5 (the Fortran driver is called with explicit work arrays per call;
tested here on 1.15.3, bit-equal to the serial loop in tests/test_host_logic.py).  ``serial=True`` (or the
environment variable SPX_REFINE_SERIAL=1) runs the instances one after the other instead.

Python 2/3 common subset.
"""
from __future__ import absolute_import, print_function

import os
import threading

import numpy as np
import scipy.optimize as spo


def lbfgs_many(eval_batch, points, bounds, log=None, serial=None):
    """Minimise the objective from every row of ``points`` (P, D) with L-BFGS-B.

    eval_batch(X[k, D]) -> (f[k], grad[k, D]) evaluates any subset of the problems at once.
    Returns the (P, D) array of optima, row i being what
    ``fmin_l_bfgs_b(lambda x: eval_batch(x[None])[0][0], points[i], bounds=bounds)`` returns."""
    pts = np.array(points, dtype=float, copy=True)
    n = pts.shape[0]
    if n == 0:
        return pts
    if serial is None:
        serial = os.environ.get("SPX_REFINE_SERIAL", "0") not in ("", "0")
    if serial:
        out = pts.copy()
        for i in range(n):
            def one(x):
                f, g = eval_batch(np.asarray(x, dtype=float)[None, :])
                return float(f[0]), np.array(g[0], dtype=float, copy=True)   # as the threaded path hands them over
            out[i, :] = spo.fmin_l_bfgs_b(one, pts[i, :].flatten(), bounds=bounds, disp=0)[0]
        return out
    cv = threading.Condition()
    state = {"live": n}
    req, res, errs = {}, {}, []
    out = pts.copy()

    def worker(i):
        def objective(x):
            with cv:
                req[i] = np.array(x, dtype=float, copy=True)
                cv.notify_all()
                while i not in res:
                    cv.wait()
                r = res[i] if state.get("abort") is res[i] else res.pop(i)
            if isinstance(r, BaseException):
                raise r
            return r

        try:
            out[i, :] = spo.fmin_l_bfgs_b(objective, pts[i, :].flatten(), bounds=bounds, disp=0)[0]
        except BaseException as ex:   # re-raised in the caller's thread
            errs.append(ex)
        finally:
            with cv:
                state["live"] -= 1
                cv.notify_all()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    for t in threads:
        t.daemon = True
        t.start()
    calls = 0
    try:
        with cv:
            while state["live"] > 0:
                while state["live"] > 0 and len(req) < state["live"]:
                    cv.wait()
                if state["live"] == 0:
                    break
                ids = sorted(req)
                X = np.vstack([req[i] for i in ids])
                req.clear()
                try:
                    f, g = eval_batch(X)
                    for k, i in enumerate(ids):
                        res[i] = (float(f[k]), np.array(g[k], dtype=float, copy=True))
                except Exception as ex:          # the objective failed: every waiting instance re-raises it
                    for i in ids:
                        res[i] = ex
                calls += 1
                cv.notify_all()
    finally:
        # Leaving the dispatch loop abnormally (KeyboardInterrupt in cv.wait(), an error in eval_batch's own
        # plumbing) must not strand the instances on the condition variable, holding `comp` and the engine callback
        # alive: every instance still running gets an abort exception as its next objective value and unwinds.
        with cv:
            if state["live"] > 0:
                abort = RuntimeError("lbfgs_many: dispatch loop aborted")
                state["abort"] = abort
                for i in range(n):
                    res.setdefault(i, abort)
                cv.notify_all()
        for t in threads:
            t.join()
    if errs and not state.get("abort"):
        raise errs[0]
    if log is not None:
        log("refined %d points with %d batched objective calls" % (n, calls))
    return out

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

Lock-step form (round 6, the default where it applies): the Python loop of scipy's own driver (`_minimize_lbfgsb`: call
`_lbfgsb.setulb` until it asks for f and g, evaluate, repeat) is run for all instances in ONE thread -- every instance is
advanced to its next request, the waiting points go to the GPU in one call -- so no thread is started and no condition
variable is passed around (at N = 256 the threads cost 21 of the refinement's 22 ms; the GPU calls 0.7 ms).  It drives the
same compiled L-BFGS-B code with the same arguments, work arrays and defaults as `fmin_l_bfgs_b(..., bounds=bounds)`, but
through scipy's private `_lbfgsb.setulb`, whose signature differs between scipy versions: it is used only if, in this
process, it reproduces the public `fmin_l_bfgs_b` bit for bit on a fixed test problem (`lockstep_ok()`); otherwise the
threaded form runs.  SPX_REFINE_THREADS=1 forces the threaded form.

Python 2/3 common subset.
"""
from __future__ import absolute_import, print_function

import os
import threading

import numpy as np
import scipy.optimize as spo


_LOCKSTEP = {}


def _lockstep(eval_batch, pts, bounds):
    """All L-BFGS-B instances advanced in one thread (scipy's `_minimize_lbfgsb` loop, per instance).  Returns the (P, D)
    optima and the number of batched objective calls."""
    from scipy.optimize import _lbfgsb
    n_pts, n = pts.shape
    m, factr, pgtol, maxls, maxfun, maxiter = 10, 1e7, 1e-5, 20, 15000, 15000      # fmin_l_bfgs_b's defaults
    factr = (factr * np.finfo(float).eps) / np.finfo(float).eps                    # (as fmin_l_bfgs_b -> _minimize_lbfgsb pass it)
    lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=float)
    up = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=float)
    nbd = np.zeros(n, np.int32)
    low_bnd = np.zeros(n, np.float64)
    upper_bnd = np.zeros(n, np.float64)
    for i in range(n):
        has_l, has_u = not np.isinf(lo[i]), not np.isinf(up[i])
        if has_l:
            low_bnd[i] = lo[i]
        if has_u:
            upper_bnd[i] = up[i]
        nbd[i] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[(has_l, has_u)]
    st = []
    for i in range(n_pts):
        st.append({"x": np.array(np.clip(pts[i], lo, up), dtype=np.float64), "f": 0.0, "g": np.zeros(n, np.float64),
                   "wa": np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64), "iwa": np.zeros(3 * n, np.int32),
                   "task": np.zeros(2, np.int32), "ln_task": np.zeros(2, np.int32), "lsave": np.zeros(4, np.int32),
                   "isave": np.zeros(44, np.int32), "dsave": np.zeros(29, np.float64), "nfev": 0, "nit": 0})
    live = list(range(n_pts))
    calls = 0
    while live:
        need, still = [], []
        for i in live:
            s = st[i]
            while True:
                _lbfgsb.setulb(m, s["x"], low_bnd, upper_bnd, nbd, s["f"], s["g"], factr, pgtol, s["wa"], s["iwa"],
                               s["task"], s["lsave"], s["isave"], s["dsave"], maxls, s["ln_task"])
                t = s["task"][0]
                if t == 3:                       # wants f and g at x
                    need.append(i)
                    still.append(i)
                    break
                if t == 1:                       # new iteration
                    s["nit"] += 1
                    if s["nit"] >= maxiter:
                        s["task"][0], s["task"][1] = 5, 504
                    elif s["nfev"] > maxfun:
                        s["task"][0], s["task"][1] = 5, 502
                    continue
                break                            # converged / stopped
        live = still
        if need:
            X = np.vstack([st[i]["x"] for i in need])
            f, g = eval_batch(X)
            calls += 1
            for k, i in enumerate(need):
                st[i]["f"] = float(f[k])
                st[i]["g"] = np.array(g[k], dtype=np.float64, copy=True)
                st[i]["nfev"] += 1
    return np.vstack([s["x"] for s in st]), calls


def lockstep_ok():
    """True if, in this process, _lockstep reproduces scipy's public fmin_l_bfgs_b bit for bit on a fixed bounded test
    problem (checked once; the private setulb interface is scipy-version specific)."""
    if "ok" not in _LOCKSTEP:
        try:
            A = np.array([[3.0, 0.4, -0.2], [0.4, 2.0, 0.3], [-0.2, 0.3, 1.5]])
            c = np.array([0.9, -0.3, 1.4])

            def fg(x):
                d = x - c
                q = A.dot(d)
                return float(0.5 * d.dot(q) + 0.1 * np.sum(np.cos(3.0 * x))), q - 0.3 * np.sin(3.0 * x)

            def batch(X):
                r = [fg(x) for x in X]
                return np.array([v[0] for v in r]), np.array([v[1] for v in r])
            pts = np.array([[0.2, 0.8, 0.5], [1.3, -0.2, 0.1], [0.0, 0.0, 1.0]])
            bounds = [(0, 1)] * 3
            got, _ = _lockstep(batch, pts, bounds)
            want = np.array([spo.fmin_l_bfgs_b(fg, p.copy(), bounds=bounds, disp=0)[0] for p in pts])
            _LOCKSTEP["ok"] = bool(np.array_equal(got, want))
        except Exception:
            _LOCKSTEP["ok"] = False
    return _LOCKSTEP["ok"]


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
    if os.environ.get("SPX_REFINE_THREADS", "0") in ("", "0") and lockstep_ok():
        out, calls = _lockstep(eval_batch, pts, bounds)
        if log is not None:
            log("refined %d points with %d batched objective calls" % (n, calls))
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

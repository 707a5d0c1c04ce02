"""Test-only helpers: an oracle-backed stand-in for the GPU engine so the host
logic of the choosers can be exercised on a CPU-only box.  It lives under
tests/ and is injected explicitly (chooser._eng = OracleEngine()); the product
package has no such fallback."""
import numpy as np

from oracle import gp_ei_oracle as orc


def sample_hypers_with(logprob_rows, cfg, hyper, hist, rng_state=None, lib=None):
    """spx_sample_hypers_with: the library's sampler on a caller-supplied log-likelihood
    `logprob_rows(rows[n, 3 + D]) -> lp[n]` (-inf = not positive definite).  No GPU, no handle -- the CPU tests' way to
    hold the native sampler to the reference's chain.  It lives under tests/: the product package binds no host-evaluator
    path (spearmint_amd/engine.py declares the symbol's prototype with the rest of include/spx.h and never calls it)."""
    import numpy.random as npr
    from spearmint_amd.engine import (load_library, RngState, LOGPROB_FN, SPX_ERR_ARG, _dp, _c_int64_p, _sampler_result)
    import ctypes
    lib = load_library(lib)
    D = int(cfg.D)
    failure = []

    def cb(ctx, rows_p, n, out_p):
        try:
            rows = np.ctypeslib.as_array(rows_p, shape=(n, 3 + D)).copy()
            lp = np.asarray(logprob_rows(rows), dtype=np.float64)
            for i in range(n):
                out_p[i] = lp[i]
            return 0
        except BaseException as ex:        # must not unwind through C
            failure.append(ex)
            return SPX_ERR_ARG
    rng = RngState.from_numpy() if rng_state is None else rng_state
    rows = np.empty((int(cfg.n_iter), 3 + D))
    stats = np.zeros(41, dtype=np.int64)
    rc = lib.spx_sample_hypers_with(LOGPROB_FN(cb), None, ctypes.byref(cfg), ctypes.byref(rng), _dp(hyper), _dp(rows),
                                    _dp(hist), stats.ctypes.data_as(_c_int64_p))
    if rng_state is None:
        npr.set_state(rng.to_numpy())
    if failure:
        raise failure[0]
    return _sampler_result(lib, rc, rows, stats)


class OracleEngine(object):
    def __init__(self, covar="Matern52"):
        self.calls = []
        self.native_calls = []   # rows per log-likelihood batch of the native sampler
        self.fant = None
        self.covar = covar

    def set_covar(self, name):
        self.covar = name

    def last_warning(self):
        return None

    # -- resident-data mode (what the pending path of the choosers uses) ------------
    def set_observations(self, comp, vals):
        self.comp, self.vals, self.fant = np.asarray(comp, float), np.asarray(vals, float), None

    def set_candidates(self, cand, index_base=0):
        self.cand = np.asarray(cand, float)

    def set_hypers(self, hypers):
        self.hypers, self.fant = np.atleast_2d(hypers), None
        self._time_set = False          # spx_set_hypers drops the time model, as the library does

    def set_time_model(self, log_durs, time_hypers):
        self.log_durs, self.time_hypers = np.asarray(log_durs, float), np.atleast_2d(time_hypers)
        self._time_set = True

    def get_time_mean(self, draw):
        import scipy.linalg as spla
        t_mean, t_noise, t_amp2, t_ls = orc.unpack_hyper(self.time_hypers[draw])
        chol = spla.cholesky(orc.cov(t_amp2, t_ls, self.comp) + t_noise * np.eye(len(self.comp)), lower=True)
        t_alpha = spla.cho_solve((chol, True), self.log_durs - t_mean)
        return np.exp(np.dot(orc.cov(t_amp2, t_ls, self.comp, self.cand).T, t_alpha) + t_mean)

    def factor(self):
        self.chols = [orc.posterior(self.comp, self.vals, h)[1] for h in self.hypers]
        # what spx_ei_grad_batch works against from here on: this factorisation (and its time model, if one is set)
        self._last_comp, self._last_vals, self._last_rows = self.comp, self.vals, self.hypers
        self._last_time = (self.log_durs, self.time_hypers) if getattr(self, "_time_set", False) else None

    def gp_logprob(self):
        """The sampler's data term for every resident hyper row [mean, noise, amp2, ls...] (-inf where the covariance
        is not positive definite), from the oracle's restatement of GPEIChooser.py:281-285."""
        out = np.empty(self.hypers.shape[0])
        for k, h in enumerate(self.hypers):
            try:
                out[k] = orc.gp_logprob(self.comp, self.vals, h[0], h[2], h[1], h[3:])
            except np.linalg.LinAlgError:
                out[k] = -np.inf
        return out

    def sample_hypers(self, cfg, hyper, hist, rng_state=None):
        """Engine.sample_hypers on a box without a GPU: libspx's OWN sampler (spx_sample_hypers_with: the same C++ control
        flow, priors, speculation and random stream the GPU path runs) with the oracle's data term as the log-likelihood
        callback -- so the CPU tests drive the native sampler through the choosers exactly as the GPU path does."""

        def rows_lp(rows):
            self.set_hypers(rows)
            self.native_calls.append(len(rows))
            return self.gp_logprob()
        return sample_hypers_with(rows_lp, cfg, hyper, hist, rng_state=rng_state)

    def get_factor(self, draw, want_K=True, want_L=True, want_alpha=True):
        return None, self.chols[draw], None

    def get_factor_rows(self, draw, row0, nrows, want_gamma=True):
        import scipy.linalg as spla
        chol = self.chols[draw]
        gam = spla.solve_triangular(chol, self.vals - self.hypers[draw][0], lower=True) if want_gamma else None
        return np.array(chol[row0:row0 + nrows, :]), gam

    def set_fantasies(self, fant, bests):
        self.fant, self.bests = np.asarray(fant, float), np.asarray(bests, float)

    def ei_step(self, flags=0):
        self.fant = None            # a new factorisation drops the fantasies, as spx_factor / spx_ei_step do
        self.factor()
        self.ei_run(flags)

    def ei_run(self, flags=0):
        H = self.hypers.shape[0]
        if self.fant is None:
            ei = orc.ei_over_hypers(self.comp, self.cand, self.vals, self.hypers)
        else:
            ei = np.stack([orc.compute_ei_fantasies(self.comp, self.cand, self.hypers[h], self.fant[h], self.bests[h])
                           for h in range(H)], axis=1)
        self._ei = ei
        self.calls.append(("ei_run", self.cand.shape[0], H, None if self.fant is None else self.fant.shape[2]))

    def best(self):
        m = np.mean(self._ei, axis=1)
        i = int(np.argmax(m))
        return i, float(m[i])

    def ei_mean(self):
        return np.mean(self._ei, axis=1)

    def ei_draws(self):
        return self._ei

    def ei_grid(self, comp, vals, cand, hypers, want_mean=True, want_draws=False, flags=0):
        ei = orc.ei_over_hypers(comp, cand, vals, hypers)
        self._last_comp, self._last_vals, self._last_rows = comp, vals, np.atleast_2d(hypers)
        self._last_time, self.fant = None, None
        mean = np.mean(ei, axis=1)
        idx = int(np.argmax(mean))
        self.calls.append(("ei_grid", cand.shape[0], np.atleast_2d(hypers).shape[0]))
        return idx, float(mean[idx]), mean, (ei if want_draws else None)

    def sobol_grid(self, dirs, dim, n, skip, fetch=True, as_candidates=False):
        from oracle import sobol_oracle
        self.calls.append(("sobol_grid", int(dim), int(n), int(skip)))
        return np.ascontiguousarray(sobol_oracle.i4_sobol_generate(dim, n, skip, dirs).T), 0.0

    def ei_grad_batch(self, points):
        """The refinement objective from the oracle's restatement of grad_optimize_ei_over_hypers,
        against whatever the last EI pass left "resident" (plain, fantasies, or per second)."""
        points = np.atleast_2d(points)
        f = np.zeros(points.shape[0]); g = np.zeros(points.shape)
        for i, x in enumerate(points):
            if self.fant is not None:
                for h in range(self.hypers.shape[0]):
                    e, gr = orc.grad_optimize_ei_fantasies(x, self.comp, self.hypers[h], self.fant[h], self.bests[h])
                    f[i] += e; g[i] = g[i] + gr
            elif getattr(self, "_last_time", None) is not None:
                f[i], g[i] = orc.grad_optimize_ei_over_hypers(x, self._last_comp, self._last_vals, self._last_rows,
                                                              log_durs=self._last_time[0], time_hypers=self._last_time[1])
            else:
                f[i], g[i] = orc.grad_optimize_ei_over_hypers(x, self._last_comp, self._last_vals, self._last_rows)
        return f, g

    def ei_grad(self, x):
        f, g = self.ei_grad_batch(np.asarray(x)[None, :])
        return float(f[0]), g[0]

    def ei_per_sec_grid(self, comp, vals, log_durs, cand, hypers, time_hypers,
                        want_mean=True, want_draws=False, flags=0):
        ei = orc.ei_per_s_over_hypers(comp, cand, vals, log_durs, hypers, time_hypers)
        self._last_comp, self._last_vals, self._last_rows = comp, vals, np.atleast_2d(hypers)
        self._last_time, self.fant = (log_durs, np.atleast_2d(time_hypers)), None
        mean = np.mean(ei, axis=1)
        idx = int(np.argmax(mean))
        self.calls.append(("ei_per_sec_grid", cand.shape[0], np.atleast_2d(hypers).shape[0]))
        return idx, float(mean[idx]), mean, (ei if want_draws else None)


def _under_covar(fn):
    def wrapped(self, *a, **k):
        with orc.covar(self.covar):
            return fn(self, *a, **k)
    wrapped.__name__ = fn.__name__
    return wrapped


for _name in ("get_time_mean", "factor", "gp_logprob", "ei_run", "ei_grid", "ei_grad_batch", "ei_per_sec_grid"):
    setattr(OracleEngine, _name, _under_covar(getattr(OracleEngine, _name)))


def np_mean_device_order(row):
    """Python mirror of k_mean_over_draws (predict_kernels.hip): numpy's
    pairwise summation order, then / H."""
    def pw(a):
        n = len(a)
        if n < 8:
            res = -0.0
            for x in a:
                res = res + x
            return res
        if n <= 128:
            r = [a[q] for q in range(8)]
            i = 8
            while i < n - (n % 8):
                for q in range(8):
                    r[q] = r[q] + a[i + q]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res = res + a[i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])
    row = [np.float64(x) for x in row]
    return (np.float64(0.0) + pw(row)) / np.float64(len(row))


def branin(x0, x1):
    a = x0 * 15
    b = (x1 * 15) - 5
    return (np.square(b - (5.1 / (4 * np.square(np.pi))) * np.square(a) + (5 / np.pi) * a - 6)
            + 10 * (1 - (1. / (8 * np.pi))) * np.cos(a) + 10)


def run_trajectory(make_chooser, grid, iters, seed):
    """The loop oracle/make_golden.py:run_trajectory drove the REFERENCE's choosers with (a fresh chooser per
    proposal, restarted from its state pickle, new points appended to the grid), here for our choosers."""
    import numpy.random as npr
    grid = np.array(grid, copy=True)
    values = np.zeros(grid.shape[0]) + np.nan
    durations = np.zeros(grid.shape[0]) + np.nan
    done = np.zeros(grid.shape[0], dtype=bool)
    out = []
    for it in range(iters):
        ch = make_chooser()
        npr.seed(seed + it)
        job = ch.next(grid, values, durations, np.nonzero(~done)[0], np.zeros(0, dtype=int), np.nonzero(done)[0])
        if isinstance(job, tuple):
            pt = np.asarray(job[1], dtype=float).ravel()
            grid = np.vstack((grid, pt[None, :]))
            values, durations, done = np.append(values, np.nan), np.append(durations, np.nan), np.append(done, False)
            idx = grid.shape[0] - 1
            out.append((1, idx, pt))
        else:
            idx = int(job)
            out.append((0, idx, grid[idx].copy()))
        values[idx] = branin(grid[idx, 0], grid[idx, 1])
        durations[idx] = 1.0 + 3.0 * grid[idx, 0] + np.sin(5 * grid[idx, 1]) ** 2
        done[idx] = True
        if type(ch).__name__ == "GPEIChooser" and getattr(ch, "D", -1) != -1:
            ch.__del__()
    return out

"""Shared machinery of the MI355X GP-EI choosers.

The three public modules (GPEIChooser, GPEIOptChooser, GPEIperSecChooser) keep
the reference's plugin contract --

    module.init(expt_dir, arg_string) -> chooser
    chooser.next(grid, values, durations, candidates, pending, complete)
        -> int grid index | (int, ndarray) new off-grid point

(spearmint/spearmint/main.py:164-165, :254; spearmint-lite.py:99-100, :187-190)
-- and differ from the reference only in WHERE the EI grid is computed: the
``for mcmc_iter: compute_ei(...)`` / ``ei_over_hypers`` block plus the
``argmax(mean(...))`` is one call into libspx.so (HIP, gfx950) through
``spearmint_amd.engine``.  Hyper-parameter slice sampling, state pickles,
logging and the 20-point L-BFGS refinement stay on the host, as SURVEY.md
section 8(b) lays out.  There is no numpy fallback for the EI grid: if the
library or the GPU is missing, ``next`` raises.

Python 2/3 common subset on purpose (the reference driver is Python 2).
"""
from __future__ import absolute_import, print_function

import os

import numpy as np

from .. import hostgp
from .. import util
from ..helpers import log, pickle_atomically, unpickle
from ..Locker import Locker


def _as_bool(v):
    return bool(int(v))


class GPEIBase(object):
    """State + sampler + GPU dispatch common to the GP-EI choosers."""

    # knobs the subclasses override
    max_ls = 2                     # top-hat prior on length scales
    noise_scale = 0.1              # horseshoe prior scale
    amp2_scale = 1                 # log-normal prior scale
    amp2_prior_on_sqrt = False     # lognormal on sqrt(amp2) (Opt) vs amp2 (GPEIChooser)
    noiseless_checks_mean = True   # mean in [min, max] also in noiseless mode
    state_keys = ("dims", "ls", "amp2", "noise", "mean")
    max_rows_per_call = 32         # spx_gp_logprob: one data-flow launch up to 32 hyper rows (spx_api.hip: do_factor, `rl`)

    def __init__(self, expt_dir, covar="Matern52", mcmc_iters=10, pending_samples=100,
                 noiseless=False, device=0, ndev=1, lib=None, gpu_logprob="auto", gpu_refine="auto",
                 lookahead="auto", follow="auto", sampler="native", gpu_sobol=0, **unused):
        if covar not in hostgp.COVARS:
            # the reference does getattr(gp, covar) here (GPEIChooser.py:52)
            raise AttributeError("no covariance function %r (gp.py has %s)" % (covar, ", ".join(hostgp.COVARS)))
        self.covar = covar        # every kernel of gp.py runs on the GPU: option "covar" of the handle
        self.expt_dir = expt_dir
        self.locker = Locker()
        self.state_pkl = os.path.join(expt_dir, self.__module__ + ".pkl")
        self.mcmc_iters = int(mcmc_iters)
        self.pending_samples = int(pending_samples)
        if not 1 <= self.pending_samples <= 4096:
            raise ValueError("pending_samples must be in 1..4096 (got %d): the GPU path scores every "
                             "candidate against that many fantasies per draw" % self.pending_samples)
        self.noiseless = _as_bool(noiseless)
        self.device = int(device)
        self.ndev = int(ndev)     # GPUs device .. device+ndev-1, candidates sharded over them
        self.lib_path = lib
        # where the slice sampler's log-likelihood (K build + Cholesky + solve per call) runs:
        # "0" host numpy/scipy (the reference's way), "1" libspx on the GPU, "auto" = GPU once
        # the factorisation outweighs the call latency.  On the MI355X box (scripts/dev/logprob_threshold.py, round 3): a GPU call
        # costs 0.067 ms up to N = 128 whether it carries one hyper row or six (one launch for the factorisation, results
        # written into pinned host memory), a host evaluation 0.022 ms at N = 8, 0.030 at 32, 0.049 at 64, 0.081 at 96, 0.22
        # at 128; a slice move needs ~4.4 evaluations one after the other on the host or ~1.3 speculative batches on the GPU:
        # break-even near N = 10, "auto" switches at 16 for the Python batched sampler (round 6: at 2 for the native one, see
        # _use_gpu_logprob).  (N = 2048: 0.58 ms per GPU call vs 250 ms.)
        self.gpu_logprob = str(gpu_logprob)
        # same choice for the EI + gradient objective of the local refinement (spx_ei_grad_batch).  There the GPU wins at
        # every size (scripts/dev/refine_threshold.py, 10 draws: one call for all 20 refinement points 0.06 ms from N = 8 to
        # N = 128; the host's per-draw models 0.74 ms per POINT at N = 8, 0.96 at 128): "auto" = GPU.
        self.gpu_refine = str(gpu_refine)
        # Depth of the slice sampler's speculation (util.slice_sample_batched): `lookahead` = step-out points per side and
        # shrink proposals evaluated per GPU call; `follow` = "P:H": cross-move speculation -- the next coordinate move's
        # edges, ladder and first P proposals under the hypotheses "this move accepts its 1st ... H-th proposal" ride in
        # the same call.  What a row costs decides the depth: "auto" measures it once per problem size
        # (_speculation_depth).  The Markov chain and the RNG stream are the reference's at every depth.
        self.lookahead = lookahead if str(lookahead) == "auto" else max(1, int(lookahead))
        self.follow = follow if str(follow) == "auto" else tuple(int(v) for v in str(follow).split(":"))
        self._depth_cache = {}
        # who runs the slice sampler on the GPU path: "native" = spx_sample_hypers (C++ inside libspx), "python" =
        # util.slice_sample_batched around Engine.gp_logprob (round 5).  Same chain either way.
        self.sampler = str(sampler)
        if self.sampler not in ("native", "python"):
            raise ValueError("sampler must be native or python (got %r)" % (sampler,))
        self._native_hist = None
        self.sampler_stats = {"calls": 0, "rows": 0, "moves": 0, "free_moves": 0, "ns_in_calls": 0, "ns_total": 0, "calls_by_rows": [0] * 34}
        # opt-in: have the driver's ExperimentGrid build its Sobol candidate grid with the HIP
        # generator (bit-identical; spearmint-lite rebuilds the grid on every invocation)
        self.gpu_sobol = _as_bool(gpu_sobol)
        if self.gpu_sobol:
            from .. import sobol as _sobol
            _sobol.install(device=self.device, log=log)
        self._lp_key = None
        self._lp_refs = None
        self._slice_hist = {}     # per kind of move: how the ends of the slice bracket behaved (util._slice_along_batched)
        self.D = -1
        self._eng = None          # created lazily in next(): never before a fork, never pickled
        self.last_overall_ei = None

    # -- GPU handle -----------------------------------------------------------
    def engine(self):
        if self._eng is None:
            from ..engine import Engine, MultiEngine
            if self.ndev > 1:   # one handle over ndev GPUs: sharding + the RCCL all-gather live in libspx
                self._eng = MultiEngine(range(self.device, self.device + self.ndev), self.lib_path)
            else:
                self._eng = Engine(self.device, self.lib_path)
            self._eng.set_covar(self.covar)
        return self._eng

    def __getstate__(self):
        # pickling (the reference ships copy.copy(self) to multiprocessing workers, GPEIOptChooser.py:274-280)
        # drops the handle.  os.fork() does NOT pickle: a forked child inherits `_eng` as it is, and Engine
        # itself refuses calls from a pid other than its creator's (engine.py: Engine._h).
        d = dict(self.__dict__)
        d["_eng"] = None
        d["_slice_hist"] = {}     # batching statistics of THIS process: timing only, never part of a shipped copy
        d["_native_hist"] = None
        return d

    # -- persistent state -------------------------------------------------------
    def _state_dict(self):
        return {"dims": self.D, "ls": self.ls, "amp2": self.amp2, "noise": self.noise,
                "mean": self.mean}

    def _apply_state(self, state):
        self.D = state["dims"]
        self.ls = state["ls"]
        self.amp2 = state["amp2"]
        self.noise = state["noise"]
        self.mean = state["mean"]

    def _fresh_state(self, dims, values):
        """Defaults of GPEIChooser.py:100-113."""
        self.D = dims
        self.ls = np.ones(self.D)
        self.amp2 = np.std(values) + 1e-4
        self.noise = 1e-3
        self.mean = np.mean(values)

    def save_state(self):
        if self.D == -1:
            return
        self.locker.lock_wait(self.state_pkl)
        try:
            pickle_atomically(self._state_dict(), self.state_pkl)
        finally:
            self.locker.unlock(self.state_pkl)

    def _real_init(self, dims, values, *more):
        self.locker.lock_wait(self.state_pkl)
        try:
            if os.path.exists(self.state_pkl):
                self._apply_state(unpickle(self.state_pkl))
                self._loaded_from_disk = True
            else:
                self._fresh_state(dims, values, *more)
                self._loaded_from_disk = False
        finally:
            self.locker.unlock(self.state_pkl)

    # -- log-likelihood data term: host or GPU ------------------------------------------
    def _use_gpu_logprob(self, n):
        if self.gpu_logprob == "auto":   # thresholds measured on one MI355X box; SPX_LOGPROB_MIN_N overrides them
            # (round 6: with the sampler inside the library a call costs 35 us and nothing between calls -- a whole next() at
            # N = 2 ... 32 takes 0.011-0.014 s against 0.019-0.026 s with the reference's serial sampler on the host,
            # profiles/r06_logprob_threshold.log: the GPU from the first GP proposal on.  The Python batched sampler breaks even
            # near N = 10, as measured in round 3: 16.)
            return n >= int(os.environ.get("SPX_LOGPROB_MIN_N", "2" if self.sampler == "native" else "16"))
        return _as_bool(self.gpu_logprob)

    def _use_gpu_refine(self, n):
        if self.gpu_refine == "auto":    # the GPU wins from N = 8 on; SPX_REFINE_MIN_N keeps the first proposals on the host
            return n >= int(os.environ.get("SPX_REFINE_MIN_N", "0"))
        return _as_bool(self.gpu_refine)

    def _speculation_depth(self, comp, vals):
        """(lookahead, (follow proposals, follow hypotheses)) for this problem size.

        A log-likelihood call is latency-bound at Spearmint's operating sizes: on one MI355X a call costs the same for 1
        and for 32 hyper rows up to N = 512 (0.046 ms at N = 64, 0.089 at 256, 0.146 -> 0.18 at 512;
        profiles/r06_lean_rows.log), +40 % for 20 rows at N = 1024, and +8 % PER ROW at N = 2048.  So the depth follows
        the measured marginal cost of a row, r = (t(17 rows) - t(1 row)) / (16 t(1 row)), taken once per (N, D) with the
        chain's current hyper row (no random numbers are drawn; 2 x 3 calls).  Measured with the native sampler
        (profiles/r06_next_phases.log; next() at N = 64 / 256 / 1024 / 2048):
            r <= 0.01 and a call >= 65 us   lookahead 8, follow 6:3   (N = 256: 252 -> 180 calls, next() 31.3 -> 26.5 ms)
            r <= 0.03                       lookahead 8, follow 4:2   (N = 64: 280 -> 207 calls; N = 1024: 445 -> 341,
                                            139 -> 122 ms; below 65 us per call the host's ~0.6 us per extra row decides)
            else                            lookahead 6, no follow    (N = 2048: a row costs 12 % of a call)
        (<= 32 rows per call either way: the one-launch limit of spx_gp_logprob.)
        Explicit `lookahead=` / `follow=` method arguments override either part."""
        key = (comp.shape[0], comp.shape[1])
        got = self._depth_cache.get(key)
        if got is None:
            la, fo = self.lookahead, self.follow
            if la == "auto" or fo == "auto":
                import time
                eng = self.engine()
                self._resident_observations(eng, comp, vals)
                row = self.current_hyper_row()[None, :]
                t = {}
                for n in (1, 17):
                    rows = np.repeat(row, n, axis=0)
                    best = np.inf
                    for _ in range(3):
                        t0 = time.perf_counter()
                        eng.set_hypers(rows)
                        eng.gp_logprob()
                        best = min(best, time.perf_counter() - t0)
                    t[n] = best
                r = (t[17] - t[1]) / (16.0 * t[1])
                auto = (8, (6, 3)) if (r <= 0.01 and t[1] >= 65e-6) else ((8, (4, 2)) if r <= 0.03 else (6, (0, 0)))
                la = auto[0] if la == "auto" else la
                fo = auto[1] if fo == "auto" else fo
                self._depth_info = {"N": key[0], "D": key[1], "t1_ms": 1e3 * t[1], "t17_ms": 1e3 * t[17], "row_cost": r,
                                    "lookahead": la, "follow": fo}
            got = self._depth_cache[key] = (la, tuple(fo))
        return got

    def data_logprob(self, comp, vals, mean, amp2, noise, ls):
        """-sum log diag L - 0.5 r'K^-1 r (GPEIChooser.py:281-285).  The slice sampler's
        control flow and RNG use stay on the host; only this O(N^3) term moves."""
        if not self._use_gpu_logprob(comp.shape[0]):
            return hostgp.data_logprob(comp, vals, mean, amp2, noise, ls, self.covar)
        eng = self.engine()
        self._resident_observations(eng, comp, vals)
        eng.set_hypers(np.concatenate(([mean, noise, amp2], np.asarray(ls, dtype=float)))[None, :])
        return float(eng.gp_logprob(raise_not_pd=True)[0])

    def _resident_observations(self, eng, comp, vals):
        """Upload (comp, vals) unless the engine already holds these very arrays.  The cache keeps
        references to the arrays it was filled from, so CPython cannot hand their ids to new arrays
        while the entry is alive; sample_hypers() and every EI pass drop the entry."""
        key = (id(comp), id(vals), comp.shape)
        if self._lp_key != key:
            eng.set_observations(comp, vals)
            self._lp_key = key
            self._lp_refs = (comp, vals)

    def data_logprob_many(self, comp, vals, rows):
        """The data term for several hyper rows [mean, noise, amp2, ls...] in ONE GPU call
        (the batched factorisation has the latency of a single one).  Returns (values, not_pd)."""
        eng = self.engine()
        self._resident_observations(eng, comp, vals)
        eng.set_hypers(rows)
        lp = eng.gp_logprob()
        return lp, np.isneginf(lp)

    def _speculative_logprob(self, comp, vals, to_row, finish, kind="ls", admissible=None, to_rows=None):
        """Adapter for util.slice_sample_batched: `to_row(x)` gives the hyper row to evaluate or
        None when x is rejected a priori (-inf without touching the GP, as the reference's
        closures do); `finish(x, data_lp)` adds the priors.  `to_rows(X[K, D]) -> list of K rows / None`, when
        given, does the same for a whole batch in a few array operations (a default-depth next() passes 16 000
        points through here, two thirds of them ladder points outside the priors' support)."""
        # Data terms of the last two batches, keyed by the hyper row's bytes: every slice move starts by
        # re-evaluating the point the previous move accepted (util.py:46), which an earlier batch already
        # holds -- the value is a deterministic function of the row (and does not depend on which other rows share
        # the call: tests/test_gpu_a_parity.py, batch-size bit invariance), so it is reused, not recomputed.  The same
        # memo is what makes cross-move speculation work: a batch may carry `extras`, points the NEXT move will ask
        # for if this one ends as guessed; that move then finds them here and needs no call of its own.
        memo = [{}, {}]          # [current batch, the one before]
        max_rows = self.max_rows_per_call

        def rows_of(xs):
            if to_rows is not None and len(xs) > 1:
                return to_rows(np.array(xs, dtype=float))
            return [to_row(x) for x in xs]

        def submit(xs, extras=()):
            values = [-np.inf] * len(xs)
            errors = [None] * len(xs)
            missing = {}

            def settle(k, lpk, badk):
                if badk:   # spla.cholesky would raise here -- only if the sampler really gets to it
                    errors[k] = np.linalg.LinAlgError("covariance not positive definite")
                else:
                    values[k] = finish(xs[k], lpk)

            for k, r in enumerate(rows_of(xs)):
                if r is None:
                    continue                      # -inf a priori, as the reference's closures return it
                key = np.asarray(r, dtype=float).tobytes()
                got = memo[0].get(key) or memo[1].get(key)
                if got is not None:
                    settle(k, got[0], got[1])
                else:
                    missing[k] = (key, r)
            if not missing:
                return util._LazyValues(values, errors)

            def fill(lv):
                rows, index = [], {}
                for k, (key, r) in missing.items():
                    if key not in index:
                        index[key] = len(rows)
                        rows.append(r)
                room = max_rows - len(rows)
                if room > 0 and len(extras):
                    for r in rows_of(list(extras)):
                        if room <= 0:
                            break
                        if r is None:
                            continue
                        key = np.asarray(r, dtype=float).tobytes()
                        if key in index or key in memo[0] or key in memo[1]:
                            continue
                        index[key] = len(rows)
                        rows.append(r)
                        room -= 1
                lp, bad = self.data_logprob_many(comp, vals, np.array(rows))
                memo[1] = memo[0]
                memo[0] = dict((key, (lp[j], bool(bad[j]))) for key, j in index.items())
                for k, (key, r) in missing.items():
                    j = index[key]
                    settle(k, lp[j], bool(bad[j]))
                lv.missing = ()
            return util._LazyValues(values, errors, set(missing), fill)

        def many(xs):            # the eager form: everything the batch names is evaluated now
            lv = submit(xs)
            if lv.missing:
                lv.fill(lv)
            return lv
        many.submit = submit
        # what lets the sampler plan its speculative batches: where the log-probability is -inf whatever the data
        # say, and how the two ends of the bracket behaved in this chooser's earlier moves of the same kind
        many.admissible = admissible if admissible is not None else (lambda x: to_row(x) is not None)
        many.history = self.__dict__.setdefault("_slice_hist", {}).setdefault(kind, {})
        return many

    # -- hyper-parameter sampling (host; GPEIChooser.py:268-346) -----------------
    def _amp2_logprior(self, amp2, scale):
        a = np.sqrt(amp2) if self.amp2_prior_on_sqrt else amp2
        return -0.5 * (np.log(a) / scale) ** 2

    def _draw_mean_amp_noise(self, comp, vals, ls, cur, noise_scale, amp2_scale,
                             noiseless, on_sqrt=None):
        """One joint slice move over [mean, amp2, noise]; returns the new triple."""
        lo, hi = np.min(vals), np.max(vals)
        check_mean = (not noiseless) or self.noiseless_checks_mean
        sqrt_prior = self.amp2_prior_on_sqrt if on_sqrt is None else on_sqrt
        ls = np.asarray(ls, dtype=float)

        def admissible(h):
            mean, amp2 = h[0], h[1]
            noise = 1e-3 if noiseless else h[2]
            if check_mean and (mean > hi or mean < lo):
                return None
            if amp2 < 0 or noise < 0:
                return None
            return mean, amp2, noise

        def priors(h, lp):
            amp2 = h[1]
            noise = 1e-3 if noiseless else h[2]
            if not noiseless:
                lp += np.log(np.log(1 + (noise_scale / noise) ** 2))
            a = np.sqrt(amp2) if sqrt_prior else amp2
            lp -= 0.5 * (np.log(a) / amp2_scale) ** 2
            return lp

        def logprob(h):
            ok = admissible(h)
            if ok is None:
                return -np.inf
            return priors(h, self.data_logprob(comp, vals, ok[0], ok[1], ok[2], ls))

        start = np.array(cur, dtype=float)
        if self._use_gpu_logprob(comp.shape[0]):
            def to_row(h):
                ok = admissible(h)
                return None if ok is None else np.concatenate(([ok[0], ok[2], ok[1]], ls))
            new = util.slice_sample_batched(start, self._speculative_logprob(comp, vals, to_row, priors, kind="joint",
                                                                             admissible=lambda h: admissible(h) is not None),
                                            compwise=False, lookahead=self._speculation_depth(comp, vals)[0])
        else:
            new = util.slice_sample(start, logprob, compwise=False)
        return new[0], new[1], (1e-3 if noiseless else new[2])

    def _draw_ls(self, comp, vals, mean, amp2, noise, ls, max_ls):
        def inside(cand_ls):
            return not ((cand_ls < 0).any() or (cand_ls > max_ls).any())

        def logprob(cand_ls):
            if not inside(cand_ls):
                return -np.inf
            return self.data_logprob(comp, vals, mean, amp2, noise, cand_ls)

        if self._use_gpu_logprob(comp.shape[0]):
            def to_row(cand_ls):
                return np.concatenate(([mean, noise, amp2], cand_ls)) if inside(cand_ls) else None

            def to_rows(X):      # the same for the K points of a batch: one comparison over [K, D], one block of rows
                ok = ~((X < 0) | (X > max_ls)).any(axis=1)
                R = np.empty((X.shape[0], 3 + X.shape[1]))
                R[:, 0] = mean; R[:, 1] = noise; R[:, 2] = amp2
                R[:, 3:] = X
                return [R[k] if ok[k] else None for k in range(X.shape[0])]
            la, fo = self._speculation_depth(comp, vals)
            return util.slice_sample_batched(ls, self._speculative_logprob(comp, vals, to_row, lambda x, lp: lp,
                                                                           admissible=inside, to_rows=to_rows),
                                             compwise=True, lookahead=la, follow=fo)
        return util.slice_sample(ls, logprob, compwise=True)

    def _use_native_sampler(self, n):
        return self.sampler == "native" and self._use_gpu_logprob(n) and hasattr(self.engine(), "sample_hypers")

    def _sample_model(self, comp, vals, state, noise_scale, amp2_scale, noiseless, on_sqrt, max_ls, n_iter):
        """`n_iter` iterations of the reference's sample_hypers for ONE GP (GPEIChooser.py:268-274: `_sample_noisy` /
        `_sample_noiseless`, then `_sample_ls`): state = (mean, amp2, noise, ls) in; the list of states after each
        iteration out, and the exception that ended the loop early (or None).  On the GPU path the whole loop is one call into libspx (spx_sample_hypers: the sampler's control
        flow, priors, speculation and numpy's random stream in C++, one spx_gp_logprob per batch); `sampler=python` keeps
        the round-5 form (util.slice_sample_batched around Engine.gp_logprob), tiny problems / gpu_logprob=0 the
        reference's own serial form on the host.  All three are the same Markov chain."""
        mean, amp2, noise, ls = state
        if n_iter <= 0:
            return [], None
        if self._use_native_sampler(comp.shape[0]):
            from ..engine import SamplerCfg
            eng = self.engine()
            self._resident_observations(eng, comp, vals)
            la, fo = self._speculation_depth(comp, vals)
            D = comp.shape[1]
            cfg = SamplerCfg(D=D, n_iter=int(n_iter), noiseless=int(bool(noiseless)),
                             check_mean=int((not noiseless) or self.noiseless_checks_mean),
                             amp2_prior_on_sqrt=int(self.amp2_prior_on_sqrt if on_sqrt is None else on_sqrt),
                             lookahead=int(la), follow_props=int(fo[0]), follow_hyps=int(fo[1]),
                             max_rows=int(self.max_rows_per_call), noise_scale=float(noise_scale),
                             amp2_scale=float(amp2_scale), max_ls=float(max_ls),
                             vals_min=float(np.min(vals)), vals_max=float(np.max(vals)))
            hyper = np.concatenate(([mean, noise, amp2], np.asarray(ls, dtype=float)))
            if self._native_hist is None:
                self._native_hist = np.zeros(12)
            try:
                rows, st = eng.sample_hypers(cfg, hyper, self._native_hist)
                err = None
            except Exception as ex:
                st, rows = getattr(ex, "stats", None), getattr(ex, "rows_done", None)
                if st is None:
                    raise
                err = ex                    # (the chain stays where the reference's exception leaves it)
                err.state_at_error = (hyper[0], hyper[2], hyper[1], hyper[3:].copy())
            for k in ("calls", "rows", "moves", "free_moves", "ns_in_calls", "ns_total"):
                self.sampler_stats[k] += st[k]
            self.sampler_stats["calls_by_rows"] = [a + b for a, b in zip(self.sampler_stats["calls_by_rows"], st["calls_by_rows"])]
            return [(r[0], r[2], r[1], r[3:].copy()) for r in rows], err
        out = []
        for _ in range(n_iter):
            if noiseless:
                noise = 1e-3
            try:
                mean, amp2, noise = self._draw_mean_amp_noise(comp, vals, ls, [mean, amp2, noise], noise_scale, amp2_scale,
                                                              noiseless, on_sqrt=on_sqrt)
                ls = self._draw_ls(comp, vals, mean, amp2, noise, ls, max_ls)
            except Exception as ex:
                ex.state_at_error = (mean, amp2, noise, ls)
                return out, ex
            out.append((mean, amp2, noise, ls))
        return out, None

    def sample_hypers(self, comp, vals):
        self._lp_key = None       # whatever the engine holds from an earlier call is not trusted
        self.sample_hypers_many(comp, vals, 1)

    def sample_hypers_many(self, comp, vals, n_iter, after_each=None):
        """n_iter iterations of sample_hypers; after_each(i) runs after every one with self.mean / amp2 / noise / ls set
        to that iteration's sample (logging, collecting hyper_samples).  One library call on the native path."""
        if self.noiseless:
            self.noise = 1e-3
        states, err = self._sample_model(comp, vals, (self.mean, self.amp2, self.noise, self.ls), self.noise_scale,
                                         self.amp2_scale, self.noiseless, None, self.max_ls, n_iter)
        for i, (mean, amp2, noise, ls) in enumerate(states):
            self.mean, self.amp2, self.noise, self.ls = mean, amp2, noise, ls
            if after_each is not None:
                after_each(i)
        if err is not None:       # a finished joint move stays applied, an unfinished sweep does not -- as in the reference
            self.mean, self.amp2, self.noise, self.ls = err.state_at_error
            raise err
        self._log_engine_warning()

    def _log_engine_warning(self):
        """The log-likelihood calls are the main users of the one-launch factorisation: its fallback warning (a hand-off
        time-out; results unaffected) is polled after the sampler too, not only after an EI pass."""
        if self._eng is not None and hasattr(self._eng, "last_warning"):
            warning = self._eng.last_warning()
            if warning:
                log("libspx " + warning)

    def current_hyper_row(self):
        return np.concatenate(([self.mean, self.noise, self.amp2], np.asarray(self.ls, dtype=float)))

    def _log_hypers(self, prefix=""):
        log("%smean: %f  amp: %f  noise: %f  min_ls: %f  max_ls: %f"
            % (prefix, self.mean, np.sqrt(self.amp2), self.noise, np.min(self.ls), np.max(self.ls)))

    # -- the hot path: one call into libspx ---------------------------------------
    def ei_over_hypers_gpu(self, comp, pend, cand, vals, hyper_rows, want_draws=True, randn=None):
        """overall_ei[M, H] and the index of argmax(mean) -- the replacement of
        the reference's compute_ei loop (GPEIChooser.py:143-153,
        GPEIOptChooser.py:331-341).  Raises numpy.linalg.LinAlgError when a
        covariance is not positive definite, exactly where spla.cholesky would."""
        if pend.shape[0] > 0:
            return self._ei_with_pending_gpu(comp, pend, cand, vals, hyper_rows, randn, want_draws)
        hyper_rows = np.ascontiguousarray(np.atleast_2d(hyper_rows), dtype=np.float64)
        self._lp_key = None     # the one-shot call below replaces the engine's resident observations
        idx, val, mean, draws = self.engine().ei_grid(comp, vals, cand, hyper_rows,
                                                      want_mean=True, want_draws=want_draws)
        warning = self.engine().last_warning()    # (a hand-off time-out of the one-launch factorisation: results are unaffected)
        if warning:
            log("libspx " + warning)
        self.last_overall_ei = draws
        self.last_ei_mean = mean
        return idx, mean, draws

    def _ei_with_pending_gpu(self, comp, pend, cand, vals, hyper_rows, randn, want_draws):
        """Pending experiments: fantasise their outcomes (GPEIChooser.py:209-266).
        GPU: factorisation of cov([comp; pend]) for every draw, K(X*,X), the solve,
        Sigma beta^2 and the S fantasy means, EI averaged over fantasies.  Host: the
        O(N^2 P) posterior of the P pending points and the S joint fantasy draws
        (`randn[h]` is the (P, S) standard-normal matrix of draw h, drawn by the
        caller at the point where the reference consumes the RNG)."""
        hyper_rows = np.ascontiguousarray(np.atleast_2d(hyper_rows), dtype=np.float64)
        H, n_comp, n_pend = hyper_rows.shape[0], comp.shape[0], pend.shape[0]
        eng = self.engine()
        self._lp_key = None
        comp_pend = np.concatenate((comp, pend))
        eng.set_observations(comp_pend, np.concatenate((vals, np.zeros(n_pend))))
        eng.set_candidates(cand)
        eng.set_hypers(hyper_rows)
        eng.factor()
        S = randn[0].shape[1]
        fant = np.empty((H, n_comp + n_pend, S))
        bests = np.empty((H, S))
        for h in range(H):
            # the bottom P rows of the factor and gamma are all the posterior of the pending points needs (hostgp:
            # fantasize_from_factor_rows) -- not the N x N sub-Cholesky and two O(N^2 P) host solves against it
            l_rows, gam = eng.get_factor_rows(h, n_comp, n_pend)
            fant[h], bests[h] = hostgp.fantasize_from_factor_rows(vals, hyper_rows[h], l_rows, gam, randn[h])
        eng.set_fantasies(fant, bests)
        eng.ei_run()
        idx, _ = eng.best()
        mean = eng.ei_mean()
        draws = eng.ei_draws() if want_draws else None
        self.last_overall_ei = draws
        self.last_ei_mean = mean
        return idx, mean, draws

    @staticmethod
    def _split(grid, values, candidates, pending, complete):
        grid = np.asarray(grid, dtype=np.float64)
        values = np.asarray(values, dtype=np.float64)
        comp = grid[complete, :]
        cand = grid[candidates, :]
        pend = grid[pending, :]
        vals = values[complete]
        return comp, cand, pend, vals

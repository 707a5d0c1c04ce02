"""Expected improvement per second: a second GP on log-durations, EI divided by
the predicted duration -- the MI355X drop-in for
spearmint/spearmint/chooser/GPEIperSecChooser.py.

Reference defects on this path and what this module does about them
(SURVEY.md section 8(a) row A10):
  (i)  ei_over_hypers returns from inside its loop (:302), so only draw 0 is
       ever evaluated;
  (ii) time_hyper_samples is never cleared (:199 clears only hyper_samples), so
       ei_over_hypers pairs draw i with a burn-in time sample.
The default here is the intended semantics (all draws, paired samples);
``ref_compat=1`` reproduces (i) and (ii) for side-by-side comparisons."""
from __future__ import absolute_import, print_function

import os

import numpy as np
import numpy.random as npr
import scipy.optimize as spo

from .. import hostgp
from .. import refine
from .. import util
from ..helpers import log
from ._base import GPEIBase, _as_bool


def init(expt_dir, arg_string):
    args = util.unpack_args(arg_string)
    return GPEIperSecChooser(expt_dir, **args)


class GPEIperSecChooser(GPEIBase):
    amp2_prior_on_sqrt = False      # objective GP: log(amp2) (:626, :672)
    noiseless_checks_mean = True    # :660-661
    max_ls = 10                     # :74
    time_noise_scale = 0.1
    time_amp2_scale = 1
    time_max_ls = 10

    def __init__(self, expt_dir, covar="Matern52", mcmc_iters=10, pending_samples=100,
                 noiseless=False, burnin=100, grid_subset=20, ref_compat=False, **kw):
        GPEIBase.__init__(self, expt_dir, covar=covar, mcmc_iters=mcmc_iters,
                          pending_samples=pending_samples, noiseless=noiseless, **kw)
        self.stats_file = os.path.join(expt_dir, self.__module__ + "_hyperparameters.txt")
        self.burnin = int(burnin)
        self.needs_burnin = True
        self.grid_subset = int(grid_subset)
        self.ref_compat = _as_bool(ref_compat)
        self.hyper_samples = []
        self.time_hyper_samples = []

    # -- state (:83-143) -----------------------------------------------------------
    def _state_dict(self):
        d = GPEIBase._state_dict(self)
        d.update({"time_ls": self.time_ls, "time_amp2": self.time_amp2,
                  "time_noise": self.time_noise, "time_mean": self.time_mean})
        return d

    def _apply_state(self, state):
        GPEIBase._apply_state(self, state)
        self.time_ls = state["time_ls"]
        self.time_amp2 = state["time_amp2"]
        self.time_noise = state["time_noise"]
        self.time_mean = state["time_mean"]

    def _fresh_state(self, dims, values, durations):
        GPEIBase._fresh_state(self, dims, values)
        self.time_ls = np.ones(self.D)
        self.time_amp2 = np.std(durations) + 1e-4
        self.time_noise = 1e-3
        self.time_mean = np.mean(np.log(durations))

    def dump_hypers(self):
        self.save_state()

    # -- sampling (:550-563) ---------------------------------------------------------
    def sample_hypers(self, comp, vals, durs):
        GPEIBase.sample_hypers(self, comp, vals)
        states, err = self._sample_model(comp, durs, (self.time_mean, self.time_amp2, self.time_noise, self.time_ls),
                                         self.time_noise_scale, self.time_amp2_scale, False, True, self.time_max_ls, 1)
        if err is not None:
            self.time_mean, self.time_amp2, self.time_noise, self.time_ls = err.state_at_error
            raise err
        self.time_mean, self.time_amp2, self.time_noise, self.time_ls = states[0]
        self.hyper_samples.append((self.mean, self.noise, self.amp2, self.ls))
        self.time_hyper_samples.append((self.time_mean, self.time_noise, self.time_amp2, self.time_ls))

    @staticmethod
    def _rows(samples):
        return np.array([np.concatenate(([h[0], h[1], h[2]], np.asarray(h[3], dtype=float)))
                         for h in samples])

    def _paired_samples(self):
        """(objective rows, time rows) that ei_over_hypers evaluates."""
        H = self.mcmc_iters
        if self.ref_compat:
            # (i) only draw 0, (ii) paired with time_hyper_samples[0] of the never-cleared list
            return self._rows(self.hyper_samples[:1]), self._rows(self.time_hyper_samples[:1])
        return self._rows(self.hyper_samples[:H]), self._rows(self.time_hyper_samples[-H:])

    # -- the hot path ----------------------------------------------------------------
    def ei_per_s_over_hypers_gpu(self, comp, pend, cand, vals, durs):
        rows, trows = self._paired_samples()
        if self.ref_compat:
            # a consequence of defect (i): ei_over_hypers leaves the chooser's CURRENT hypers at the one
            # pair it evaluated (:288-296), so that is what dump_hypers pickles and where the next call's
            # chain starts
            self.mean, self.noise, self.amp2, self.ls = self.hyper_samples[0]
            self.time_mean, self.time_noise, self.time_amp2, self.time_ls = self.time_hyper_samples[0]
        self._lp_key = None
        self._resident_plain = pend.shape[0] == 0   # the engine then holds exactly (comp, rows, trows)
        if pend.shape[0] > 0:
            return self._ei_per_s_with_pending(comp, pend, cand, vals, durs, rows, trows)
        idx, val, mean, draws = self.engine().ei_per_sec_grid(comp, vals, durs, cand, rows, trows,
                                                              want_mean=True, want_draws=False)
        if self.ref_compat:
            # the other mcmc_iters-1 columns of overall_ei stay zero in the reference
            mean = mean / float(self.mcmc_iters)
        return idx, mean

    def _ei_per_s_with_pending(self, comp, pend, cand, vals, durs, rows, trows):
        """Pending branch (:492-548): EI averaged over fantasies (objective GP over
        [comp; pend], GPU) divided by the predicted duration (time GP over comp only,
        GPU).  The two GPs have different observation sets, so they are two engine
        passes; the final M x H division, mean and argmax are a tiny host step."""
        from ..engine import FLAG_KEEP_MOMENTS, FLAG_PER_SEC, FLAG_TIME_ONLY
        eng = self.engine()
        H = rows.shape[0]
        # pass 1: durations.  The fantasy normals are drawn first, where the reference
        # consumes the RNG (once per evaluated draw, :523).
        randn = [npr.randn(pend.shape[0], int(self.pending_samples)) for _ in range(H)]
        eng.set_observations(comp, vals)
        eng.set_candidates(cand)
        eng.set_hypers(rows)
        eng.set_time_model(durs, trows)
        eng.ei_step(FLAG_PER_SEC | FLAG_KEEP_MOMENTS | FLAG_TIME_ONLY)   # factor + the predicted durations, nothing else
        time_m = np.stack([eng.get_time_mean(h) for h in range(H)], axis=1)
        # pass 2: EI averaged over fantasies
        _, _, ei = self._ei_with_pending_gpu(comp, pend, cand, vals, rows, randn, True)
        overall = np.zeros((cand.shape[0], self.mcmc_iters))
        overall[:, :H] = ei / time_m
        mean = np.mean(overall, axis=1)
        return int(np.argmax(mean)), mean

    def _refine(self, points, comp, vals, durs):
        """L-BFGS-B on the summed EI per second (:223-230; the reference ignores pending jobs
        here).  On the GPU the objective uses the dual factorisation the first pass left
        resident (spx_ei_grad_batch), or sets it up when that is not the one summed over."""
        bounds = [(0, 1)] * comp.shape[1]
        if self.covar == "SE":   # getattr(gp, 'grad_SE') at :383
            raise AttributeError("gp has no attribute 'grad_SE': the reference's refinement cannot run with covar=SE")
        rows, trows = (self.hyper_samples[:self.mcmc_iters],
                       (self.time_hyper_samples[:self.mcmc_iters] if self.ref_compat
                        else self.time_hyper_samples[-self.mcmc_iters:]))
        if self._use_gpu_refine(comp.shape[0]):
            eng = self.engine()
            if self.ref_compat or not self._resident_plain:
                # the engine does not hold the dual factorisation this objective sums over -- bug-compatible mode pairs
                # the draws differently from the EI pass (:322-434 iterate the never-cleared list from its start), and with
                # pending jobs the EI pass left [comp; pend] resident -- so it is set up here: one factorisation, then
                # every refinement point of every L-BFGS-B instance goes through spx_ei_grad_batch as usual
                eng.set_observations(comp, vals)
                eng.set_hypers(self._rows(rows))
                eng.set_time_model(durs, self._rows(trows))
                eng.factor()
                self._lp_key = None
                self._resident_plain = False
            return refine.lbfgs_many(eng.ei_grad_batch, points, bounds, log=log)
        models = [hostgp.PerSecPointModel(comp, vals, durs, h, t, self.covar) for h, t in zip(rows, trows)]

        def objective(x):
            total, grad = 0.0, np.zeros(x.shape[0])
            for m in models:
                e, g = m.neg_ei_and_grad(x)
                total += e
                grad = grad + g
            return total, grad

        out = np.array(points, dtype=float, copy=True)
        for i in range(out.shape[0]):
            log("Optimizing candidate %d/%d" % (i + 1, out.shape[0]))
            out[i, :] = spo.fmin_l_bfgs_b(objective, out[i, :].flatten(), bounds=bounds, disp=0)[0]
        return out

    # -- plugin entry (:155-281) -------------------------------------------------------
    def next(self, grid, values, durations, candidates, pending, complete):
        if complete.shape[0] < 2:
            return int(candidates[0])
        durations = np.asarray(durations, dtype=np.float64)
        if self.D == -1:
            self._real_init(np.asarray(grid).shape[1], np.asarray(values)[complete],
                            durations[complete])
        comp, cand, pend, vals = self._split(grid, values, candidates, pending, complete)
        durs = np.log(durations[complete]).squeeze()   # log domain keeps times positive (:174-176)
        numcand = cand.shape[0]
        best_comp = np.argmin(vals)
        cand2 = np.vstack((np.random.randn(10, comp.shape[1]) * 0.001 + comp[best_comp, :], cand))

        if self.mcmc_iters <= 0:
            raise NotImplementedError("mcmc_iters=0: the reference's own branch (GPEIperSecChooser.py:243-281) cannot run -- it reads "
                                      "overall_ei before assigning it (UnboundLocalError); use mcmc_iters >= 1")

        if self.needs_burnin:
            for it in range(self.burnin):
                self.sample_hypers(comp, vals, durs)
                self._log_hypers("BURN %d/%d] " % (it + 1, self.burnin))
            self.needs_burnin = False

        self.hyper_samples = []
        if not self.ref_compat:
            self.time_hyper_samples = []
        for it in range(self.mcmc_iters):
            self.sample_hypers(comp, vals, durs)
            self._log_hypers("%d/%d] " % (it + 1, self.mcmc_iters))
            log("%d/%d] time_mean: %.2fs time_amp: %.2f  time_noise: %.4f time_min_ls: %.4f  time_max_ls: %.4f"
                % (it + 1, self.mcmc_iters, np.exp(self.time_mean), np.sqrt(self.time_amp2),
                   np.exp(self.time_noise), np.min(self.time_ls), np.max(self.time_ls)))
        self.dump_hypers()

        _, mean1 = self.ei_per_s_over_hypers_gpu(comp, pend, cand2, vals, durs)
        keep = np.argsort(mean1)[-self.grid_subset:]
        refined = self._refine(cand2[keep, :], comp, vals, durs)

        # second pass (:233-236).  Without pending jobs only the refined points are new: the grid rows keep
        # their first-pass values (a candidate's result does not depend on the others of the call).  With
        # pending jobs the reference draws FRESH fantasy normals in every pass (:523), so the grid is scored
        # again, as it does.
        if pend.shape[0] > 0:
            cand_all = np.vstack((cand, refined))
            best, _ = self.ei_per_s_over_hypers_gpu(comp, pend, cand_all, vals, durs)
        else:
            _, mean_ref = self.ei_per_s_over_hypers_gpu(comp, pend, refined, vals, durs)
            best = int(np.argmax(np.concatenate((mean1[10:], mean_ref))))
        self.dump_hypers()
        if best >= numcand:
            return (int(numcand), refined[best - numcand, :])
        return int(candidates[best])

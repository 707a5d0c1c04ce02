"""Default Spearmint chooser: GP-EI over the grid, MCMC over hypers, the best
few candidates refined by L-BFGS-B on the summed EI -- the MI355X drop-in for
spearmint/spearmint/chooser/GPEIOptChooser.py.

Both full passes over the candidate grid (GPEIOptChooser.py:269 and :293) run
on the GPU, and so does the objective of the `grid_subset` (=20) L-BFGS-B
refinements: all points that wait for an
evaluation share one spx_ei_grad_batch call (spearmint_amd/refine.py)."""
from __future__ import absolute_import, print_function

import os

import numpy as np
import numpy.random as npr
import scipy.optimize as spo

from .. import hostgp
from .. import refine
from .. import util
from ..helpers import log, unpickle
from ._base import GPEIBase, _as_bool


def init(expt_dir, arg_string):
    args = util.unpack_args(arg_string)
    return GPEIOptChooser(expt_dir, **args)


class GPEIOptChooser(GPEIBase):
    amp2_prior_on_sqrt = True       # :668
    noiseless_checks_mean = True    # :685-686
    max_ls = 2

    def __init__(self, expt_dir, covar="Matern52", mcmc_iters=10, pending_samples=100,
                 noiseless=False, burnin=100, grid_subset=20, use_multiprocessing=True, rescore_grid=0, **kw):
        GPEIBase.__init__(self, expt_dir, covar=covar, mcmc_iters=mcmc_iters,
                          pending_samples=pending_samples, noiseless=noiseless, **kw)
        self.stats_file = os.path.join(expt_dir, self.__module__ + "_hyperparameters.txt")
        self.burnin = int(burnin)
        self.needs_burnin = True
        self.grid_subset = int(grid_subset)
        # accepted for command-line compatibility; the refinement below is serial
        # (a fork-based Pool cannot share a HIP context)
        self.use_multiprocessing = _as_bool(use_multiprocessing)
        self.rescore_grid = _as_bool(rescore_grid)   # 1: score [grid; refined] again in the second pass, as the reference does
        self.hyper_samples = []

    # -- state -----------------------------------------------------------------
    def _state_dict(self):
        d = GPEIBase._state_dict(self)
        d["hyper_samples"] = self.hyper_samples
        return d

    def _apply_state(self, state):
        GPEIBase._apply_state(self, state)
        self.hyper_samples = state["hyper_samples"]
        self.needs_burnin = False

    def _fresh_state(self, dims, values):
        GPEIBase._fresh_state(self, dims, values)
        self.hyper_samples.append((self.mean, self.noise, self.amp2, self.ls))

    def _real_init(self, dims, values):
        self.randomstate = npr.get_state()   # replayed by the pending branch (:171, :588)
        GPEIBase._real_init(self, dims, values)

    def dump_hypers(self):
        """State pickle + human-readable table (:84-120)."""
        self.save_state()
        with open(self.stats_file, "w") as fh:
            fh.write("Mean Noise Amplitude <length scales>\n")
            fh.write("-----------ALL SAMPLES-------------\n")
            rows = [np.hstack(h) for h in self.hyper_samples]
            avg = 0 * rows[0]
            for r in rows:
                avg = avg + (1 / float(len(rows))) * r
                fh.write(" ".join(str(v) for v in r) + " \n")
            fh.write("-----------MEAN OF SAMPLES-------------\n")
            fh.write(" ".join(str(v) for v in avg) + " \n")

    def generate_stats_html(self):
        """HTML/JS snippet for the status page (web/app.py:75-80; :125-148)."""
        if not os.path.exists(self.state_pkl):
            return "Chooser not yet ready to display output"
        self._apply_state(unpickle(self.state_pkl))
        try:
            mean_mean = np.mean(np.vstack([h[0] for h in self.hyper_samples]))
            mean_noise = np.mean(np.vstack([h[1] for h in self.hyper_samples]))
            mean_ls = np.mean(np.vstack([h[3][np.newaxis, :] for h in self.hyper_samples]), 0)
            out = ('<br /><span class="label label-info">Estimated mean:</span> ' + str(mean_mean) +
                   '<br /><span class="label label-info">Estimated noise:</span> ' + str(mean_noise) +
                   '<br /><br /><span class="label label-info">Inverse parameter sensitivity'
                   ' - Gaussian Process length scales</span><br /><br />'
                   '<div id="lschart"></div><script type="text/javascript">'
                   'var lsdata = [' + ",".join(["%.2f" % v for v in mean_ls]) + '];')
        except Exception:
            return "Chooser not yet ready to display output."
        return out + 'bar_chart("#lschart", lsdata, ' + str(self.max_ls) + ');</script>'

    # -- sampling ----------------------------------------------------------------
    def sample_hypers(self, comp, vals):
        GPEIBase.sample_hypers(self, comp, vals)
        self.hyper_samples.append((self.mean, self.noise, self.amp2, self.ls))

    def _sample_and_collect(self, comp, vals, n_iter, prefix):
        """n_iter x sample_hypers (one library call on the native path), each sample logged and appended like :621-628."""
        def after(i):
            self.hyper_samples.append((self.mean, self.noise, self.amp2, self.ls))
            self._log_hypers(prefix % (i + 1, n_iter))
        self._lp_key = None
        self.sample_hypers_many(comp, vals, n_iter, after)

    def hyper_rows(self):
        return np.array([np.concatenate(([h[0], h[1], h[2]], np.asarray(h[3], dtype=float)))
                         for h in self.hyper_samples])

    # -- local refinement on the host (:360-388 summed over draws) ---------------
    def _fantasy_normals(self, pend):
        """The reference restores the RNG state captured in _real_init before every
        fantasy draw (:588), so all draws share one (P, S) matrix -- and the global
        stream is left right after it."""
        out = []
        for _ in self.hyper_samples:
            npr.set_state(self.randomstate)
            out.append(npr.randn(pend.shape[0], self.pending_samples))
        return out

    def _refine(self, points, comp, vals, pend):
        """L-BFGS-B on the summed EI of each kept point (:285-289).  On the GPU all kept points are
        optimised together: every instance's next objective evaluation goes into one
        spx_ei_grad_batch call against the factorisation (and, with pending jobs, the fantasies)
        the first EI pass left resident.  Tiny problems use host models, one point at a time."""
        bounds = [(0, 1)] * comp.shape[1]
        if self.covar == "SE":   # getattr(gp, 'grad_SE') at :404 / :486
            raise AttributeError("gp has no attribute 'grad_SE': the reference's refinement cannot run with covar=SE")
        if self._use_gpu_refine(comp.shape[0]):
            return refine.lbfgs_many(self.engine().ei_grad_batch, points, bounds, log=log)
        if pend.shape[0] > 0:
            models = []
            for h in self.hyper_samples:
                npr.set_state(self.randomstate)
                models.append(hostgp.PendingPointModel(comp, pend, vals, h,
                                                       npr.randn(pend.shape[0], self.pending_samples), self.covar))
        else:
            models = [hostgp.PointModel(comp, vals, h, self.covar) for h in self.hyper_samples]

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

    # -- plugin entry ---------------------------------------------------------------
    def next(self, grid, values, durations, candidates, pending, complete):
        if complete.shape[0] < 2:
            return int(candidates[0])
        if self.D == -1:
            self._real_init(np.asarray(grid).shape[1], np.asarray(values)[complete])
        comp, cand, pend, vals = self._split(grid, values, candidates, pending, complete)
        numcand = cand.shape[0]

        # ten jittered copies of the incumbent in front of the grid (:234-238)
        best_comp = np.argmin(vals)
        cand2 = np.vstack((np.random.randn(10, comp.shape[1]) * 0.001 + comp[best_comp, :], cand))

        if self.mcmc_iters <= 0:
            raise NotImplementedError("mcmc_iters=0: the reference's own branch (GPEIOptChooser.py:300-321) cannot run -- it passes "
                                      "(comp, vals, True) where grad_optimize_ei expects (comp, pend, vals) and dies with a "
                                      "ValueError; use GPEIChooser for ML-II hypers or mcmc_iters >= 1")

        if self.needs_burnin:
            self._sample_and_collect(comp, vals, self.burnin, "BURN %d/%d] ")
            self.needs_burnin = False

        self.hyper_samples = []
        self._sample_and_collect(comp, vals, self.mcmc_iters, "%d/%d] ")
        self.dump_hypers()
        rows = self.hyper_rows()

        # pass 1 over grid + sprayed points, keep the grid_subset best (:269-271)
        randn = self._fantasy_normals(pend) if pend.shape[0] > 0 else None
        _, mean1, _ = self.ei_over_hypers_gpu(comp, pend, cand2, vals, rows, want_draws=False, randn=randn)
        keep = np.argsort(mean1)[-self.grid_subset:]
        refined = self._refine(cand2[keep, :], comp, vals, pend)

        # pass 2 over grid + refined points (:292-299).  The reference scores the whole grid again; a
        # candidate's EI does not depend on which other candidates share the call (bit for bit: the
        # sharding tests rely on it), so the grid rows keep their pass-1 values and only the refined
        # points are scored.  rescore_grid=1 runs the literal second pass instead.
        randn = self._fantasy_normals(pend) if pend.shape[0] > 0 else None
        if self.rescore_grid:
            cand_all = np.vstack((cand, refined))
            best, _, _ = self.ei_over_hypers_gpu(comp, pend, cand_all, vals, rows, randn=randn)
        else:
            _, mean_ref, _ = self.ei_over_hypers_gpu(comp, pend, refined, vals, rows, want_draws=False, randn=randn)
            best = int(np.argmax(np.concatenate((mean1[10:], mean_ref))))
        if best >= numcand:
            return (int(numcand), refined[best - numcand, :])
        return int(candidates[best])

"""GP expected-improvement chooser on the candidate grid, MCMC over hypers --
the MI355X drop-in for spearmint/spearmint/chooser/GPEIChooser.py.

Same plugin API (init / next), same state pickle, same sampler; the EI grid is
evaluated by libspx.so on the GPU (see _base.py)."""
from __future__ import absolute_import, print_function

import numpy as np
import numpy.random as npr

from .. import hostgp
from .. import util
from ._base import GPEIBase


def init(expt_dir, arg_string):
    args = util.unpack_args(arg_string)
    return GPEIChooser(expt_dir, **args)


class GPEIChooser(GPEIBase):
    # GPEIChooser.py:312 puts the log-normal prior on amp2 itself and its
    # noiseless sampler has no bounds check on the mean (:323-340).
    amp2_prior_on_sqrt = False
    noiseless_checks_mean = False
    max_ls = 2

    def __del__(self):
        # the reference persists its hypers when the object dies (:66-83)
        try:
            self.save_state()
        except Exception:
            pass

    def next(self, grid, values, durations, candidates, pending, complete):
        # Too little data for a GP: take the first candidate (:127-128). No GPU touched.
        if complete.shape[0] < 2:
            return int(candidates[0])
        if self.D == -1:
            self._real_init(np.asarray(grid).shape[1], np.asarray(values)[complete])
        comp, cand, pend, vals = self._split(grid, values, candidates, pending, complete)

        if self.mcmc_iters <= 0:
            # ML-II point estimate instead of MCMC (:157-175): optimise the hypers on the host, fall
            # back to the defaults if that fails, then score the grid once -- the same GPU call with
            # a single hyper row (argmax of a one-column mean == argmax(ei))
            try:
                self.mean, self.amp2, self.noise, self.ls = hostgp.optimize_hypers(comp, vals, self.covar)
            except Exception:
                self.ls = np.ones(self.D)
                self.amp2 = np.std(vals)
                self.noise = 1e-3
            self._log_hypers()
            randn = [npr.randn(pend.shape[0], self.pending_samples)] if pend.shape[0] > 0 else []
            best, _, _ = self.ei_over_hypers_gpu(comp, pend, cand, vals, self.current_hyper_row()[None, :],
                                                 randn=randn)
            return int(candidates[best])
        # The reference alternates "sample hypers" and "compute_ei" (:145-151).
        # Without pending experiments compute_ei consumes no random numbers, so
        # drawing all H samples first and scoring them in one GPU call is the
        # same computation; with pending ones its fantasy normals are drawn here,
        # right after the sample they belong to.
        rows, randn = [], []
        if pend.shape[0] > 0:
            for _ in range(self.mcmc_iters):
                self.sample_hypers(comp, vals)
                self._log_hypers()
                rows.append(self.current_hyper_row())
                # compute_ei's only use of the RNG (:238), at the same point of the stream
                randn.append(npr.randn(pend.shape[0], self.pending_samples))
        else:       # nothing else draws random numbers between two samples: all of them in one sampler call
            def after(i):
                self._log_hypers()
                rows.append(self.current_hyper_row())
            self._lp_key = None
            self.sample_hypers_many(comp, vals, self.mcmc_iters, after)
        best, _, _ = self.ei_over_hypers_gpu(comp, pend, cand, vals, np.array(rows), randn=randn)
        return int(candidates[best])

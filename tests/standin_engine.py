"""TEST INFRASTRUCTURE -- a stand-in for spearmint_amd.engine.Engine backed by the CPU oracle.

bench.py's launch / rank / collective plumbing (self-launch of N ranks, the one all-gather of 16-byte records,
ranks_seen, the strong-scaling sub-records) has to be testable where no GPU exists.  The product never imports this
file: bench.py reaches it only through the SPX_BENCH_ENGINE test hook, and a JSON line produced with it carries
"engine": "tests.standin_engine:Engine" -- it is not a measurement of anything.
"""
import numpy as np

from oracle import gp_ei_oracle as orc

FLAG_PER_SEC = 1


class Engine(object):
    def __init__(self, device=0, lib=None, devices=None):
        self.device = int(device)
        self.comp = self.vals = self.cand = self.hypers = None
        self.log_durs = self.time_hypers = None
        self.index_base = 0
        self._best = None
        self._draws = None
        self.N = self.M = self.D = self.H = 0

    def set_observations(self, comp, vals):
        self.comp, self.vals = np.array(comp, dtype=float), np.array(vals, dtype=float).ravel()
        self.N, self.D = self.comp.shape

    def set_candidates(self, cand, index_base=0):
        self.cand, self.index_base = np.array(cand, dtype=float), int(index_base)
        self.M = self.cand.shape[0]

    def set_hypers(self, hypers):
        self.hypers = np.atleast_2d(np.array(hypers, dtype=float))
        self.H = self.hypers.shape[0]

    def set_time_model(self, log_durs, time_hypers):
        self.log_durs = None if log_durs is None else np.array(log_durs, dtype=float).ravel()
        self.time_hypers = None if time_hypers is None else np.atleast_2d(np.array(time_hypers, dtype=float))

    def set_option(self, name, value):
        pass

    def ei_step(self, flags=0):
        if flags & FLAG_PER_SEC:
            ei = orc.ei_per_s_over_hypers(self.comp, self.cand, self.vals, self.log_durs, self.hypers, self.time_hypers)
        else:
            ei = orc.ei_over_hypers(self.comp, self.cand, self.vals, self.hypers)
        self._draws = ei
        mean = np.mean(ei, axis=1)
        i = int(np.argmax(mean))
        self._best = (i + self.index_base, float(mean[i]))

    def best(self):
        return self._best

    def ei_draws(self):
        return self._draws

    def stat(self, name):
        return {"ranks_seen": 1}.get(name, 0)

    def timings(self):
        return {"predict_gemm": (0.0, 0), "cov_cross": (0.0, 0), "ei_finalize": (0.0, 0)}

    def ei_grid(self, comp, vals, cand, hypers, want_mean=True, want_draws=False, flags=0):
        self.set_observations(comp, vals); self.set_candidates(cand); self.set_hypers(hypers)
        self.ei_step(0)
        return self._best[0], self._best[1], None, None

    def ei_per_sec_grid(self, comp, vals, log_durs, cand, hypers, time_hypers, want_mean=True, want_draws=False, flags=0):
        self.set_observations(comp, vals); self.set_candidates(cand); self.set_hypers(hypers)
        self.set_time_model(log_durs, time_hypers)
        self.ei_step(FLAG_PER_SEC)
        return self._best[0], self._best[1], None, None

    def comm_unique_id(self):
        raise RuntimeError("the stand-in engine has no RCCL communicator")

    def comm_attach(self, uid, nranks, rank):
        raise RuntimeError("the stand-in engine has no RCCL communicator")

    def set_partition(self, *a):
        raise RuntimeError("the stand-in engine has no RCCL communicator")

    def close(self):
        pass

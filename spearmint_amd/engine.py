"""ctypes binding of libspx.so (include/spx.h) -- the only way the Python side
reaches the GPU.  There is NO CPU fallback: if the shared library or a HIP
device is missing, construction / the first call raises.

Written in the Python 2/3 common subset so the chooser modules can be dropped
into a Python-2 Spearmint checkout.
"""
from __future__ import print_function

import ctypes
import os

import numpy as np

try:
    from numpy.linalg import LinAlgError
except ImportError:  # pragma: no cover
    LinAlgError = ArithmeticError

SPX_OK = 0
SPX_ERR_ARG = -1
SPX_ERR_HIP = -2
SPX_ERR_NOT_PD = -3
SPX_ERR_SLICE_NAN = -4
SPX_ERR_SLICE_ZERO = -5

COVAR = {"Matern52": 0, "Matern32": 1, "ARDSE": 2, "SE": 3}   # include/spx.h SPX_COVAR_*

FLAG_PER_SEC = 1
FLAG_KEEP_MOMENTS = 2
FLAG_TIMING = 4
FLAG_TIME_ONLY = 8

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)



class RngState(ctypes.Structure):
    """include/spx.h: spx_rng_state == numpy.random.get_state() of the legacy MT19937 generator."""
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32),
                ("gauss", ctypes.c_double)]

    @classmethod
    def from_numpy(cls, state=None):
        import numpy.random as npr
        st = npr.get_state() if state is None else state
        if st[0] != "MT19937":
            raise ValueError("the global numpy generator is not MT19937")
        out = cls()
        ctypes.memmove(out.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
        out.pos, out.has_gauss, out.gauss = int(st[2]), int(st[3]), float(st[4])
        return out

    def to_numpy(self):
        return ("MT19937", np.frombuffer(self.key, dtype=np.uint32).copy(), int(self.pos), int(self.has_gauss), float(self.gauss))


class SamplerCfg(ctypes.Structure):
    """include/spx.h: spx_sampler_cfg."""
    _fields_ = [("D", ctypes.c_int32), ("n_iter", ctypes.c_int32), ("noiseless", ctypes.c_int32),
                ("check_mean", ctypes.c_int32), ("amp2_prior_on_sqrt", ctypes.c_int32), ("lookahead", ctypes.c_int32),
                ("follow_props", ctypes.c_int32), ("follow_hyps", ctypes.c_int32), ("max_rows", ctypes.c_int32),
                ("noise_scale", ctypes.c_double), ("amp2_scale", ctypes.c_double), ("max_ls", ctypes.c_double),
                ("vals_min", ctypes.c_double), ("vals_max", ctypes.c_double)]


LOGPROB_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int32,
                              ctypes.POINTER(ctypes.c_double))

# every symbol include/spx.h declares: name -> (restype, argtypes)
_vp = ctypes.c_void_p
ABI = {
    "spx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "spx_create_multi": (ctypes.c_int, [_c_int32_p, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "spx_create_multi_transport": (ctypes.c_int, [_c_int32_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "spx_multi_query": (ctypes.c_int, [_vp, _c_int32_p, _c_int32_p, _c_int32_p, ctypes.c_int32]),
    "spx_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "spx_rccl_version": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int32)]),
    "spx_comm_attach": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]),
    "spx_set_partition": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]),
    "spx_destroy": (None, [_vp]),
    "spx_last_error": (ctypes.c_char_p, []),
    "spx_version": (ctypes.c_int, []),
    "spx_device_count": (ctypes.c_int, []),
    "spx_set_observations": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int32]),
    "spx_set_candidates": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64]),
    "spx_set_hypers": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int32]),
    "spx_set_time_model": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "spx_factor": (ctypes.c_int, [_vp]),
    "spx_set_fantasies": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, ctypes.c_int32]),
    "spx_ei_run": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "spx_ei_step": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "spx_get_best": (ctypes.c_int, [_vp, _c_int64_p, _c_double_p]),
    "spx_get_ei_mean": (ctypes.c_int, [_vp, _c_double_p]),
    "spx_get_ei_draws": (ctypes.c_int, [_vp, _c_double_p]),
    "spx_ei_grid": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int32,
                                   _c_double_p, ctypes.c_int64, _c_double_p, ctypes.c_int32,
                                   ctypes.c_int32, _c_double_p, _c_double_p, _c_int64_p, _c_double_p]),
    "spx_ei_per_sec_grid": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64,
                                           ctypes.c_int32, _c_double_p, ctypes.c_int64, _c_double_p,
                                           _c_double_p, ctypes.c_int32, ctypes.c_int32, _c_double_p,
                                           _c_double_p, _c_int64_p, _c_double_p]),
    "spx_get_factor": (ctypes.c_int, [_vp, ctypes.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "spx_get_factor_rows": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, _c_double_p, _c_double_p]),
    "spx_get_cross_cov": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, _c_double_p]),
    "spx_get_moments": (ctypes.c_int, [_vp, ctypes.c_int32, _c_double_p, _c_double_p]),
    "spx_get_time_mean": (ctypes.c_int, [_vp, ctypes.c_int32, _c_double_p]),
    "spx_gp_logprob": (ctypes.c_int, [_vp, _c_double_p]),
    "spx_ei_grad": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p]),
    "spx_ei_grad_batch": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int32, _c_double_p, _c_double_p]),
    "spx_sobol_grid": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_int64, ctypes.c_int64, _c_double_p, ctypes.c_int32, _c_double_p]),
    "spx_not_pd_info": (ctypes.c_int, [_vp, _c_int32_p, _c_int32_p]),
    "spx_sample_hypers": (ctypes.c_int, [_vp, ctypes.POINTER(SamplerCfg), ctypes.POINTER(RngState), _c_double_p, _c_double_p,
                                         _c_double_p, _c_int64_p]),
    "spx_sample_hypers_with": (ctypes.c_int, [LOGPROB_FN, _vp, ctypes.POINTER(SamplerCfg), ctypes.POINTER(RngState),
                                              _c_double_p, _c_double_p, _c_double_p, _c_int64_p]),
    "spx_rng_draw": (ctypes.c_int, [ctypes.POINTER(RngState), ctypes.c_int32, _c_double_p, ctypes.c_int32, _c_double_p,
                                    ctypes.c_int32, _c_int32_p]),
    "spx_get_timings": (ctypes.c_int, [_vp, _c_double_p, _c_int64_p, ctypes.c_int]),
    "spx_get_stat": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "spx_timing_name": (ctypes.c_char_p, [ctypes.c_int]),
    "spx_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
}

_lib = None


def default_lib_path():
    return os.environ.get("SPX_LIB") or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "libspx.so")


def load_library(path=None):
    """dlopen libspx.so and declare every prototype.  Raises OSError when the
    library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
    or `make -C spearmint_amd/csrc`)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or default_lib_path()
    if not os.path.exists(p):
        raise OSError("libspx.so not found at %s -- build it with `make -C spearmint_amd/csrc` "
                      "(there is no CPU fallback)" % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class SpxError(RuntimeError):
    pass


def _sampler_result(lib, rc, rows, stats):
    st = {"calls": int(stats[0]), "rows": int(stats[1]), "moves": int(stats[2]), "free_moves": int(stats[3]),
          "iterations": int(stats[4]), "calls_by_rows": [int(v) for v in stats[5:39]], "ns_in_calls": int(stats[39]), "ns_total": int(stats[40])}
    if rc == SPX_OK:
        return rows, st
    msg = lib.spx_last_error()
    msg = msg.decode("utf-8", "replace") if isinstance(msg, bytes) else str(msg)
    if rc == SPX_ERR_NOT_PD:
        err = LinAlgError(msg)
    elif rc in (SPX_ERR_SLICE_NAN, SPX_ERR_SLICE_ZERO):
        from .util import SliceSamplerError
        err = SliceSamplerError(msg)
    elif rc == SPX_ERR_ARG:
        err = ValueError(msg)
    else:
        err = SpxError(msg)
    err.rows_done, err.stats = rows[:st["iterations"]], st
    raise err


TRANSPORT_NAMES = {0: "none", 1: "rccl", 2: "host"}


class Engine(object):
    """One handle of libspx: one GPU (``device=``) or several GPUs of this node behind the same
    calls (``devices=[...]`` -> spx_create_multi: candidates sharded over the devices, one RCCL
    all-gather of {best EI, index} records per EI pass).  Mirrors the resident-data C API.

    Typical use (what the choosers do, GPEIOptChooser.py:331-341 + :294):

        eng = Engine(device=0)
        best, value, ei_mean, overall_ei = eng.ei_grid(comp, vals, cand, hypers, want_draws=True)
    """

    def __init__(self, device=0, lib=None, devices=None, transport=None):
        """transport (with devices=): None = by the device list, "rccl" / "host" = spx_create_multi_transport."""
        self._lib = load_library(lib)
        self._handle = _vp()
        # The handle, its HIP streams and device buffers belong to the process that created them.  The
        # reference's drivers fork AFTER the chooser has run (job processes from the main loop,
        # spearmint/driver/local.py:9-44; the status web server, main.py:126-141): a child inherits this
        # object but not a usable HIP context, so every call from another pid is refused (`_h` below) and
        # such a copy never destroys the parent's handle.
        self._pid = os.getpid()
        if devices is not None:
            devs = [int(d) for d in devices]
            if not devs:
                raise ValueError("Engine needs at least one device")
            arr = (ctypes.c_int32 * len(devs))(*devs)
            if transport is None:
                self._check(self._lib.spx_create_multi(arr, len(devs), ctypes.byref(self._handle)))
            else:
                code = {"rccl": 1, "host": 2}[transport]
                self._check(self._lib.spx_create_multi_transport(arr, len(devs), code, ctypes.byref(self._handle)))
            self.devices = devs
            self.device = devs[0]
        else:
            self._check(self._lib.spx_create(int(device), ctypes.byref(self._handle)))
            self.device = int(device)
            self.devices = [self.device]
        self.N = self.M = self.D = self.H = 0

    @property
    def _h(self):
        """The C handle -- only in the process that created it."""
        if self._pid != os.getpid():
            raise SpxError("this Engine's HIP context belongs to pid %d and is not usable in pid %d (a forked "
                           "child): create a new Engine in this process" % (self._pid, os.getpid()))
        if self._handle is None or not self._handle:
            raise SpxError("Engine is closed")
        return self._handle

    def owned_by_this_process(self):
        return self._pid == os.getpid()

    def comm_unique_id(self):
        """128 bytes identifying a new RCCL communicator (call on one rank, ship to the others)."""
        buf = ctypes.create_string_buffer(128)
        self._check(self._lib.spx_comm_unique_id(buf))
        return buf.raw

    def comm_attach(self, uid, nranks, rank):
        """One process per GPU: join the communicator `uid`; every later ei_run ends with the library's own
        ncclAllGather of the ranks' {best EI, index} records and best() returns the global winner."""
        if len(uid) != 128:
            raise ValueError("uid must be the 128 bytes of comm_unique_id()")
        self._check(self._lib.spx_comm_attach(self._h, uid, int(nranks), int(rank)))

    def set_partition(self, hyper_shards, M_total=0, H_total=0):
        """Optional 2-D partition (draws x candidates, one all-reduce(SUM) of the EI-sum vector) -- include/spx.h.
        Multi-device handle: hyper_shards only (call before set_hypers / set_candidates).  With a communicator
        attached: this rank's shards are set by the caller, M_total / H_total are the totals."""
        self._check(self._lib.spx_set_partition(self._h, int(hyper_shards), int(M_total), int(H_total)))

    def transport(self):
        """"none" (one GPU), "rccl" (ncclAllGather over the devices) or "host" (repeated device
        ids: records staged through host memory)."""
        t = ctypes.c_int32(0)
        self._check(self._lib.spx_multi_query(self._h, None, ctypes.byref(t), None, 0))
        return TRANSPORT_NAMES.get(int(t.value), "?")

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc == SPX_OK:
            return
        msg = self._lib.spx_last_error()
        msg = msg.decode("utf-8", "replace") if isinstance(msg, bytes) else str(msg)
        if rc == SPX_ERR_NOT_PD:
            raise LinAlgError(msg)  # what spla.cholesky raises in the reference
        if rc == SPX_ERR_ARG:
            raise ValueError(msg)
        raise SpxError(msg)

    def close(self):
        h = self.__dict__.get("_handle")
        if h is not None and h:
            if self.__dict__.get("_pid") == os.getpid():   # a forked copy must leave the parent's handle alone
                self._lib.spx_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        raise TypeError("Engine holds a GPU context and cannot be pickled; choosers drop it in __getstate__")

    # -- resident data ----------------------------------------------------
    def set_observations(self, comp, vals):
        comp = _f64(comp)
        vals = _f64(vals).ravel()
        if comp.ndim != 2 or comp.shape[0] != vals.shape[0]:
            raise ValueError("comp must be (N, D) and vals (N,)")
        self.N, self.D = comp.shape
        self._check(self._lib.spx_set_observations(self._h, _dp(comp), _dp(vals), self.N, self.D))

    def set_candidates(self, cand, index_base=0):
        cand = _f64(cand)
        if cand.ndim != 2:
            raise ValueError("cand must be (M, D)")
        self.M = cand.shape[0]
        self._check(self._lib.spx_set_candidates(self._h, _dp(cand), self.M, cand.shape[1], int(index_base)))

    def set_hypers(self, hypers):
        hypers = _f64(np.atleast_2d(hypers))
        if hypers.shape[1] != 3 + self.D:
            raise ValueError("hypers must be (H, 3 + D) rows [mean, noise, amp2, ls...]")
        self.H = hypers.shape[0]
        self._check(self._lib.spx_set_hypers(self._h, _dp(hypers), self.H))

    def set_time_model(self, log_durs, time_hypers):
        if log_durs is None:
            self._check(self._lib.spx_set_time_model(self._h, None, None))
            return
        log_durs = _f64(log_durs).ravel()
        time_hypers = _f64(np.atleast_2d(time_hypers))
        if log_durs.shape[0] != self.N or time_hypers.shape != (self.H, 3 + self.D):
            raise ValueError("log_durs must be (N,), time_hypers (H, 3 + D)")
        self._check(self._lib.spx_set_time_model(self._h, _dp(log_durs), _dp(time_hypers)))

    def set_option(self, name, value):
        self._check(self._lib.spx_set_option(self._h, name.encode("ascii"), int(value)))

    def set_covar(self, name):
        """The GP's correlation function by its gp.py name (the choosers' covar=)."""
        if name not in COVAR:
            raise AttributeError("no covariance function %r (gp.py has %s)" % (name, ", ".join(sorted(COVAR))))
        self.set_option("covar", COVAR[name])

    # -- hot path ---------------------------------------------------------
    def factor(self):
        self._check(self._lib.spx_factor(self._h))

    def set_fantasies(self, fant, bests):
        """fant (H, n, S) fantasy value columns per draw, bests (H, S) -- see include/spx.h."""
        if fant is None:
            self._check(self._lib.spx_set_fantasies(self._h, None, None, 0))
            return
        fant = _f64(fant)
        bests = _f64(bests)
        if fant.ndim != 3 or fant.shape[0] != self.H or fant.shape[1] != self.N or bests.shape != (self.H, fant.shape[2]):
            raise ValueError("fant must be (H, n, S) and bests (H, S)")
        self._check(self._lib.spx_set_fantasies(self._h, _dp(fant), _dp(bests), fant.shape[2]))

    def ei_run(self, flags=0):
        self._check(self._lib.spx_ei_run(self._h, int(flags)))

    def ei_step(self, flags=0):
        """factor() + ei_run() with one host synchronisation (include/spx.h: spx_ei_step)."""
        self._check(self._lib.spx_ei_step(self._h, int(flags)))

    def best(self):
        idx = ctypes.c_int64(-1)
        val = ctypes.c_double(0.0)
        self._check(self._lib.spx_get_best(self._h, ctypes.byref(idx), ctypes.byref(val)))
        return int(idx.value), float(val.value)

    def ei_mean(self):
        out = np.empty(self.M)
        self._check(self._lib.spx_get_ei_mean(self._h, _dp(out)))
        return out

    def ei_draws(self):
        out = np.empty((self.M, self.H))
        self._check(self._lib.spx_get_ei_draws(self._h, _dp(out)))
        return out

    # -- one-shot (host buffers in, results out) ---------------------------
    def ei_grid(self, comp, vals, cand, hypers, want_mean=True, want_draws=False, flags=0):
        """(best_index, best_value, ei_mean or None, overall_ei[M,H] or None)"""
        comp = _f64(comp); vals = _f64(vals).ravel(); cand = _f64(cand)
        hypers = _f64(np.atleast_2d(hypers))
        N, D = comp.shape
        M, H = cand.shape[0], hypers.shape[0]
        if hypers.shape[1] != 3 + D or cand.shape[1] != D or vals.shape[0] != N:
            raise ValueError("shape mismatch")
        mean = np.empty(M) if want_mean else None
        draws = np.empty((M, H)) if want_draws else None
        idx = ctypes.c_int64(-1); val = ctypes.c_double(0.0)
        self._check(self._lib.spx_ei_grid(self._h, _dp(comp), _dp(vals), N, D, _dp(cand), M, _dp(hypers),
                                          H, int(flags), _dp(mean), _dp(draws),
                                          ctypes.byref(idx), ctypes.byref(val)))
        self.N, self.D, self.M, self.H = N, D, M, H
        return int(idx.value), float(val.value), mean, draws

    def ei_per_sec_grid(self, comp, vals, log_durs, cand, hypers, time_hypers,
                        want_mean=True, want_draws=False, flags=0):
        comp = _f64(comp); vals = _f64(vals).ravel(); cand = _f64(cand)
        log_durs = _f64(log_durs).ravel()
        hypers = _f64(np.atleast_2d(hypers)); time_hypers = _f64(np.atleast_2d(time_hypers))
        N, D = comp.shape
        M, H = cand.shape[0], hypers.shape[0]
        if (hypers.shape[1] != 3 + D or time_hypers.shape != hypers.shape or cand.shape[1] != D
                or vals.shape[0] != N or log_durs.shape[0] != N):
            raise ValueError("shape mismatch")
        mean = np.empty(M) if want_mean else None
        draws = np.empty((M, H)) if want_draws else None
        idx = ctypes.c_int64(-1); val = ctypes.c_double(0.0)
        self._check(self._lib.spx_ei_per_sec_grid(self._h, _dp(comp), _dp(vals), _dp(log_durs), N, D,
                                                  _dp(cand), M, _dp(hypers), _dp(time_hypers), H,
                                                  int(flags), _dp(mean), _dp(draws),
                                                  ctypes.byref(idx), ctypes.byref(val)))
        self.N, self.D, self.M, self.H = N, D, M, H
        return int(idx.value), float(val.value), mean, draws

    # -- building blocks ---------------------------------------------------
    def get_factor(self, draw, want_K=True, want_L=True, want_alpha=True):
        N = self.N
        K = np.empty((N, N)) if want_K else None
        L = np.empty((N, N)) if want_L else None
        a = np.empty(N) if want_alpha else None
        self._check(self._lib.spx_get_factor(self._h, int(draw), _dp(K), _dp(L), _dp(a)))
        return K, L, a

    def get_factor_rows(self, draw, row0, nrows, want_gamma=True):
        """Rows [row0, row0 + nrows) of L (nrows x N) and gamma = L^-1 (vals - mean) (N) of one draw."""
        rows = np.empty((int(nrows), self.N))
        gam = np.empty(self.N) if want_gamma else None
        self._check(self._lib.spx_get_factor_rows(self._h, int(draw), int(row0), int(nrows), _dp(rows),
                                                  _dp(gam) if want_gamma else None))
        return rows, gam

    def get_cross_cov(self, draw, c0=0, nc=None):
        nc = self.M - c0 if nc is None else nc
        out = np.empty((self.N, nc))
        self._check(self._lib.spx_get_cross_cov(self._h, int(draw), int(c0), int(nc), _dp(out)))
        return out

    def get_moments(self, draw):
        m = np.empty(self.M); v = np.empty(self.M)
        self._check(self._lib.spx_get_moments(self._h, int(draw), _dp(m), _dp(v)))
        return m, v

    def get_time_mean(self, draw):
        out = np.empty(self.M)
        self._check(self._lib.spx_get_time_mean(self._h, int(draw), _dp(out)))
        return out

    def gp_logprob(self, raise_not_pd=False):
        """Data term of the GP log posterior for every resident draw
        (-sum log diag L - 0.5 r' K^-1 r, GPEIChooser.py:281-285); -inf where the
        covariance is not positive definite, or LinAlgError if raise_not_pd
        (what spla.cholesky does inside the reference's logprob closures)."""
        out = np.empty(self.H)
        self._check(self._lib.spx_gp_logprob(self._h, _dp(out)))
        if raise_not_pd:
            draw, pivot = self.not_pd_info()
            if draw >= 0:
                raise LinAlgError("%d-th leading minor of the array is not positive definite" % (pivot + 1))
        return out

    def sample_hypers(self, cfg, hyper, hist, rng_state=None):
        """spx_sample_hypers on the resident observations: cfg.n_iter iterations of (joint move over [mean, amp2, noise],
        component-wise sweep over the length scales) -- the reference's sample_hypers (GPEIChooser.py:268-346) -- inside
        the library.  `hyper` = [mean, noise, amp2, ls...] (updated in place), `hist` = 12 doubles of bracket statistics
        (updated in place).  The GLOBAL numpy generator supplies the random stream and is left where the reference's
        would be (rng_state: use / update this RngState instead).  Returns (rows[n_iter, 3 + D], stats dict).  Raises
        what the reference raises: LinAlgError (not positive definite), SliceSamplerError (NaN / shrank to zero)."""
        import numpy.random as npr
        for name, a, n in (("hyper", hyper, 3 + int(cfg.D)), ("hist", hist, 12)):     # updated in place: no silent copies
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (n,)):
                raise ValueError("%s must be a contiguous float64 array of %d entries" % (name, n))
        rng = RngState.from_numpy() if rng_state is None else rng_state
        rows = np.empty((int(cfg.n_iter), 3 + int(cfg.D)))
        stats = np.zeros(41, dtype=np.int64)
        rc = self._lib.spx_sample_hypers(self._h, ctypes.byref(cfg), ctypes.byref(rng), _dp(hyper), _dp(rows), _dp(hist),
                                         stats.ctypes.data_as(_c_int64_p))
        if rng_state is None:
            npr.set_state(rng.to_numpy())
        return _sampler_result(self._lib, rc, rows, stats)

    def ei_grad(self, point):
        """(-sum_h EI_h(x), gradient) at one point for the resident factorisation --
        the L-BFGS objective of GPEIOptChooser.py:360-388."""
        x = _f64(point).ravel()
        if x.shape[0] != self.D:
            raise ValueError("point must have D entries")
        f = ctypes.c_double(0.0)
        g = np.empty(self.D)
        self._check(self._lib.spx_ei_grad(self._h, _dp(x), ctypes.byref(f), _dp(g)))
        return float(f.value), g

    def ei_grad_batch(self, points):
        """The same objective at P points in one call: (f[P], grad[P, D]).  A point's result does
        not depend on which other points share the call.  With fantasies set the objective is the
        pending branch of the reference (GPEIOptChooser.py:441-525)."""
        x = _f64(np.atleast_2d(points))
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError("points must be (P, D)")
        f = np.empty(x.shape[0])
        g = np.empty(x.shape)
        self._check(self._lib.spx_ei_grad_batch(self._h, _dp(x), x.shape[0], _dp(f), _dp(g)))
        return f, g

    def sobol_grid(self, dirs, dim, n, skip, fetch=True, as_candidates=False):
        """The reference's Sobol grid, generated on the GPU: returns (grid, kernel_ms) with
        grid (n, dim) == np.transpose(i4_sobol_generate(dim, n, skip)) bit for bit
        (ExperimentGrid.py:192-196), or None when fetch is False.  dirs: (dim_max, 30) uint32
        direction integers (spearmint_amd.sobol.load_dirs).  as_candidates leaves the grid
        resident as the candidate set."""
        dirs = np.ascontiguousarray(dirs, dtype=np.uint32)
        if dirs.ndim != 2 or dirs.shape[1] != 30:
            raise ValueError("dirs must be (dim_max, 30) uint32")
        out = np.empty((int(n), int(dim))) if fetch else None
        ms = ctypes.c_double(0.0)
        self._check(self._lib.spx_sobol_grid(
            self._h, dirs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), dirs.shape[0], int(dim), int(n),
            int(skip), _dp(out) if fetch else None, 1 if as_candidates else 0, ctypes.byref(ms)))
        if as_candidates:
            self.M = int(n)
            if not self.D:
                self.D = int(dim)
        return out, float(ms.value)

    def not_pd_info(self):
        d = ctypes.c_int32(-1); p = ctypes.c_int32(-1)
        self._check(self._lib.spx_not_pd_info(self._h, ctypes.byref(d), ctypes.byref(p)))
        return int(d.value), int(p.value)

    def last_warning(self):
        """The warning the data-flow factorisation's fallback left in spx_last_error() (text starting with "warning:",
        include/spx.h: spx_get_stat) since the previous call of this method, or None.  (spx_last_error() is sticky; a new
        warning is told from an old one by the handle's fallback counter.)"""
        try:
            n = self.stat("flow_fallbacks")
        except (ValueError, SpxError):      # (a multi-device handle keeps its counters on the per-device handles)
            return None
        if n == self.__dict__.get("_warned_fallbacks", 0):
            return None
        self._warned_fallbacks = n
        msg = self._lib.spx_last_error()
        msg = msg.decode("utf-8", "replace") if isinstance(msg, bytes) else str(msg or "")
        return msg if msg.startswith("warning:") else "warning: the data-flow factorisation fell back (%d so far)" % n

    def stat(self, name):
        """A counter of the handle: "flow_fallbacks", "flow_rearms", "flow_enabled", "n_cu", "last_step_fused", "ranks_seen" (include/spx.h: spx_get_stat)."""
        v = ctypes.c_int64(0)
        self._check(self._lib.spx_get_stat(self._h, name.encode("ascii"), ctypes.byref(v)))
        return int(v.value)

    def timings(self):
        """{stage: (ms_total, launches)} accumulated since set_option('timing', 1)."""
        n = self._lib.spx_get_timings(self._h, None, None, 0)
        ms = (ctypes.c_double * n)(); cnt = (ctypes.c_int64 * n)()
        self._lib.spx_get_timings(self._h, ms, cnt, n)
        out = {}
        for i in range(n):
            name = self._lib.spx_timing_name(i)
            name = name.decode("ascii") if isinstance(name, bytes) else name
            out[name] = (float(ms[i]), int(cnt[i]))
        return out


def device_count(lib=None):
    n = load_library(lib).spx_device_count()
    return n if n > 0 else 0


def rccl_version(lib=None):
    """Version code of the librccl that libspx binds (22707 = 2.27.7); SpxError when it cannot be loaded or is outside
    the range the hand-declared binding accepts (include/spx.h: spx_rccl_version)."""
    l = load_library(lib)
    v = ctypes.c_int32(0)
    if l.spx_rccl_version(ctypes.byref(v)) != 0:
        raise SpxError(l.spx_last_error().decode("utf-8", "replace"))
    return int(v.value)


class MultiEngine(Engine):
    """Several GPUs driven from ONE process (the unmodified Spearmint driver is a single
    process): an Engine over spx_create_multi.  Sharding, the per-device host threads and the
    RCCL all-gather all live inside libspx (csrc/spx_multi.hip); this class only keeps the
    round-1 constructor signature."""

    def __init__(self, devices, lib=None, transport=None):
        Engine.__init__(self, lib=lib, devices=list(devices), transport=transport)

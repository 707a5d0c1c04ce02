"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/scipy restatement (Python 3) of the arithmetic on Spearmint's GP-EI
hot path, kept in the reference's operation order so that it can be compared
value-for-value with the real reference.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``spearmint_amd/`` does.

Parity status: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against *outputs of the
reference itself run in the build container*: ``oracle/make_golden.py`` runs
the lib2to3-converted reference (``oracle/ref_py3.py``) on seeded inputs and
commits the vectors under ``tests/golden/``; ``tests/test_oracle_golden.py``
asserts this module reproduces them bit-for-bit (same numpy/scipy build) or to
1e-13 (other BLAS builds).

Citations are relative to /root/reference/spearmint/spearmint/ ("S/").
"""
import re

import numpy as np
import numpy.random as npr
import scipy.linalg as spla
import scipy.stats as sps

SQRT_3 = np.sqrt(3.0)  # S/gp.py:31
SQRT_5 = np.sqrt(5.0)  # S/gp.py:32

# The choosers pick their correlation function by name, getattr(gp, covar) (GPEIChooser.py:52); every
# function below follows the module's ACTIVE one (default Matern52, the north star's).  Tests select
# another with ``with covar("ARDSE"): ...``.
COVARS = ("Matern52", "Matern32", "ARDSE", "SE")
_active_covar = "Matern52"


class covar(object):
    """Context manager: the covariance function (a name from gp.py) used by everything in this module."""

    def __init__(self, name):
        if name not in COVARS:
            raise AttributeError("gp has no covariance function %r" % (name,))
        self.name = name

    def __enter__(self):
        global _active_covar
        self.prev, _active_covar = _active_covar, self.name
        return self

    def __exit__(self, *exc):
        global _active_covar
        _active_covar = self.prev
        return False


# --------------------------------------------------------------------------
# covariance  (S/gp.py:34-54, :120-127; S/chooser/GPEIChooser.py:117-122)
# --------------------------------------------------------------------------
def dist2(ls, x1, x2=None):
    """ARD-scaled pairwise squared distance, GEMM form.  S/gp.py:34-54."""
    xx1 = x1 / ls
    xx2 = xx1 if x2 is None else x2 / ls
    g = np.dot(xx1, 2 * xx2.T)
    s1 = np.sum(xx1 * xx1, axis=1)[:, np.newaxis]
    s2 = np.sum(xx2 * xx2, axis=1)[:, np.newaxis].T
    return np.maximum(-(g - s1 - s2), 0.0)


def matern52(ls, x1, x2=None):
    """ARD Matern-5/2 correlation.  S/gp.py:120-127."""
    r2 = np.abs(dist2(ls, x1, x2))
    r = np.sqrt(r2)
    return (1.0 + SQRT_5 * r + (5.0 / 3.0) * r2) * np.exp(-SQRT_5 * r)


def matern32(ls, x1, x2=None):
    """S/gp.py:107-113."""
    r = np.sqrt(dist2(ls, x1, x2))
    return (1 + SQRT_3 * r) * np.exp(-SQRT_3 * r)


def ardse(ls, x1, x2=None):
    """S/gp.py:95-100."""
    return np.exp(-0.5 * dist2(ls, x1, x2))


def se(ls, x1, x2=None):
    """S/gp.py:87-93: the length scales are replaced by ones."""
    return np.exp(-0.5 * dist2(np.ones(np.shape(ls)), x1, x2))


def corr(ls, x1, x2=None):
    """self.cov_func = getattr(gp, covar) (GPEIChooser.py:52) for the active covariance."""
    return {"Matern52": matern52, "Matern32": matern32, "ARDSE": ardse, "SE": se}[_active_covar](ls, x1, x2)


def cov(amp2, ls, x1, x2=None):
    """Chooser covariance: jittered self-cov or plain cross-cov.
    S/chooser/GPEIChooser.py:117-122 (= GPEIOptChooser.py:207-212,
    GPEIperSecChooser.py:145-150)."""
    if x2 is None:
        return amp2 * (corr(ls, x1, None) + 1e-6 * np.eye(x1.shape[0]))
    return amp2 * corr(ls, x1, x2)


# --------------------------------------------------------------------------
# EI, no pending experiments  (S/chooser/GPEIChooser.py:178-208)
# --------------------------------------------------------------------------
def unpack_hyper(hyper):
    """hyper row layout used across the C ABI: [mean, noise, amp2, ls[0..D)]
    -- the tuple order of GPEIOptChooser.py:628."""
    hyper = np.asarray(hyper, dtype=np.float64)
    return hyper[0], hyper[1], hyper[2], hyper[3:]


def posterior(comp, vals, hyper):
    """K, L, alpha for one hyper draw.  GPEIChooser.py:186-194."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    comp_cov = cov(amp2, ls, comp)
    obsv_cov = comp_cov + noise * np.eye(comp.shape[0])
    obsv_chol = spla.cholesky(obsv_cov, lower=True)
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    return obsv_cov, obsv_chol, alpha


def compute_ei(comp, cand, vals, hyper, stages=None):
    """EI of every candidate under one hyper draw (no-pending branch).
    GPEIChooser.py:178-208 == GPEIOptChooser.py:527-557."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    best = np.min(vals)
    obsv_cov, obsv_chol, alpha = posterior(comp, vals, hyper)
    cand_cross = cov(amp2, ls, comp, cand)
    beta = spla.solve_triangular(obsv_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        func_s = np.sqrt(func_v)
        u = (best - func_m) / func_s
        ncdf = sps.norm.cdf(u)
        npdf = sps.norm.pdf(u)
        ei = func_s * (u * ncdf + npdf)
    if stages is not None:
        stages.update(K=obsv_cov, L=obsv_chol, alpha=alpha, Kstar=cand_cross,
                      func_m=func_m, func_v=func_v)
    return ei


def ei_over_hypers(comp, cand, vals, hypers):
    """overall_ei[M, H].  GPEIOptChooser.py:331-341 / GPEIChooser.py:143-151."""
    hypers = np.atleast_2d(hypers)
    out = np.zeros((cand.shape[0], hypers.shape[0]))
    for h in range(hypers.shape[0]):
        out[:, h] = compute_ei(comp, cand, vals, hypers[h])
    return out


def choose(overall_ei):
    """argmax of the MCMC-mean EI: first NaN wins, else first max.
    GPEIChooser.py:153."""
    return int(np.argmax(np.mean(overall_ei, axis=1)))


def ei_grid_chunked(comp, cand, vals, hypers, chunk=20000):
    """Same result as ei_over_hypers+choose, candidates processed in column
    chunks so that the N x M temporaries stay bounded.
    The factorisation is done once per draw; per-column results are unchanged
    up to BLAS blocking."""
    hypers = np.atleast_2d(hypers)
    M, H = cand.shape[0], hypers.shape[0]
    out = np.zeros((M, H))
    best = np.min(vals)
    for h in range(H):
        mean, noise, amp2, ls = unpack_hyper(hypers[h])
        _, chol, alpha = posterior(comp, vals, hypers[h])
        for c0 in range(0, M, chunk):
            cc = cand[c0:c0 + chunk]
            cross = cov(amp2, ls, comp, cc)
            beta = spla.solve_triangular(chol, cross, lower=True)
            func_m = np.dot(cross.T, alpha) + mean
            func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
            with np.errstate(invalid="ignore", divide="ignore"):
                func_s = np.sqrt(func_v)
                u = (best - func_m) / func_s
                out[c0:c0 + chunk, h] = func_s * (u * sps.norm.cdf(u) + sps.norm.pdf(u))
    return out


# --------------------------------------------------------------------------
# EI with pending experiments ("fantasies")  (GPEIChooser.py:209-266)
# --------------------------------------------------------------------------
def compute_ei_pending(comp, pend, cand, vals, hyper, randn_ps):
    """Pending branch; ``randn_ps`` is the (P, S) standard-normal matrix the
    reference draws with npr.randn (GPEIChooser.py:238; GPEIOptChooser.py:588
    replays a saved RNG state first)."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    n = comp.shape[0]
    comp_pend = np.concatenate((comp, pend))
    cp_cov = cov(amp2, ls, comp_pend) + noise * np.eye(comp_pend.shape[0])
    cp_chol = spla.cholesky(cp_cov, lower=True)
    pend_cross = cov(amp2, ls, comp, pend)
    pend_kappa = cov(amp2, ls, pend)
    obsv_chol = cp_chol[:n, :n]
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.cho_solve((obsv_chol, True), pend_cross)
    pend_m = np.dot(pend_cross.T, alpha) + mean
    pend_K = pend_kappa - np.dot(pend_cross.T, beta)
    pend_chol = spla.cholesky(pend_K, lower=True)
    pend_fant = np.dot(pend_chol, randn_ps) + pend_m[:, None]
    S = randn_ps.shape[1]
    fant_vals = np.concatenate((np.tile(vals[:, np.newaxis], (1, S)), pend_fant))
    bests = np.min(fant_vals, axis=0)
    cand_cross = cov(amp2, ls, comp_pend, cand)
    alpha = spla.cho_solve((cp_chol, True), fant_vals - mean)
    beta = spla.solve_triangular(cp_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        func_s = np.sqrt(func_v[:, np.newaxis])
        u = (bests[np.newaxis, :] - func_m) / func_s
        ei = func_s * (u * sps.norm.cdf(u) + sps.norm.pdf(u))
    return np.mean(ei, axis=1)


def fantasize(comp, pend, vals, hyper, randn_ps):
    """First half of the pending branch (GPEIChooser.py:213-249): the S joint fantasy outcomes of
    the P pending jobs.  Returns fant_vals ((N+P) x S) and bests (S,)."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    n = comp.shape[0]
    comp_pend = np.concatenate((comp, pend))
    cp_chol = spla.cholesky(cov(amp2, ls, comp_pend) + noise * np.eye(comp_pend.shape[0]), lower=True)
    pend_cross = cov(amp2, ls, comp, pend)
    pend_kappa = cov(amp2, ls, pend)
    obsv_chol = cp_chol[:n, :n]
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.cho_solve((obsv_chol, True), pend_cross)
    pend_m = np.dot(pend_cross.T, alpha) + mean
    pend_K = pend_kappa - np.dot(pend_cross.T, beta)
    pend_chol = spla.cholesky(pend_K, lower=True)
    pend_fant = np.dot(pend_chol, randn_ps) + pend_m[:, None]
    fant_vals = np.concatenate((np.tile(vals[:, np.newaxis], (1, randn_ps.shape[1])), pend_fant))
    return fant_vals, np.min(fant_vals, axis=0)


def compute_ei_fantasies(comp_pend, cand, hyper, fant_vals, bests):
    """Second half of the pending branch (GPEIChooser.py:251-266) for given
    fantasy values: EI of every candidate against every fantasy, averaged."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    cp_cov = cov(amp2, ls, comp_pend) + noise * np.eye(comp_pend.shape[0])
    cp_chol = spla.cholesky(cp_cov, lower=True)
    cand_cross = cov(amp2, ls, comp_pend, cand)
    alpha = spla.cho_solve((cp_chol, True), fant_vals - mean)
    beta = spla.solve_triangular(cp_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        func_s = np.sqrt(func_v[:, np.newaxis])
        u = (bests[np.newaxis, :] - func_m) / func_s
        ei = func_s * (u * sps.norm.cdf(u) + sps.norm.pdf(u))
    return np.mean(ei, axis=1)


# --------------------------------------------------------------------------
# EI per second  (S/chooser/GPEIperSecChooser.py:437-491)
# --------------------------------------------------------------------------
def compute_ei_per_s(comp, cand, vals, log_durs, hyper, time_hyper):
    """EI / exp(predicted log-duration), no-pending branch.
    GPEIperSecChooser.py:437-491.  ``log_durs`` is log(durations[complete])
    (:176)."""
    t_mean, t_noise, t_amp2, t_ls = unpack_hyper(time_hyper)
    comp_time_cov = cov(t_amp2, t_ls, comp)
    cand_time_cross = cov(t_amp2, t_ls, comp, cand)
    obsv_time_chol = spla.cholesky(comp_time_cov + t_noise * np.eye(comp.shape[0]), lower=True)
    t_alpha = spla.cho_solve((obsv_time_chol, True), log_durs - t_mean)
    func_time_m = np.exp(np.dot(cand_time_cross.T, t_alpha) + t_mean)
    ei = compute_ei(comp, cand, vals, hyper)
    return ei / func_time_m


def ei_per_s_over_hypers(comp, cand, vals, log_durs, hypers, time_hypers, ref_compat=False):
    """overall_ei[M, H] for the per-second chooser.

    ref_compat=True reproduces GPEIperSecChooser.py:284-302 literally: the
    ``return`` sits inside the loop, so only draw 0 is evaluated and the
    other columns stay 0.  ref_compat=False is the intended all-draw form."""
    hypers = np.atleast_2d(hypers)
    time_hypers = np.atleast_2d(time_hypers)
    out = np.zeros((cand.shape[0], hypers.shape[0]))
    for h in range(hypers.shape[0]):
        out[:, h] = compute_ei_per_s(comp, cand, vals, log_durs, hypers[h], time_hypers[h])
        if ref_compat:
            return out
    return out


# --------------------------------------------------------------------------
# host-side helpers that feed the path  (S/util.py:26-93, chooser logprobs)
# --------------------------------------------------------------------------
def unpack_args(s):
    """'k=v,k=v' -> dict of strings.  S/util.py:26-32."""
    if len(s) > 1:
        eq_re = re.compile(r"\s*=\s*")
        return dict(map(lambda x: eq_re.split(x), re.compile(r"\s*,\s*").split(s)))
    return {}


def slice_sample(init_x, logprob, sigma=1.0, step_out=True, max_steps_out=1000, compwise=False):
    """Univariate slice sampler with step-out, same RNG call order as
    S/util.py:34-93 (global numpy.random)."""
    def direction_slice(direction, init_x):
        def dir_logprob(z):
            return logprob(direction * z + init_x)
        upper = sigma * npr.rand()
        lower = upper - sigma
        llh_s = np.log(npr.rand()) + dir_logprob(0.0)
        l_steps_out = 0
        u_steps_out = 0
        if step_out:
            while dir_logprob(lower) > llh_s and l_steps_out < max_steps_out:
                l_steps_out += 1
                lower -= sigma
            while dir_logprob(upper) > llh_s and u_steps_out < max_steps_out:
                u_steps_out += 1
                upper += sigma
        while True:
            new_z = (upper - lower) * npr.rand() + lower
            new_llh = dir_logprob(new_z)
            if np.isnan(new_llh):
                raise Exception("Slice sampler got a NaN")
            if new_llh > llh_s:
                break
            elif new_z < 0:
                lower = new_z
            elif new_z > 0:
                upper = new_z
            else:
                raise Exception("Slice sampler shrank to zero!")
        return new_z * direction + init_x

    if not init_x.shape:
        init_x = np.array([init_x])
    dims = init_x.shape[0]
    if compwise:
        ordering = list(range(dims))
        npr.shuffle(ordering)
        cur_x = init_x.copy()
        for d in ordering:
            direction = np.zeros((dims))
            direction[d] = 1.0
            cur_x = direction_slice(direction, cur_x)
        return cur_x
    direction = npr.randn(dims)
    direction = direction / np.sqrt(np.sum(direction ** 2))
    return direction_slice(direction, init_x)


def gp_logprob(comp, vals, mean, amp2, noise, ls):
    """-sum(log diag L) - 0.5 r' K^-1 r, the data term shared by every
    sampler closure (GPEIChooser.py:281-285, :303-306)."""
    n = comp.shape[0]
    c = amp2 * (corr(ls, comp, None) + 1e-6 * np.eye(n)) + noise * np.eye(n)
    chol = spla.cholesky(c, lower=True)
    solve = spla.cho_solve((chol, True), vals - mean)
    return -np.sum(np.log(np.diag(chol))) - 0.5 * np.dot(vals - mean, solve)


# --------------------------------------------------------------------------
# EI + gradient at a point: the objective of the local refinement
# (S/chooser/GPEIOptChooser.py:360-525; S/chooser/GPEIperSecChooser.py:322-434;
#  covariance gradients S/gp.py:56-85, :129-132)
# --------------------------------------------------------------------------
def grad_dist2(ls, x1, x2=None):
    """d r2(i,j) / d x1(i,d) for the ARD-scaled squared distance.  S/gp.py:56-85 (the numpy
    fallback branch; scipy.weave no longer exists): gX[i,j,d] = 2 (x1[i,d] - x2[j,d]) (1/ls[d])
    on the rescaled inputs."""
    if x2 is None:
        x2 = x1
    x1 = x1 / ls
    x2 = x2 / ls
    gX = np.zeros((x1.shape[0], x2.shape[0], x1.shape[1]))
    for i in range(x1.shape[0]):
        gX[i, :, :] = 2 * (x1[i, :] - x2[:, :]) * (1 / ls)
    return gX


def grad_matern52(ls, x1, x2=None):
    """S/gp.py:129-132: dk/dr2 * dr2/dx1; note r here is sqrt(dist2) without the abs."""
    r = np.sqrt(dist2(ls, x1, x2))
    grad_r2 = -(5.0 / 6.0) * np.exp(-SQRT_5 * r) * (1 + SQRT_5 * r)
    return grad_r2[:, :, np.newaxis] * grad_dist2(ls, x1, x2)


def grad_matern32(ls, x1, x2=None):
    """S/gp.py:115-118."""
    r = np.sqrt(dist2(ls, x1, x2))
    grad_r2 = -1.5 * np.exp(-SQRT_3 * r)
    return grad_r2[:, :, np.newaxis] * grad_dist2(ls, x1, x2)


def grad_ardse(ls, x1, x2=None):
    """S/gp.py:102-105."""
    r2 = dist2(ls, x1, x2)
    return -0.5 * np.exp(-0.5 * r2)[:, :, np.newaxis] * grad_dist2(ls, x1, x2)


def grad_corr(ls, x1, x2=None):
    """getattr(gp, 'grad_' + covar) (GPEIOptChooser.py:404): gp.py defines no grad_SE, so that raises."""
    if _active_covar == "SE":
        raise AttributeError("gp has no attribute 'grad_SE'")
    return {"Matern52": grad_matern52, "Matern32": grad_matern32, "ARDSE": grad_ardse}[_active_covar](ls, x1, x2)


def grad_optimize_ei(cand, comp, vals, hyper):
    """(-sum EI, gradient) at the point(s) ``cand`` under ONE hyper draw, no pending jobs.
    GPEIOptChooser.py:391-440, including its factor one half in grad_xp."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    best = np.min(vals)
    cand = np.reshape(cand, (-1, comp.shape[1]))
    comp_cov = cov(amp2, ls, comp)
    cand_cross = cov(amp2, ls, comp, cand)
    obsv_cov = comp_cov + noise * np.eye(comp.shape[0])
    obsv_chol = spla.cholesky(obsv_cov, lower=True)
    cand_cross_grad = grad_corr(ls, comp, cand)
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.solve_triangular(obsv_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    func_s = np.sqrt(func_v)
    u = (best - func_m) / func_s
    ncdf = sps.norm.cdf(u)
    npdf = sps.norm.pdf(u)
    ei = func_s * (u * ncdf + npdf)
    g_ei_m = -ncdf
    g_ei_s2 = 0.5 * npdf / func_s
    grad_cross = np.squeeze(cand_cross_grad)
    grad_xp_m = np.dot(alpha.transpose(), grad_cross)
    grad_xp_v = np.dot(-2 * spla.cho_solve((obsv_chol, True), cand_cross).transpose(), grad_cross)
    grad_xp = 0.5 * amp2 * (grad_xp_m * g_ei_m + grad_xp_v * g_ei_s2)
    return -np.sum(ei), grad_xp.flatten()


def grad_optimize_ei_pending(cand, comp, pend, vals, hyper, randn_ps):
    """The pending branch of the same function (GPEIOptChooser.py:441-525): EI averaged over
    the S fantasies and the mean of the per-fantasy gradients.  ``randn_ps`` is the (P, S)
    normal matrix the reference obtains by restoring its saved RNG state (:476-477)."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    d = comp.shape[1]
    cand = np.reshape(cand, (-1, d))
    n = comp.shape[0]
    comp_pend = np.concatenate((comp, pend))
    cp_cov = cov(amp2, ls, comp_pend) + noise * np.eye(comp_pend.shape[0])
    cp_chol = spla.cholesky(cp_cov, lower=True)
    pend_cross = cov(amp2, ls, comp, pend)
    pend_kappa = cov(amp2, ls, pend)
    obsv_chol = cp_chol[:n, :n]
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.cho_solve((obsv_chol, True), pend_cross)
    pend_m = np.dot(pend_cross.T, alpha) + mean
    pend_K = pend_kappa - np.dot(pend_cross.T, beta)
    pend_chol = spla.cholesky(pend_K, lower=True)
    pend_fant = np.dot(pend_chol, randn_ps) + pend_m[:, None]
    S = randn_ps.shape[1]
    fant_vals = np.concatenate((np.tile(vals[:, np.newaxis], (1, S)), pend_fant))
    bests = np.min(fant_vals, axis=0)
    return grad_optimize_ei_fantasies(cand, comp_pend, hyper, fant_vals, bests)


def grad_optimize_ei_fantasies(cand, comp_pend, hyper, fant_vals, bests):
    """Second half of the pending branch (GPEIOptChooser.py:494-525) for given fantasy values
    ((N+P) x S) and their minima: EI averaged over the fantasies, mean of the gradients."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    d = comp_pend.shape[1]
    cand = np.reshape(cand, (-1, d))
    cp_cov = cov(amp2, ls, comp_pend) + noise * np.eye(comp_pend.shape[0])
    cp_chol = spla.cholesky(cp_cov, lower=True)
    cand_cross = cov(amp2, ls, comp_pend, cand)
    cand_cross_grad = grad_corr(ls, comp_pend, cand)
    alpha = spla.cho_solve((cp_chol, True), fant_vals - mean)
    beta = spla.solve_triangular(cp_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    func_s = np.sqrt(func_v[:, np.newaxis])
    u = (bests[np.newaxis, :] - func_m) / func_s
    ncdf = sps.norm.cdf(u)
    npdf = sps.norm.pdf(u)
    ei = func_s * (u * ncdf + npdf)
    g_ei_m = -ncdf
    g_ei_s2 = 0.5 * npdf / func_s
    if d == 1:
        grad_cross = np.squeeze(cand_cross_grad, axis=(2,))
    else:
        grad_cross = np.squeeze(cand_cross_grad)
    grad_xp_m = np.dot(alpha.transpose(), grad_cross)
    grad_xp_v = np.dot(-2 * spla.cho_solve((cp_chol, True), cand_cross).transpose(), grad_cross)
    grad_xp = 0.5 * amp2 * (grad_xp_m * np.tile(g_ei_m, (d, 1)).T + (grad_xp_v.T * g_ei_s2).T)
    return float(-np.mean(ei, axis=1)[0]), np.mean(grad_xp, axis=0).flatten()


def grad_optimize_ei_per_s(cand, comp, vals, log_durs, hyper, time_hyper):
    """EI per second and its gradient at a point.  GPEIperSecChooser.py:349-434."""
    mean, noise, amp2, ls = unpack_hyper(hyper)
    t_mean, t_noise, t_amp2, t_ls = unpack_hyper(time_hyper)
    best = np.min(vals)
    cand = np.reshape(cand, (-1, comp.shape[1]))
    comp_time_cov = cov(t_amp2, t_ls, comp)
    cand_time_cross = cov(t_amp2, t_ls, comp, cand)
    obsv_time_chol = spla.cholesky(comp_time_cov + t_noise * np.eye(comp.shape[0]), lower=True)
    t_alpha = spla.cho_solve((obsv_time_chol, True), log_durs - t_mean)
    func_time_m = np.exp(np.dot(cand_time_cross.T, t_alpha) + t_mean)
    grad_cross_t = np.squeeze(grad_corr(t_ls, comp, cand))
    comp_cov = cov(amp2, ls, comp)
    cand_cross = cov(amp2, ls, comp, cand)
    obsv_chol = spla.cholesky(comp_cov + noise * np.eye(comp.shape[0]), lower=True)
    cand_cross_grad = grad_corr(ls, comp, cand)
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.solve_triangular(obsv_chol, cand_cross, lower=True)
    func_m = np.dot(cand_cross.T, alpha) + mean
    func_v = amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
    func_s = np.sqrt(func_v)
    u = (best - func_m) / func_s
    ncdf = sps.norm.cdf(u)
    npdf = sps.norm.pdf(u)
    ei = func_s * (u * ncdf + npdf)
    ei_per_s = -np.sum(ei / func_time_m)
    grad_time_xp_m = np.dot(t_alpha.transpose(), grad_cross_t)
    g_ei_m = -ncdf
    g_ei_s2 = 0.5 * npdf / func_s
    grad_cross = np.squeeze(cand_cross_grad)
    grad_xp_m = np.dot(alpha.transpose(), grad_cross)
    grad_xp_v = np.dot(-2 * spla.cho_solve((obsv_chol, True), cand_cross).transpose(), grad_cross)
    grad_xp = 0.5 * amp2 * (grad_xp_m * g_ei_m + grad_xp_v * g_ei_s2)
    grad_time_xp_m = 0.5 * t_amp2 * grad_time_xp_m * func_time_m
    grad_xp = (func_time_m * grad_xp - ei * grad_time_xp_m) / (func_time_m ** 2)
    return ei_per_s, grad_xp.flatten()


def grad_optimize_ei_over_hypers(cand, comp, vals, hypers, pend=None, randn_ps=None,
                                 log_durs=None, time_hypers=None):
    """Summed over the hyper draws in draw order: the L-BFGS-B objective.
    GPEIOptChooser.py:360-388; GPEIperSecChooser.py:322-347."""
    summed_ei = 0
    summed_grad = np.zeros(np.asarray(cand).shape).flatten()
    hypers = np.atleast_2d(hypers)
    for h in range(hypers.shape[0]):
        if time_hypers is not None:
            ei, g = grad_optimize_ei_per_s(cand, comp, vals, log_durs, hypers[h], np.atleast_2d(time_hypers)[h])
        elif pend is not None and pend.shape[0] > 0:
            ei, g = grad_optimize_ei_pending(cand, comp, pend, vals, hypers[h], randn_ps)
        else:
            ei, g = grad_optimize_ei(cand, comp, vals, hypers[h])
        summed_grad = summed_grad + g
        summed_ei += ei
    return summed_ei, summed_grad

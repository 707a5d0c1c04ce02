"""Host-side (numpy) GP pieces that stay on the CPU by design.

What lives here is everything the choosers do *outside* the EI-grid hot path
(SURVEY.md section 8(a) marks it out of scope / "next"):

  * the log-posterior evaluated thousands of times by the sequential slice
    sampler (spearmint/spearmint/chooser/GPEIChooser.py:276-346), and
  * the EI value + gradient used by the L-BFGS-B refinement of the 20 best
    candidates (GPEIOptChooser.py:360-440), a 20-point problem.

The EI grid itself (candidates x hyper draws) never comes through this module:
it goes to libspx.so via spearmint_amd.engine, and there is no code path that
substitutes these functions for it.
"""
from __future__ import absolute_import, print_function

import numpy as np
import scipy.linalg as spla
import scipy.stats as sps

ROOT5 = np.sqrt(5.0)
ROOT3 = np.sqrt(3.0)
COVARS = ("Matern52", "Matern32", "ARDSE", "SE")   # the reference's covar= choices (gp.py:87-132)


def scaled_sqdist(ls, a, b=None):
    """Pairwise squared distance of a/ls and b/ls in the expanded (GEMM) form,
    clamped at zero (gp.py:34-54)."""
    sa = a / ls
    sb = sa if b is None else b / ls
    cross = np.dot(sa, 2 * sb.T)
    na = np.sum(sa * sa, axis=1)[:, None]
    nb = np.sum(sb * sb, axis=1)[None, :]
    return np.maximum(-(cross - na - nb), 0.0)


def matern52(ls, a, b=None):
    """gp.py:120-127."""
    r2 = np.abs(scaled_sqdist(ls, a, b))
    r = np.sqrt(r2)
    return (1.0 + ROOT5 * r + (5.0 / 3.0) * r2) * np.exp(-ROOT5 * r)


def matern52_grad_wrt_first(ls, a, b):
    """d k(a_i, b_j) / d a_i, shape (Na, Nb, D): dk/dr2 * dr2/da
    (gp.py:129-132 with gp.py:56-85)."""
    sa = a / ls
    sb = b / ls
    r = np.sqrt(scaled_sqdist(ls, a, b))
    dk_dr2 = -(5.0 / 6.0) * np.exp(-ROOT5 * r) * (1 + ROOT5 * r)
    dr2_da = 2.0 * (sa[:, None, :] - sb[None, :, :]) * (1.0 / ls)
    return dk_dr2[:, :, None] * dr2_da


def corr(covar, ls, a, b=None):
    """The correlation function the chooser was built with: getattr(gp, covar) (GPEIChooser.py:52)."""
    if covar == "Matern52":
        return matern52(ls, a, b)
    if covar == "Matern32":                                   # gp.py:107-113
        r = np.sqrt(scaled_sqdist(ls, a, b))
        return (1 + ROOT3 * r) * np.exp(-ROOT3 * r)
    if covar == "ARDSE":                                      # gp.py:95-100
        return np.exp(-0.5 * scaled_sqdist(ls, a, b))
    if covar == "SE":                                         # gp.py:87-93: the length scales are ignored
        return np.exp(-0.5 * scaled_sqdist(np.ones(np.shape(ls)), a, b))
    raise AttributeError("no covariance function %r (gp.py has %s)" % (covar, ", ".join(COVARS)))


def corr_grad_wrt_first(covar, ls, a, b):
    """getattr(gp, 'grad_' + covar)(ls, a, b): d k(a_i, b_j) / d a_i, shape (Na, Nb, D)
    (gp.py:102-105, :115-118, :129-132 with grad_dist2 :56-85).  There is no grad_SE in gp.py: the
    reference's refinement raises AttributeError for covar=SE, and so does this."""
    if covar == "Matern52":
        return matern52_grad_wrt_first(ls, a, b)
    if covar not in ("Matern32", "ARDSE"):
        raise AttributeError("gp has no attribute 'grad_%s'" % (covar,))
    sa = a / ls
    sb = b / ls
    r2 = scaled_sqdist(ls, a, b)
    if covar == "Matern32":
        dk_dr2 = -1.5 * np.exp(-ROOT3 * np.sqrt(r2))
    else:
        dk_dr2 = -0.5 * np.exp(-0.5 * r2)
    dr2_da = 2.0 * (sa[:, None, :] - sb[None, :, :]) * (1.0 / ls)
    return dk_dr2[:, :, None] * dr2_da


def obs_cov(amp2, noise, ls, x, covar="Matern52"):
    """amp2 (k + 1e-6 I) + noise I  (GPEIChooser.py:117-122, :190)."""
    n = x.shape[0]
    return amp2 * (corr(covar, ls, x) + 1e-6 * np.eye(n)) + noise * np.eye(n)


def data_logprob(x, y, mean, amp2, noise, ls, covar="Matern52"):
    """-sum log diag L - 0.5 r^T K^-1 r  (GPEIChooser.py:281-285).
    Lets numpy.linalg.LinAlgError propagate, as the reference does."""
    chol = spla.cholesky(obs_cov(amp2, noise, ls, x, covar), lower=True)
    resid = y - mean
    sol = spla.cho_solve((chol, True), resid)
    return -np.sum(np.log(np.diag(chol))) - 0.5 * np.dot(resid, sol)


def optimize_hypers(comp, vals, covar="Matern52"):
    """ML-II point estimate of (mean, amp2, noise, ls) -- the ``mcmc_iters=0`` branch of the
    reference (gp.GP.optimize_hypers, gp.py:181-292, called from GPEIChooser.py:158-160).
    Host-side by design: it runs once per ``next`` and is not on the EI hot path.

    Restated exactly, including what is peculiar about it: the mean is pinned to mean(vals); the
    search runs in log space from (log std(vals), log 1e-3, log 1) under the box
    [-10, 10]^2 x [-10, 5]^D; and the length-scale entries of the gradient are the reference's own
    expression (:256-258), which is not the derivative of the objective -- L-BFGS-B is fed the same
    numbers, so it lands on the same point the reference does."""
    import scipy.optimize as spo
    n, dims = comp.shape
    mean = np.mean(vals)
    diffs = vals - mean
    eye = np.eye(n)
    memo = {}

    def jittered_cholesky(covmat):
        jitter = 1e-8
        while True:
            if jitter > 100000:
                return spla.cholesky(eye)
            try:
                return spla.cholesky(covmat + jitter * eye, lower=True)
            except ValueError:          # numpy.linalg.LinAlgError is a ValueError
                jitter = jitter * 1.1

    def factor(amp2, noise, ls):
        if ("corr" not in memo or memo["amp2"] != amp2 or memo["noise"] != noise
                or np.any(memo["ls"] != ls)):
            kmat = corr(covar, ls, comp)
            # cov_func(ls, comp, None, grad=True): SE pairs its value with grad_ARDSE on unit length scales (gp.py:88-91)
            grad_corr = (corr_grad_wrt_first("ARDSE", np.ones(np.shape(ls)), comp, comp) if covar == "SE"
                         else corr_grad_wrt_first(covar, ls, comp, comp))
            covmat = amp2 * (kmat + 1e-6 * eye) + noise * eye
            memo.update(corr=kmat, grad_corr=grad_corr, chol=jittered_cholesky(covmat),
                        amp2=amp2, noise=noise, ls=ls)
        return memo["chol"], memo["corr"], memo["grad_corr"]

    def unpack(h):
        return np.exp(h[0]), np.exp(h[1]), np.exp(h[2:])

    def nlogprob(h):
        amp2, noise, ls = unpack(h)
        chol = factor(amp2, noise, ls)[0]
        solve = spla.cho_solve((chol, True), diffs)
        return -(-np.sum(np.log(np.diag(chol))) - 0.5 * np.dot(diffs, solve))

    def grad_nlogprob(h):
        amp2, noise, ls = unpack(h)
        chol, corr, grad_corr = factor(amp2, noise, ls)
        solve = spla.cho_solve((chol, True), diffs)
        inv_cov = spla.cho_solve((chol, True), eye)
        jac = np.outer(solve, solve) - inv_cov
        grad = np.zeros(dims + 2)
        grad[0] = 0.5 * np.trace(np.dot(jac, corr + 1e-6 * eye)) * amp2
        grad[1] = 0.5 * np.trace(np.dot(jac, eye)) * noise
        for dd in range(dims):
            grad[dd + 2] = 1 * np.trace(np.dot(jac, -amp2 * grad_corr[:, :, dd] * comp[:, dd][:, np.newaxis]
                                               / (np.exp(ls[dd])))) * np.exp(ls[dd])
        return -grad

    start = np.zeros(dims + 2)
    start[0] = np.log(np.std(vals))
    start[1] = np.log(1e-3)
    start[2:] = np.log(np.ones(dims))
    bounds = [(-10, 10), (-10, 10)] + [(-10, 5)] * dims
    best = spo.fmin_l_bfgs_b(nlogprob, start, grad_nlogprob, args=(), bounds=bounds, disp=0)[0]
    amp2, noise, ls = unpack(best)
    return mean, amp2, noise, ls


class PointModel(object):
    """Posterior at ONE hyper draw, factorised once, for evaluating EI and its
    gradient at a handful of points during local refinement."""

    def __init__(self, comp, vals, hyper, covar="Matern52"):
        self.covar = covar
        self.mean, self.noise, self.amp2 = float(hyper[0]), float(hyper[1]), float(hyper[2])
        self.ls = np.asarray(hyper[3], dtype=float)
        self.comp = comp
        self.best = np.min(vals)
        self.chol = spla.cholesky(obs_cov(self.amp2, self.noise, self.ls, comp, covar), lower=True)
        self.alpha = spla.cho_solve((self.chol, True), vals - self.mean)

    def neg_ei_and_grad(self, x):
        """(-sum EI, gradient) at the point(s) x, in the reference's scaling:
        GPEIOptChooser.py:391-440 returns 0.5 x the analytic gradient (its
        grad_xp carries an extra factor one half); L-BFGS-B sees exactly that."""
        x = np.reshape(x, (-1, self.comp.shape[1]))
        kx = self.amp2 * corr(self.covar, self.ls, self.comp, x)
        beta = spla.solve_triangular(self.chol, kx, lower=True)
        m = np.dot(kx.T, self.alpha) + self.mean
        v = self.amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
        s = np.sqrt(v)
        u = (self.best - m) / s
        cdf = sps.norm.cdf(u)
        pdf = sps.norm.pdf(u)
        ei = s * (u * cdf + pdf)
        dk = np.squeeze(corr_grad_wrt_first(self.covar, self.ls, self.comp, x))
        d_m = np.dot(self.alpha.T, dk)
        d_v = np.dot(-2 * spla.cho_solve((self.chol, True), kx).T, dk)
        grad = 0.5 * self.amp2 * (d_m * (-cdf) + d_v * (0.5 * pdf / s))
        return -np.sum(ei), grad.flatten()


class PerSecPointModel(PointModel):
    """PointModel plus the log-duration GP (mean only) for EI-per-second
    refinement (GPEIperSecChooser.py:349-434)."""

    def __init__(self, comp, vals, log_durs, hyper, time_hyper, covar="Matern52"):
        PointModel.__init__(self, comp, vals, hyper, covar)
        self.t_mean, self.t_noise, self.t_amp2 = (float(time_hyper[0]), float(time_hyper[1]),
                                                  float(time_hyper[2]))
        self.t_ls = np.asarray(time_hyper[3], dtype=float)
        t_chol = spla.cholesky(obs_cov(self.t_amp2, self.t_noise, self.t_ls, comp, covar), lower=True)
        self.t_alpha = spla.cho_solve((t_chol, True), log_durs - self.t_mean)

    def neg_ei_and_grad(self, x):
        x = np.reshape(x, (-1, self.comp.shape[1]))
        kt = self.t_amp2 * corr(self.covar, self.t_ls, self.comp, x)
        time_m = np.exp(np.dot(kt.T, self.t_alpha) + self.t_mean)
        dkt = np.squeeze(corr_grad_wrt_first(self.covar, self.t_ls, self.comp, x))

        kx = self.amp2 * corr(self.covar, self.ls, self.comp, x)
        beta = spla.solve_triangular(self.chol, kx, lower=True)
        m = np.dot(kx.T, self.alpha) + self.mean
        v = self.amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
        s = np.sqrt(v)
        u = (self.best - m) / s
        cdf = sps.norm.cdf(u)
        pdf = sps.norm.pdf(u)
        ei = s * (u * cdf + pdf)
        dk = np.squeeze(corr_grad_wrt_first(self.covar, self.ls, self.comp, x))
        d_m = np.dot(self.alpha.T, dk)
        d_v = np.dot(-2 * spla.cho_solve((self.chol, True), kx).T, dk)
        g = 0.5 * self.amp2 * (d_m * (-cdf) + d_v * (0.5 * pdf / s))
        g_t = 0.5 * self.t_amp2 * np.dot(self.t_alpha.T, dkt) * time_m
        g = (time_m * g - ei * g_t) / (time_m ** 2)
        return -np.sum(ei / time_m), g.flatten()


def fantasize_pending(comp, pend, vals, hyper_row, obsv_chol, randn_ps, covar="Matern52"):
    """Host part of the pending branch (GPEIChooser.py:219-249), O(N^2 P):
    posterior of the P pending points given the N completed ones, S joint
    fantasy outcomes.  `obsv_chol` is the N x N leading block of the Cholesky
    factor of cov([comp; pend]) -- the reference's "sub-Cholesky" (:226) --
    which the caller fetches from the GPU factorisation.
    Returns fant_vals ((N+P) x S) and bests (S,)."""
    mean, noise, amp2 = hyper_row[0], hyper_row[1], hyper_row[2]
    ls = np.asarray(hyper_row[3:], dtype=float)
    p = pend.shape[0]
    pend_cross = amp2 * corr(covar, ls, comp, pend)
    pend_kappa = amp2 * (corr(covar, ls, pend) + 1e-6 * np.eye(p))
    alpha = spla.cho_solve((obsv_chol, True), vals - mean)
    beta = spla.cho_solve((obsv_chol, True), pend_cross)
    pend_m = np.dot(pend_cross.T, alpha) + mean
    pend_k = pend_kappa - np.dot(pend_cross.T, beta)
    pend_chol = spla.cholesky(pend_k, lower=True)
    pend_fant = np.dot(pend_chol, randn_ps) + pend_m[:, None]
    s = randn_ps.shape[1]
    fant_vals = np.concatenate((np.tile(vals[:, np.newaxis], (1, s)), pend_fant))
    return fant_vals, np.min(fant_vals, axis=0)


def fantasize_from_factor_rows(vals, hyper_row, l_rows, gamma, randn_ps):
    """The same posterior from the bottom P rows of the Cholesky factor of cov([comp; pend]) + noise I and
    gamma = L^-1 ([vals; 0] - mean), which is all the GPU path ships home (spx_get_factor_rows): with
    L = [[L_A, 0], [L21, L_S]],  L21 = (L_A^-1 B)^T  and  L_S L_S^T = C - B^T A^-1 B  (B = cov(comp, pend),
    C = pend_kappa + noise I), so   pend_m = B^T A^-1 (vals - mean) + mean = L21 gamma[:N] + mean   and
    pend_K = pend_kappa - B^T A^-1 B = L_S L_S^T - noise I   -- GPEIChooser.py:229-236 without the two
    O(N^2 P) solves against the N x N sub-Cholesky and without moving that factor to the host.

    Not the reference's arithmetic: the factorisation adds the noise to the diagonal and it is subtracted again here, so
    pend_K differs from pend_kappa - cross^T beta by O(eps (noise + amp2)) per entry (both forms cancel; neither is the
    more accurate one).  The 1e-6 amp2 jitter of pend_kappa keeps pend_K that far from singular, ten orders above the
    difference, so positive definiteness is decided the same way; the fantasies agree to 1e-9 of their scale on ordinary
    problems and to 1e-6 with noise ~ amp2 and pending points 1e-5 apart
    (tests/test_host_logic.py::test_fantasies_from_factor_rows_with_large_noise_and_nearly_duplicate_pending_points; through
    the choosers, the reference's own pending goldens reproduce)."""
    mean, noise = hyper_row[0], hyper_row[1]
    n = vals.shape[0]
    p = l_rows.shape[0]
    l21, ls_ = l_rows[:, :n], l_rows[:, n:n + p]
    pend_m = np.dot(l21, gamma[:n]) + mean
    pend_k = np.dot(ls_, ls_.T) - noise * np.eye(p)
    pend_chol = spla.cholesky(pend_k, lower=True)
    pend_fant = np.dot(pend_chol, randn_ps) + pend_m[:, None]
    s = randn_ps.shape[1]
    fant_vals = np.concatenate((np.tile(vals[:, np.newaxis], (1, s)), pend_fant))
    return fant_vals, np.min(fant_vals, axis=0)


class PendingPointModel(object):
    """EI (averaged over fantasies) and its gradient at a few points, with
    pending experiments -- the host-side refinement of GPEIOptChooser.py:441-525."""

    def __init__(self, comp, pend, vals, hyper, randn_ps, covar="Matern52"):
        self.covar = covar
        self.mean, self.noise, self.amp2 = float(hyper[0]), float(hyper[1]), float(hyper[2])
        self.ls = np.asarray(hyper[3], dtype=float)
        self.comp_pend = np.concatenate((comp, pend))
        n = comp.shape[0]
        self.chol = spla.cholesky(obs_cov(self.amp2, self.noise, self.ls, self.comp_pend, covar), lower=True)
        row = np.concatenate(([self.mean, self.noise, self.amp2], self.ls))
        fant_vals, self.bests = fantasize_pending(comp, pend, vals, row, self.chol[:n, :n], randn_ps, covar)
        self.alpha = spla.cho_solve((self.chol, True), fant_vals - self.mean)

    def neg_ei_and_grad(self, x):
        d = self.comp_pend.shape[1]
        x = np.reshape(x, (-1, d))
        kx = self.amp2 * corr(self.covar, self.ls, self.comp_pend, x)
        beta = spla.solve_triangular(self.chol, kx, lower=True)
        m = np.dot(kx.T, self.alpha) + self.mean
        v = self.amp2 * (1 + 1e-6) - np.sum(beta ** 2, axis=0)
        s = np.sqrt(v[:, np.newaxis])
        u = (self.bests[np.newaxis, :] - m) / s
        cdf = sps.norm.cdf(u)
        pdf = sps.norm.pdf(u)
        ei = s * (u * cdf + pdf)
        dk = np.squeeze(corr_grad_wrt_first(self.covar, self.ls, self.comp_pend, x), axis=1)
        d_m = np.dot(self.alpha.T, dk)
        d_v = np.dot(-2 * spla.cho_solve((self.chol, True), kx).T, dk)
        g = 0.5 * self.amp2 * (d_m * np.tile(-cdf, (d, 1)).T + (d_v.T * (0.5 * pdf / s)).T)
        return float(-np.mean(ei, axis=1)[0]), np.mean(g, axis=0).flatten()

"""Seeded synthetic workloads for the GP-EI path (BASELINE.md section 3,
SURVEY.md section 8(d)).  Pure numpy; shared by bench.py and the tests so the
HIP path and the CPU oracle see byte-identical inputs."""
import numpy as np


def synthetic_problem(N, M, D, H, seed, near=10, per_sec=False):
    """Seeded synthetic (comp, cand, vals, hypers[, log_durs, time_hypers]).

    comp ~ U[0,1]^{NxD}; cand ~ U[0,1]^{MxD} with the first ``near`` rows
    replaced by jittered copies (sigma=1e-3) of the incumbent, as
    GPEIOptChooser.py:236-238 does, to exercise the small-variance regime;
    vals = standardised sum_d sin(3 x_d) + 0.5|x-0.5|^2 + N(0, 0.01^2);
    hypers[h] = [mean, noise, amp2, ls...] with ls~U[0.3,2], amp2~LogN(0,0.5),
    noise~10^U[-4,-2], mean~U[min vals, max vals]."""
    rs = np.random.RandomState(seed)
    comp = rs.rand(N, D)
    cand = rs.rand(M, D)
    f = np.sum(np.sin(3 * comp), axis=1) + 0.5 * np.sum((comp - 0.5) ** 2, axis=1)
    vals = (f - f.mean()) / f.std() + 0.01 * rs.randn(N)
    if near:
        inc = comp[np.argmin(vals)]
        cand[:near] = np.clip(inc + 1e-3 * rs.randn(near, D), 0.0, 1.0)
    hypers = np.empty((H, 3 + D))
    hypers[:, 0] = rs.uniform(vals.min(), vals.max(), H)
    hypers[:, 1] = 10.0 ** rs.uniform(-4, -2, H)
    hypers[:, 2] = np.exp(0.5 * rs.randn(H))
    hypers[:, 3:] = rs.uniform(0.3, 2.0, (H, D))
    if not per_sec:
        return comp, cand, vals, hypers
    f2 = np.sum(np.cos(2 * comp), axis=1) + np.sum(comp, axis=1) / D
    log_durs = 0.5 * (f2 - f2.mean()) / f2.std()
    th = np.empty((H, 3 + D))
    th[:, 0] = rs.uniform(log_durs.min(), log_durs.max(), H)
    th[:, 1] = 10.0 ** rs.uniform(-4, -2, H)
    th[:, 2] = np.exp(0.5 * rs.randn(H))
    th[:, 3:] = rs.uniform(0.3, 10.0, (H, D))
    return comp, cand, vals, hypers, log_durs, th

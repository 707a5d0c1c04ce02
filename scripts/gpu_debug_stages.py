"""Stage-by-stage comparison of the HIP path with the CPU oracle (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spearmint_amd.engine import Engine, FLAG_KEEP_MOMENTS, FLAG_TIMING
from spearmint_amd.synthetic import synthetic_problem
from oracle import gp_ei_oracle as orc


def rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def run(N, M, D, H, seed):
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, seed)
    eng = Engine(0)
    eng.set_observations(comp, vals); eng.set_candidates(cand); eng.set_hypers(hypers)
    t = time.time(); eng.factor(); tf = time.time() - t
    t = time.time(); eng.ei_run(FLAG_KEEP_MOMENTS); tr = time.time() - t
    print("== N=%d M=%d D=%d H=%d  factor %.3fs ei_run %.3fs" % (N, M, D, H, tf, tr))
    ei = eng.ei_draws(); bi, bv = eng.best(); em = eng.ei_mean()
    for h in range(min(H, 2)):
        st = {}
        e_ref = orc.compute_ei(comp, cand, vals, hypers[h], stages=st)
        K, L, a = eng.get_factor(h)
        Ks = eng.get_cross_cov(h, 0, min(M, 300))
        m, v = eng.get_moments(h)
        ok = np.isfinite(e_ref) & (e_ref > 1e-280)
        print(" draw %d: K %.2e  L %.2e  alpha %.2e  K* %.2e  m %.2e  v %.2e  EI(rel max) %.2e" % (
            h, rel(K, st["K"]), rel(L, st["L"]), rel(a, st["alpha"]), rel(Ks, st["Kstar"][:, :Ks.shape[1]]),
            rel(m, st["func_m"]), float(np.max(np.abs(v - st["func_v"]) / np.abs(st["func_v"]))),
            float(np.max(np.abs(ei[ok, h] - e_ref[ok]) / e_ref[ok])) if ok.any() else -1))
    ref = orc.ei_over_hypers(comp, cand, vals, hypers) if N * M * H < 3e8 else None
    if ref is not None:
        print(" argmax gpu %d ref %d ; mean bit-equal given gpu draws: %s" % (
            bi, orc.choose(ref), np.array_equal(em, np.mean(ei, axis=1))))
    eng.close()


if __name__ == "__main__":
    run(24, 300, 2, 3, 11)
    run(64, 500, 8, 4, 12)
    run(200, 700, 5, 3, 13)
    run(130, 257, 32, 2, 14)
    run(256, 20000, 8, 10, 2000)
    run(1000, 5000, 40, 3, 15)

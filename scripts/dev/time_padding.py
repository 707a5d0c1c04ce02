"""Dev tool: what skipping the padding of N buys (option gemm_partial; k_predict_gemm_tail + K(X*,X) without the pad rows).
EI step (factor + pass + argmax) wall time per call, 20 000 candidates x 10 draws, D = 8, padded vs skipped, same handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem

eng = Engine(0)
print("%6s %6s %12s %12s %8s   (ms per step; stage ms: predict_gemm, cov_cross)" % ("N", "Np", "padded", "skipped", "ratio"))
for N in (129, 144, 160, 192, 208, 224, 256, 257, 300, 350, 384, 400, 450, 520, 600, 700, 900, 1100, 1300, 1500, 2000):
    M, H, D = (20000, 10, 8) if N < 1024 else (100000, 10, 16)
    comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 11)
    eng.set_observations(comp, vals); eng.set_hypers(hypers); eng.set_candidates(cand)
    row = []
    for on in (0, 1):
        eng.set_option("gemm_partial", on)
        eng.ei_step(0); ref = eng.best()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(10):
                eng.ei_step(0)
            best = min(best, (time.perf_counter() - t) / 10 * 1e3)
        eng.set_option("timing", 1)
        eng.ei_step(0)
        tm = eng.timings()
        eng.set_option("timing", 0)
        row.append((best, ref, tm["predict_gemm"][0], tm["cov_cross"][0], eng.stat("last_step_skipped_padding")))
    assert row[0][1] == row[1][1]
    print("%6d %6d %12.4f %12.4f %8.3f   gemm %.3f -> %.3f  cov %.3f -> %.3f  skipped=%d" % (
        N, -(-N // 128) * 128, row[0][0], row[1][0], row[1][0] / row[0][0], row[0][2], row[1][2], row[0][3], row[1][3], row[1][4]))
eng.set_option("gemm_partial", -1)

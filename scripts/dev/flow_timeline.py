"""Dev tool: the timeline of the diagonal items of k_lean_flow (a library built with FLOW_STAMPS=1, see csrc/Makefile):
per block column the wall-clock time its item started, finished its history, started and finished its diagonal block.
   SPX_LIB=_variants/libspx_stamps.so python scripts/dev/flow_timeline.py H"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
H = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = Engine(0)
comp, cand, vals, hyp = synthetic_problem(N, 16, 32, H, 5)
eng.set_observations(comp, vals)
for _ in range(5):
    eng.set_hypers(hyp); eng.gp_logprob()
if len(sys.argv) > 3:
    eng.set_option("lean_flow_fast", int(sys.argv[3]))
    for _ in range(3):
        eng.set_hypers(hyp); eng.gp_logprob()
buf = np.zeros(32 * 64 * 8, dtype=np.int64)
eng._lib.spx_dev_flow_stamps(buf.ctypes.data_as(ctypes.c_void_p))
st = buf.reshape(32, 64, 8)[:H, :N // 64].astype(float) / 100.0     # 100 MHz wall clock -> us
t0 = st[:, 0, 0].min()
st -= t0
for h in (0, H - 1):
    print("draw %d: column | item start | history done | diag start | diag end | diag time | gap to previous diag end || relative to the previous diag end: "
          "last quarter asked | its rows in LDS | solve + product done | diag start | first 16 pivots done" % h)
    for c in range(N // 64):
        s = st[h, c]
        p = st[h, c - 1, 3] if c else 0.0
        print("  %2d  %8.1f %8.1f %8.1f %8.1f   %6.1f   %6.1f   || %6.2f %6.2f %6.2f %6.2f %6.2f" % (c, s[0], s[1], s[2], s[3], s[3] - s[2], s[2] - p,
              s[4] - p, s[5] - p, s[6] - p, s[2] - p, s[7] - p))
if hasattr(eng._lib, "spx_dev_flow_clk"):
    clk = np.zeros(32 * 64 * 2, dtype=np.int64)
    eng._lib.spx_dev_flow_clk(clk.ctypes.data_as(ctypes.c_void_p))
    clk = clk.reshape(32, 64, 2)[:H, :N // 64].astype(float)
    cyc = clk[:, :, 1] - clk[:, :, 0]
    us = st[:, :, 3] - st[:, :, 2]
    print("shader clock during the diagonal blocks: %.0f cycles per block on average = %.0f MHz (min %.0f, max %.0f)" % (
        cyc.mean(), (cyc / us).mean(), (cyc / us).min(), (cyc / us).max()))
print("mean diag time %.2f us, mean gap %.2f us, total %.1f us" % ((st[:, :, 3] - st[:, :, 2]).mean(), np.mean(st[:, 1:, 2] - st[:, :-1, 3]), st[:, -1, 3].max()))

"""Dev tool: where the work items of k_lean_flow spend their time (a library built from a copy of csrc/ with the accumulators
of this script's docstring added to flow_chunk -- see DESIGN.md section 10; not part of the shipped sources):
per batch size the mean item's life split into waiting for flags in its history, history steps, and what follows.
   SPX_LIB=_variants/libspx_acc.so python scripts/dev/flow_item_times.py"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
acc = (ctypes.c_ulonglong * 8)()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for H in (1, 4, 6, 8, 12, 20):
    comp, cand, vals, hyp = synthetic_problem(N, 16, 32, H, 5)
    eng.set_observations(comp, vals)
    for _ in range(3):
        eng.set_hypers(hyp); eng.gp_logprob()
    eng._lib.spx_dev_flow_acc(acc, 1)
    reps = 10
    t = time.perf_counter()
    for _ in range(reps):
        eng.set_hypers(hyp); eng.gp_logprob()
    wall = (time.perf_counter() - t) / reps
    eng._lib.spx_dev_flow_acc(acc, 1)
    a = [float(x) for x in acc]
    n = a[0] / reps
    us = lambda v: v / a[0] / 100.0
    print("N=%d H=%2d  call %.3f ms  items per call %5d | mean item life %6.1f us = history waits %6.1f + history steps %6.1f + rest %6.1f  | "
          "item-time summed %.2f ms-slots per call = %.0f slots busy on average"
          % (N, H, wall * 1e3, n, us(a[1]), us(a[2]), us(a[3]), us(a[4]), a[1] / reps / 1e5, a[1] / reps / 1e5 / (wall * 1e3)))

"""Dev: wall time of Engine.set_observations / set_candidates in a loop of next()-like call sequences (spikes?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
N, M, D, H = 256, 20000, 8, 10
for staged in (0, 1):
  eng.set_option("stage_copies", staged)
  print("=== stage_copies = %d (%s)" % (staged, "the runtime's own path for pageable host buffers" if not staged else "through the handle's page-locked staging buffer: the default"))
  ts, tl, te = [], [], []
  for rep in range(60):
      comp, cand, vals, hypers = synthetic_problem(N, M, D, H, 100 + rep)
      t = time.perf_counter(); eng.set_observations(comp, vals); ts.append(time.perf_counter() - t)
      for _ in range(20):
          eng.set_hypers(hypers); 
          t = time.perf_counter(); eng.gp_logprob(); tl.append(time.perf_counter() - t)
      t = time.perf_counter(); eng.ei_grid(comp, vals, cand, hypers, want_mean=True); te.append(time.perf_counter() - t)
  for name, a in (("set_observations", ts), ("gp_logprob", tl), ("ei_grid", te)):
      a = np.array(a) * 1e3
      print("%-18s n=%4d  median %.3f ms  p90 %.3f  max %.3f  mean %.3f   >1ms: %d" % (name, len(a), np.median(a), np.percentile(a, 90), a.max(), a.mean(), int((a > 1.0).sum() if name != "ei_grid" else (a > 3.0).sum())))
  print("set_observations sorted tail:", np.round(np.sort(np.array(ts) * 1e3)[-8:], 3))
  print("per iteration set_observations ms:", " ".join("%.2f" % (v * 1e3) for v in ts))
  print("per iteration ei_grid ms:", " ".join("%.2f" % (v * 1e3) for v in te))

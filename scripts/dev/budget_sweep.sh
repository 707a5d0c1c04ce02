#!/bin/bash
# dev: step time vs the K(X*,X) staging budget (draws per predict-GEMM launch)
cd $GRAFT_REPO_ROOT
for wl in c5 c3 c2; do
  for mb in 0 1024 2048 4096; do
    python bench.py --workload $wl --skip-extras --no-cpu-baseline --steps 5 --warmup 2 --kstar-budget-mb $mb 2>/dev/null | python -c "
import sys, json
o = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$wl budget_mb=$mb  ms/step %.3f  value %.4g  gemm frac %.3f (%d launches, %.3f ms)' % (o['ms_per_step'], o['value'], o['roofline']['frac'], o['roofline']['launches'], o['roofline']['avg_launch_ms']))"
  done
done

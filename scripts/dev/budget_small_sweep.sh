#!/bin/bash
# dev: step time vs a K(X*,X) staging budget SMALLER than the default 512 MB (does a chunk that fits the 256 MB MALL pay?)
cd $GRAFT_REPO_ROOT
for wl in ${WLS:-c3}; do
  for mb in ${MBS:-0 384 256 192 128 96 64}; do
    python bench.py --workload $wl --skip-extras --no-cpu-baseline --no-live-traffic --steps 5 --warmup 2 --kstar-budget-mb $mb 2>/dev/null | python -c "
import sys, json
o = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
st = o.get('stages_ms_per_step', {})
print('$wl budget_mb=$mb  ms/step %.3f  value %.4g  gemm frac %.3f (%d launches, %.3f ms)  cov %.2f ms' % (o['ms_per_step'], o['value'], o['roofline']['frac'], o['roofline']['launches'], o['roofline']['avg_launch_ms'], st.get('cov_cross', float('nan'))))"
  done
done

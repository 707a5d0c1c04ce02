#!/bin/bash
# Dev: kernel timeline of spx_gp_logprob calls at small N: per-kernel durations and the gaps between consecutive kernels.
#   bash scripts/dev/trace_small_lp.sh [N] [H]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-64}; H=${2:-6}
O=$R/gpurun_out/trace_small_lp; rm -rf $O; mkdir -p $O
cat > /tmp/lp_loop.py <<PY
import sys; sys.path.insert(0, "$R")
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng = Engine(0)
comp, cand, vals, hyp = synthetic_problem($N, 10, 8, $H, 5)
eng.set_observations(comp, vals)
for _ in range(60):
    eng.set_hypers(hyp); eng.gp_logprob()
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python /tmp/lp_loop.py > /dev/null 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
for f in glob.glob("$O/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
rows.sort()
tail = rows[-18:]
t0 = tail[0][0]; prev = None
for s, e, n in tail:
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, ((s - prev) / 1e3 if prev else 0.0), (e - s) / 1e3, n))
    prev = e
PY

#!/bin/bash
# Dev: kernel trace of the small-N EI step: per-kernel average and the gaps between consecutive kernels of one step.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-128}; M=${2:-20000}; D=${3:-8}; H=${4:-10}
O=$R/gpurun_out/trace_small; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/scripts/dev/small_n_loop.py $N $M $D $H 30 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
# last full step: find the last k_scale_rows<64>-like first kernel occurrences
names = [r[2] for r in rows]
# take the last 2 steps' worth: locate the last three occurrences of 'k_lean_flow'
idx = [i for i, n in enumerate(names) if n.startswith("k_lean_flow")]
a, b = idx[-3], idx[-2]
# a step starts at the scale_rows before k_lean_flow
while a > 0 and not names[a].startswith("k_lean_flow") or a == idx[-3]:
    a -= 1
    if names[a].startswith("k_scale_rows") and not names[a - 1].startswith("k_scale_rows"): break
t0 = rows[a][0]
prev_end = None
for s, e, n in rows[a:a + 24]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
    prev_end = e
PY

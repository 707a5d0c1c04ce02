#!/bin/bash
# Dev: memory-side counters of k_lean_flow at N = 2048 for a batch of H draws: L2 hits / misses, bytes fetched past the L2,
# against the operand bytes its products ask for.   bash scripts/dev/pmc_flow_mem.sh [H]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
H=${1:-8}
O=$R/gpurun_out/pmc_flow_mem; rm -rf $O; mkdir -p $O
CMD="python $R/scripts/lean_loop.py 2048 32 $H 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_lean_flow" --output-format csv -d $O/a -o a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_lean_flow" --output-format csv -d $O/b -o b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "k_lean_flow" --output-format csv -d $O/c -o c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-include-regex "k_lean_flow" --output-format csv -d $O/d -o d -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
H = $H
vals = {}
for tag in "abcd":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
            print("%-32s mean per launch %.4g   (%d launches)" % (k, vals[k], len(v)))
avg = None
for f in glob.glob("$O/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lean_flow" in r["Name"]:
            avg = float(r["AverageNs"]) * 1e-9
            print(r["Name"][:40], r["Calls"], "avg us %.1f" % (avg * 1e6))
nblk = 32
prod = sum(c * (nblk - c) for c in range(nblk)) + sum(range(nblk))     # tile products per draw (matrix + right-hand-side rows)
steps = prod / 2.0                                                       # a history step = two products on three operand tiles
opb = H * steps * 3 * 32768.0
print("H=%d: %d tile products per draw; operand bytes the history steps request (3 tiles per 2 products): %.3g; past the L2 (2 FETCH + WRITE, KiB): %.3g bytes" % (
    H, prod, opb, (2 * vals.get("FETCH_SIZE", 0) + vals.get("WRITE_SIZE", 0)) * 1024))
if avg:
    print("operand request rate %.2f TB/s; past the L2 %.2f TB/s; L2 hit rate %.3f" % (opb / avg / 1e12, (2 * vals.get("FETCH_SIZE", 0) + vals.get("WRITE_SIZE", 0)) * 1024 / avg / 1e12,
          vals.get("TCC_HIT_sum", 0) / max(1.0, vals.get("TCC_HIT_sum", 0) + vals.get("TCC_MISS_sum", 0))))
PY

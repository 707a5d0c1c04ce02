#!/bin/bash
# Dev: SQ counters of k_lean_flow at N = 2048 for a batch of H draws (default 12).   bash scripts/dev/pmc_flow.sh [H]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
H=${1:-12}
O=$R/gpurun_out/pmc_flow; rm -rf $O; mkdir -p $O
CMD="python $R/scripts/lean_loop.py 2048 32 $H 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-include-regex "k_lean_flow" --output-format csv -d $O/a -o a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-include-regex "k_lean_flow" --output-format csv -d $O/b -o b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --kernel-include-regex "k_lean_flow" --output-format csv -d $O/c -o c -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for tag in "abc":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("%-32s mean per launch %.4g   (%d launches)" % (k, sum(v) / len(v), len(v)))
for f in glob.glob("$O/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lean" in r["Name"] or "cov" in r["Name"]:
            print(r["Name"][:40], r["Calls"], "avg ns", r["AverageNs"])
PY

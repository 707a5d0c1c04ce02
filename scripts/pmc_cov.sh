#!/bin/bash
# dev tool: SQ counters of the K(X*,X) kernel (separate --pmc passes, kernel filter)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  d=$R/gpurun_out/pmc_cov_$(echo $set | cut -c1-12 | tr ' ' _)
  rocprofv3 --pmc $set --kernel-include-regex "k_cov" --output-format csv -d $d -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if not r["Kernel_Name"].startswith("void k_cov<0"): continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()): print("%-32s launches %4d  per-launch %.4g" % (k, n, v / n))
PY
done

#!/bin/bash
# round 6: SQ / TCC counters of the K(X*,X) kernel (k_cov_flat<8,0>) of a C3 step, the shipped library
# against the no-correlation-function ablation (_variants/libspx_covabl2.so: make COV_ABL=2, WRONG results, timing only).
# Separate --pmc passes (8 SQ slots), kernel filter, no trace domain beside them.   bash scripts/dev/pmc_cov_r06.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_cov_pmc.log
: > $OUT
for lib in shipped covabl2; do
  if [ $lib = covabl2 ]; then export SPX_LIB=$R/_variants/libspx_covabl2.so; else unset SPX_LIB; fi
  echo "=== $lib" >> $OUT
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
             "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
    d=$R/gpurun_out/pmc_cov_tmp
    rm -rf $d
    rocprofv3 --pmc $set --kernel-include-regex "k_cov" --output-format csv -d $d -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-extras --no-live-traffic > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -z "$f" ]; then echo "(pass failed: $set)" >> $OUT; continue; fi
    python - "$f" >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
names = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_cov" not in r["Kernel_Name"]: continue
    names[r["Kernel_Name"][:60]] += 1
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()): print("%-32s launches %4d  per-launch %.5g" % (k, n, v / n))
PY
    rm -rf $d
  done
done
unset SPX_LIB
echo "=== stage times (HIP events): shipped, then ablations" >> $OUT
python $R/scripts/dev/time_cov_abl.py $R/_variants/libspx_covabl2.so 2>/dev/null | grep -v amdgpu.ids >> $OUT
cat $OUT

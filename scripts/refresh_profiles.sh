#!/bin/bash
# Regenerate the evidence under profiles/ on the GPU box (run through gpurun; raw outputs under gpurun_out/refresh,
# the small summaries are written to profiles/ by scripts/pmc_summary.py and copied back with gpurun_out/).
#   gpurun -- 'bash scripts/refresh_profiles.sh r03 c3'      (second argument: bench.py workload, default c3)
set -u
TAG=${1:-r03}
WL=${2:-c3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh_$WL
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --skip-extras --no-live-traffic --workload $WL"
# kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 2 --warmup 1 > $O/stats.log 2>&1
# HBM counters, separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > /dev/null 2>&1
# SQ / TCC counters of the dominant kernels (separate passes: 8 SQ slots, 4 TCC slots)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-include-regex "k_predict_gemm|k_cov|k_chol|k_trinv" --output-format csv -d $O/sq1 -- $B --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --kernel-include-regex "k_predict_gemm|k_cov|k_chol|k_trinv" --output-format csv -d $O/sq2 -- $B --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_predict_gemm|k_cov|k_chol|k_trinv" --output-format csv -d $O/tcc -- $B --steps 1 --warmup 0 > /dev/null 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for n in fetch write sq1 sq2 tcc; do cp $(find $O/$n -name "*counter_collection.csv" | head -1) $O/$n.csv 2>/dev/null; done
rm -rf $O/stats $O/fetch $O/write $O/sq1 $O/sq2 $O/tcc
cd $R
mkdir -p profiles
python scripts/pmc_summary.py ${TAG}_${WL} $O/kernel_stats.csv $O/fetch.csv $O/write.csv $O/sq1.csv $O/sq2.csv $O/tcc.csv
cp profiles/${TAG}_${WL}_rocprof_summary.json profiles/${TAG}_${WL}_sq_pmc.json $O/ 2>/dev/null
cp $O/kernel_stats.csv $O/${TAG}_${WL}_kernel_stats.csv
head -c 400 profiles/${TAG}_${WL}_sq_pmc.json; echo

#!/bin/bash
# Regenerate the evidence under profiles/ on the GPU box (run through gpurun; outputs under gpurun_out/refresh,
# then `python scripts/pmc_summary.py r01_c3 ...` and copies are done on the build side).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
# HBM counters, separate passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/fetch -name "*counter_collection.csv" | head -1) $O/fetch.csv
cp $(find $O/write -name "*counter_collection.csv" | head -1) $O/write.csv
rm -rf $O/stats $O/fetch $O/write
cd $R
mkdir -p profiles
python scripts/pmc_summary.py r01_c3 $O/kernel_stats.csv $O/fetch.csv $O/write.csv
cp profiles/r01_c3_rocprof_summary.json $O/
# bench lines (the c3 line reads the summary written above for roofline.traffic)
python bench.py 2>/dev/null | tail -1 > $O/r01_c3_bench_line.json
for w in c2 c4 c5; do python bench.py --workload $w 2>/dev/null | tail -1 > $O/r01_${w}_bench_line.json; done
head -c 600 $O/r01_c3_bench_line.json; echo

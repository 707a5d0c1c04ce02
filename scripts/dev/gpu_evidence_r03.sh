#!/bin/bash
# round-3 evidence: bench lines, rocprof summaries, next() profile, fuzz sweep (raw outputs under gpurun_out/evidence)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; rm -rf $O; mkdir -p $O
echo "== bench c3 (default)"; timeout 600 python bench.py > $O/r03_c3_bench_line.json 2> $O/bench_c3.err; tail -c 600 $O/r03_c3_bench_line.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --skip-extras > $O/r03_c2_bench_line.json 2>> $O/bench_c2.err
echo "== bench c5"; timeout 300 python bench.py --workload c5 --skip-extras > $O/r03_c5_bench_line.json 2>> $O/bench_c5.err
echo "== time_lean"; timeout 300 python scripts/time_lean.py > $O/r03_time_lean.log 2>&1; cat $O/r03_time_lean.log
echo "== next()"; timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | head -14 | tee $O/r03_next_profile.log
echo "== next() kernel stats"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nextprof -- python $GRAFT_REPO_ROOT/scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" > $O/nextprof.log 2>&1)
cp $(find $O/nextprof -name "*kernel_stats.csv" | head -1) $O/r03_next_kernel_stats.csv; rm -rf $O/nextprof; head -8 $O/r03_next_kernel_stats.csv | cut -c1-160
echo "== profiles c3"; timeout 900 bash scripts/refresh_profiles.sh r03 c3 2>&1 | tail -2
echo "== fuzz"; timeout 900 python scripts/fuzz_parity.py 120 3031 mix > $O/r03_fuzz_parity.log 2>&1; tail -3 $O/r03_fuzz_parity.log

#!/bin/bash
# round-3 evidence (raw outputs under gpurun_out/evidence; copied into profiles/ afterwards).   bash scripts/dev/gpu_evidence_r03.sh a|b
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; mkdir -p $O
if [ "$1" = a ]; then
echo "== bench c3 (default)"; timeout 600 python bench.py > $O/r03_c3_bench_line.json 2> $O/bench_c3.err; tail -c 600 $O/r03_c3_bench_line.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --skip-extras > $O/r03_c2_bench_line.json 2>> $O/bench_c2.err
echo "== bench c5"; timeout 300 python bench.py --workload c5 --skip-extras > $O/r03_c5_bench_line.json 2>> $O/bench_c5.err
echo "== time_lean"; timeout 300 python scripts/time_lean.py > $O/r03_time_lean.log 2>&1; cat $O/r03_time_lean.log
echo "== flow A/B"; timeout 300 python scripts/dev/flow_ab.py 2>&1 | grep -v Warn > $O/r03_flow_ab.log; head -8 $O/r03_flow_ab.log
echo "== flow modes"; timeout 300 python scripts/dev/flow_modes.py > $O/r03_flow_modes.log 2>&1; head -8 $O/r03_flow_modes.log
echo "== stress"; (timeout 200 python scripts/dev/ps_stress.py 600 flow; timeout 200 python scripts/dev/ps_stress.py 300 ps) 2>&1 | tail -2 > $O/r03_flow_stress.log; cat $O/r03_flow_stress.log
echo "== next()"; timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | head -14 > $O/r03_next_profile.log
timeout 300 python scripts/dev/next_hist.py 2>&1 | tail -4 >> $O/r03_next_profile.log; cat $O/r03_next_profile.log
echo "== next() kernel stats"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nextprof -- python $GRAFT_REPO_ROOT/scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" > $O/nextprof.log 2>&1)
cp $(find $O/nextprof -name "*kernel_stats.csv" | head -1) $O/r03_next_kernel_stats.csv; rm -rf $O/nextprof; head -8 $O/r03_next_kernel_stats.csv | cut -c1-160
echo "== flow pmc"; timeout 300 bash scripts/dev/pmc_flow.sh 12 > $O/r03_flow_pmc_h12.log 2>&1; timeout 300 bash scripts/dev/pmc_flow.sh 1 > $O/r03_flow_pmc_h1.log 2>&1; tail -5 $O/r03_flow_pmc_h1.log
else
echo "== profiles c3"; timeout 900 bash scripts/refresh_profiles.sh r03 c3 2>&1 | tail -2
echo "== profiles c2"; timeout 600 bash scripts/refresh_profiles.sh r03 c2 2>&1 | tail -2
echo "== profiles c5"; timeout 600 bash scripts/refresh_profiles.sh r03 c5 2>&1 | tail -2
echo "== fuzz"; timeout 900 python scripts/fuzz_parity.py 120 3031 mix > $O/r03_fuzz_parity.log 2>&1; tail -3 $O/r03_fuzz_parity.log
fi

#!/bin/bash
# round-6 evidence (raw outputs under gpurun_out/evidence; copied into profiles/ afterwards).   bash scripts/dev/gpu_evidence_r06.sh a|b
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; mkdir -p $O
A="mcmc_iters=10,burnin=10,grid_subset=20"
if [ "$1" = a ]; then
echo "== bench c3 (the driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_c3_bench_line.json 2> $O/bench_c3.err; tail -c 700 $O/r06_c3_bench_line.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --skip-extras --steps 20 --warmup 3 > $O/r06_c2_bench_line.json 2>> $O/bench_c2.err
echo "== bench c5"; timeout 300 python bench.py --workload c5 --skip-extras --steps 10 --warmup 2 > $O/r06_c5_bench_line.json 2>> $O/bench_c5.err
echo "== next() vs the reference"; timeout 400 python bench.py --next-baseline > $O/r06_next_vs_reference.json 2>/dev/null; tail -c 300 $O/r06_next_vs_reference.json; echo
echo "== next() phases"; { for shape in "256 20000 8" "64 20000 8" "1024 20000 16"; do for v in "sampler=python,lookahead=6,follow=0:0" "sampler=native,lookahead=6,follow=0:0" "sampler=native"; do echo "=== $shape $v"; timeout 300 python scripts/dev/next_phases.py $shape "$A,$v" 2>&1 | grep -v amdgpu.ids; done; done; echo "=== C3 size"; for v in "sampler=python,lookahead=6,follow=0:0" "sampler=native"; do timeout 600 python scripts/dev/next_phases.py 2048 200000 32 "mcmc_iters=20,burnin=2,grid_subset=20,$v" 2>&1 | grep -v amdgpu.ids; done; } > $O/r06_next_phases.log 2>&1; grep "^===\|^next()" $O/r06_next_phases.log
echo "== log-likelihood call forms"; timeout 400 python scripts/dev/lean_one_ab.py 2>&1 | grep -v amdgpu.ids > $O/r06_lean_one_ab.log; tail -4 $O/r06_lean_one_ab.log
echo "== rows"; timeout 300 python scripts/dev/time_lean_rows.py 32:4 64:8 128:8 256:8 512:8 1024:16 2048:32 2>&1 | grep -v amdgpu.ids > $O/r06_lean_rows.log; cat $O/r06_lean_rows.log
echo "== time_lean"; timeout 300 python scripts/time_lean.py 2>&1 | grep -v amdgpu.ids > $O/r06_time_lean.log; tail -4 $O/r06_time_lean.log
echo "== next() profile N=256"; { timeout 300 python scripts/profile_next.py 256 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -40; echo; echo "--- by own time"; SPX_PROF_SORT=tottime timeout 300 python scripts/profile_next.py 256 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -30; } > $O/r06_next_profile_n256.log
{ timeout 300 python scripts/profile_next.py 64 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -40; } > $O/r06_next_profile_n64.log
cd /tmp && export TMPDIR=/tmp
for n in 256 64; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n$n -o n$n -- python $GRAFT_REPO_ROOT/scripts/profile_next.py $n 20000 8 "" "$A" > /dev/null 2>&1
f=$(find $O/prof_n$n -name "*kernel_stats.csv" | head -1); { echo; echo "--- rocprofv3 --kernel-trace --stats of the same command (cold process: includes the first call's warm-up)"; head -16 $f; } >> $O/r06_next_profile_n$n.log; rm -rf $O/prof_n$n
done
cd $GRAFT_REPO_ROOT
echo "== stress"; (timeout 200 python scripts/dev/ps_stress.py 400 flow) 2>&1 | tail -2 > $O/r06_flow_stress.log; cat $O/r06_flow_stress.log
else
echo "== profiles c3"; timeout 900 bash scripts/refresh_profiles.sh r06 c3 2>&1 | tail -2
echo "== profiles c2"; timeout 600 bash scripts/refresh_profiles.sh r06 c2 2>&1 | tail -2
echo "== fuzz"; timeout 900 python scripts/fuzz_parity.py 120 6262 mix > $O/r06_fuzz_parity.log 2>&1; tail -3 $O/r06_fuzz_parity.log
fi

#!/bin/bash
# round-5 evidence (raw outputs under gpurun_out/evidence; copied into profiles/ afterwards).   bash scripts/dev/gpu_evidence_r05.sh a|b
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; mkdir -p $O
if [ "$1" = a ]; then
echo "== bench c3 (the driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_c3_bench_line.json 2> $O/bench_c3.err; tail -c 600 $O/r05_c3_bench_line.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --skip-extras --steps 20 --warmup 3 > $O/r05_c2_bench_line.json 2>> $O/bench_c2.err
echo "== bench c5"; timeout 300 python bench.py --workload c5 --skip-extras --steps 10 --warmup 2 > $O/r05_c5_bench_line.json 2>> $O/bench_c5.err
echo "== bench --gpus 2, no launcher, both ranks on this GPU (plumbing, not a measurement)"
SPX_BENCH_BACKEND=gloo SPX_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --c4-candidates 200000 --c5-candidates 100000 > $O/r05_two_ranks_one_gpu_line.json 2> $O/bench_2r.err; tail -c 300 $O/r05_two_ranks_one_gpu_line.json; echo
echo "== padding skip"; timeout 300 python scripts/dev/time_padding.py 2>&1 | grep -v amdgpu.ids > $O/r05_padding_skip.log; tail -5 $O/r05_padding_skip.log
echo "== time_lean"; timeout 300 python scripts/time_lean.py 2>&1 | grep -v amdgpu.ids > $O/r05_time_lean.log; tail -8 $O/r05_time_lean.log
echo "== next()"; timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | grep -v amdgpu.ids | head -14 > $O/r05_next_profile.log
timeout 300 python scripts/dev/next_hist.py 2>&1 | tail -4 >> $O/r05_next_profile.log; tail -6 $O/r05_next_profile.log
echo "== next() vs the reference"; timeout 300 python bench.py --next-baseline > $O/r05_next_vs_reference.json 2>/dev/null; tail -c 400 $O/r05_next_vs_reference.json; echo
echo "== stress"; (timeout 200 python scripts/dev/ps_stress.py 400 flow) 2>&1 | tail -2 > $O/r05_flow_stress.log; cat $O/r05_flow_stress.log
else
echo "== profiles c3"; timeout 900 bash scripts/refresh_profiles.sh r05 c3 2>&1 | tail -2
echo "== profiles c2"; timeout 600 bash scripts/refresh_profiles.sh r05 c2 2>&1 | tail -2
echo "== fuzz"; timeout 900 python scripts/fuzz_parity.py 120 6161 mix > $O/r05_fuzz_parity.log 2>&1; tail -3 $O/r05_fuzz_parity.log
fi

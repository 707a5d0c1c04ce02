#!/bin/bash
# round-4 evidence (raw outputs under gpurun_out/evidence; copied into profiles/ afterwards).   bash scripts/dev/gpu_evidence_r04.sh a|b
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; mkdir -p $O
if [ "$1" = a ]; then
echo "== bench c3 (default)"; timeout 900 python bench.py > $O/r04_c3_bench_line.json 2> $O/bench_c3.err; tail -c 400 $O/r04_c3_bench_line.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --skip-extras --steps 20 --warmup 3 > $O/r04_c2_bench_line.json 2>> $O/bench_c2.err
echo "== bench c5"; timeout 300 python bench.py --workload c5 --skip-extras --steps 10 --warmup 2 > $O/r04_c5_bench_line.json 2>> $O/bench_c5.err
echo "== time_lean"; timeout 300 python scripts/time_lean.py 2>&1 | grep -v amdgpu.ids > $O/r04_time_lean.log; cat $O/r04_time_lean.log
echo "== small n"; timeout 300 python scripts/dev/time_small_n.py 2>&1 | grep -v amdgpu.ids > $O/r04_small_n.log; cut -c1-200 $O/r04_small_n.log
echo "== small lp"; timeout 300 python scripts/dev/time_small_lp.py 2>&1 | grep -v amdgpu.ids > $O/r04_small_lp.log; cat $O/r04_small_lp.log
echo "== step overlap"; timeout 300 python scripts/dev/step_overlap_ab.py 2>&1 | grep -v amdgpu.ids > $O/r04_step_overlap_ab.log; cat $O/r04_step_overlap_ab.log
echo "== stress"; (timeout 200 python scripts/dev/ps_stress.py 600 flow; timeout 200 python scripts/dev/ps_stress.py 300 ps) 2>&1 | tail -2 > $O/r04_flow_stress.log; cat $O/r04_flow_stress.log
echo "== next()"; timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | grep -v amdgpu.ids | head -14 > $O/r04_next_profile.log
timeout 300 python scripts/dev/next_hist.py 2>&1 | tail -4 >> $O/r04_next_profile.log; cat $O/r04_next_profile.log
echo "== flow pmc"; timeout 300 bash scripts/dev/pmc_flow.sh 12 2>&1 | grep -v amdgpu.ids > $O/r04_flow_pmc_h12.log; timeout 300 bash scripts/dev/pmc_flow.sh 1 2>&1 | grep -v amdgpu.ids > $O/r04_flow_pmc_h1.log; tail -5 $O/r04_flow_pmc_h1.log
echo "== leak probe"; timeout 200 python scripts/dev/leak_probe.py 6 2>&1 | grep -v amdgpu.ids > $O/r04_leak_probe_final.log; tail -3 $O/r04_leak_probe_final.log
else
echo "== profiles c3"; timeout 900 bash scripts/refresh_profiles.sh r04 c3 2>&1 | tail -2
echo "== profiles c2"; timeout 600 bash scripts/refresh_profiles.sh r04 c2 2>&1 | tail -2
echo "== profiles c5"; timeout 600 bash scripts/refresh_profiles.sh r04 c5 2>&1 | tail -2
echo "== fuzz"; timeout 900 python scripts/fuzz_parity.py 120 4041 mix > $O/r04_fuzz_parity.log 2>&1; tail -3 $O/r04_fuzz_parity.log
fi

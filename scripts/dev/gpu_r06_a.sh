#!/bin/bash
# round-6 first look: the new main-loop test on libspx, where a next() at Spearmint's operating sizes spends its time, what a
# log-likelihood call costs by rows at small N.   bash scripts/dev/gpu_r06_a.sh
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r06a; mkdir -p $O
echo "== main loop on libspx"; timeout 600 python -m pytest tests/test_gpu_i_main_loop.py -q -x 2>&1 | tail -5 | tee $O/main_loop_gpu.log
echo "== rows"; timeout 300 python scripts/dev/time_lean_rows.py 32:4 64:8 128:8 256:8 512:8 1024:16 2>&1 | grep -v amdgpu.ids | tee $O/lean_rows.log
ARGS="mcmc_iters=10,burnin=10,grid_subset=20"
for la in 6 12 20; do
echo "== hist N=256 lookahead=$la"; timeout 300 python scripts/dev/batch_hist.py 256 20000 8 "$ARGS,lookahead=$la" 2>&1 | grep -v amdgpu.ids | tee $O/hist_n256_la$la.log | tail -25
done
echo "== hist N=64"; timeout 300 python scripts/dev/batch_hist.py 64 20000 8 "$ARGS" 2>&1 | grep -v amdgpu.ids | tee $O/hist_n64.log | tail -14
echo "== cProfile N=256"; timeout 300 python scripts/profile_next.py 256 20000 8 "" "$ARGS" 2>&1 | grep -v amdgpu.ids | head -45 | tee $O/next_profile_n256.log
echo "== cProfile N=256 tottime"; SPX_PROF_SORT=tottime timeout 300 python scripts/profile_next.py 256 20000 8 "" "$ARGS" 2>&1 | grep -v amdgpu.ids | head -40 | tee $O/next_profile_n256_tot.log
cd /tmp && export TMPDIR=/tmp
echo "== rocprof N=256"; timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_n256 -o n256 -- python $GRAFT_REPO_ROOT/scripts/profile_next.py 256 20000 8 "" "$ARGS" > $O/rocprof_n256.out 2>&1; ls $O/prof_n256 | head; 
find $O/prof_n256 -name '*kernel_stats.csv' | head -1 | xargs -I{} head -25 {}

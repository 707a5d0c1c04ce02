cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/evidence; mkdir -p $O
A="mcmc_iters=10,burnin=10,grid_subset=20"
bash scripts/gate.sh r06f
echo "== next() profiles"
{ timeout 300 python scripts/profile_next.py 256 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -40; echo; echo "--- by own time"; SPX_PROF_SORT=tottime timeout 300 python scripts/profile_next.py 256 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -30; } > $O/r06_next_profile_n256.log
{ timeout 300 python scripts/profile_next.py 64 20000 8 "" "$A" 2>&1 | grep -v amdgpu.ids | head -40; } > $O/r06_next_profile_n64.log
cd /tmp && export TMPDIR=/tmp
for n in 256 64; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n$n -o n$n -- python $GRAFT_REPO_ROOT/scripts/profile_next.py $n 20000 8 "" "$A" > /dev/null 2>&1
f=$(find $O/prof_n$n -name "*kernel_stats.csv" | head -1); { echo; echo "--- rocprofv3 --kernel-trace --stats of the same command (two next() calls: the warm-up and the profiled one)"; head -16 $f; } >> $O/r06_next_profile_n$n.log; rm -rf $O/prof_n$n
done
head -12 $O/r06_next_profile_n256.log; tail -12 $O/r06_next_profile_n256.log

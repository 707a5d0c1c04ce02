#!/bin/bash
# round-6: speculation depth sweep (lookahead x follow) at Spearmint's operating sizes.   bash scripts/dev/gpu_r06_b.sh
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r06b; mkdir -p $O
ARGS="mcmc_iters=10,burnin=10,grid_subset=20"
for shape in "256 20000 8" "64 20000 8" "1024 20000 16"; do
for la in 6 8; do for fo in 0:0 3:1 4:2 6:3; do
echo "== $shape lookahead=$la follow=$fo"; timeout 300 python scripts/dev/batch_hist.py $shape "$ARGS,lookahead=$la,follow=$fo" 2>&1 | grep -v amdgpu.ids | grep "next()\|rows evaluated" | tail -3
done; done
echo "== $shape auto"; timeout 300 python scripts/dev/batch_hist.py $shape "$ARGS" 2>&1 | grep -v amdgpu.ids | tail -22
done 2>&1 | tee $O/depth_sweep.log

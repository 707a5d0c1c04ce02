#!/bin/bash
# round-6: native sampler + lock-step refinement: phases of a warm next(), python vs native, depth variants
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r06c; mkdir -p $O
A="mcmc_iters=10,burnin=10,grid_subset=20"
{
for shape in "256 20000 8" "64 20000 8" "1024 20000 16"; do
for v in "sampler=python,lookahead=6,follow=0:0" "sampler=native,lookahead=6,follow=0:0" "sampler=native,lookahead=8,follow=4:2" "sampler=native,lookahead=8,follow=6:3" "sampler=native"; do
echo "=== $shape $v"; timeout 300 python scripts/dev/next_phases.py $shape "$A,$v" 2>&1 | grep -v amdgpu.ids
done; done
echo "=== C3 size"; for v in "sampler=python,lookahead=6,follow=0:0" "sampler=native"; do timeout 600 python scripts/dev/next_phases.py 2048 200000 32 "mcmc_iters=20,burnin=2,grid_subset=20,$v" 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee $O/next_phases.log
echo "=== gpu tests (a_parity, h, i, f)"; timeout 1200 python -m pytest tests/test_gpu_a_parity.py tests/test_gpu_h_reference_patch.py tests/test_gpu_i_main_loop.py tests/test_gpu_f_lite_loop.py -q -m gpu 2>&1 | tail -8 | tee $O/pytest_subset.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call2; mkdir -p $O
echo "== pytest logprob"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logprob or loglikelihood or sampler" 2>&1 | tail -8 | tee $O/pytest.log
echo "== time_lean fused"; timeout 300 python scripts/time_lean.py 2>&1 | tee $O/time_lean_fused.log
echo "== time_lean two-launch"; SPX_LEAN_FUSED=0 timeout 300 python scripts/time_lean.py 2>&1 | tee $O/time_lean_two.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call3; mkdir -p $O
echo "== ubench_diag"; timeout 60 scripts/ubench_diag 2>&1 | tee $O/ubench_diag.log
echo "== pytest logprob"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logprob or loglikelihood or sampler or stage or golden_ei" 2>&1 | tail -8 | tee $O/pytest.log
echo "== time_lean default"; timeout 300 python scripts/time_lean.py 2>&1 | tee $O/time_lean.log

#!/bin/bash
# round-3 GPU call 1: new register-resident diagonal block vs the round-2 library
cd $GRAFT_REPO_ROOT
O=gpurun_out/call1; mkdir -p $O
echo "== ubench_diag"; timeout 60 scripts/ubench_diag 2>&1 | tee $O/ubench_diag.log
echo "== time_lean new"; timeout 300 python scripts/time_lean.py 2>&1 | tee $O/time_lean_new.log
echo "== time_lean r02"; SPX_LIB=$PWD/build_abl/libspx_r02.so timeout 300 python scripts/time_lean.py 2>&1 | tee $O/time_lean_r02.log
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
echo "== next() new"; timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | head -12 | tee $O/next_new.log
echo "== next() r02"; SPX_LIB=$PWD/build_abl/libspx_r02.so timeout 300 python scripts/profile_next.py 2048 200000 32 "" "mcmc_iters=20,grid_subset=20" 2>&1 | head -12 | tee $O/next_r02.log
echo "== profiles c5"; timeout 600 bash scripts/refresh_profiles.sh r03 c5 2>&1 | tail -3
echo "== profiles c2"; timeout 600 bash scripts/refresh_profiles.sh r03 c2 2>&1 | tail -3

#!/bin/bash
# The round's gate (run ON the GPU box through gpurun, after the LAST change of libspx.so):
#     gpurun --timeout 900 -- 'bash scripts/gate.sh r04'
# rebuilds nothing (the library travels with the snapshot), runs the FULL `pytest -m gpu` without -x and writes
# gpurun_out/<tag>_pytest_gpu.log headed by sha256(libspx.so); copy it to profiles/ and commit it.  scripts/gate_check.sh
# compares the hash in the committed log with the library in the tree.
tag=${1:-r04}
mkdir -p gpurun_out
out=gpurun_out/${tag}_pytest_gpu.log
{
  echo "libspx.so sha256 $(sha256sum spearmint_amd/libspx.so | cut -d' ' -f1)"
  echo "sources sha256 $(cat spearmint_amd/csrc/*.hip spearmint_amd/csrc/*.h include/spx.h | sha256sum | cut -d' ' -f1)"
  echo "date $(date -u +%FT%TZ)  host $(hostname)  $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 gfx9 | tr -s ' ')"
} > $out
SPX_FULL_C3=1 python -m pytest tests -m gpu -q -p no:cacheprovider -W ignore -rfEs --durations=8 2>&1 | grep -v "amdgpu.ids" >> $out
rc=${PIPESTATUS[0]}
echo "pytest rc $rc" >> $out
tail -25 $out
exit $rc

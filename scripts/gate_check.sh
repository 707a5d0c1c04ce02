#!/bin/bash
# Does the committed gate log belong to the sources in the tree?  (the .so itself is rebuilt per container: compare sources)
log=${1:-profiles/r04_pytest_gpu.log}
want=$(grep -m1 '^sources sha256' $log | cut -d' ' -f3)
have=$(cat spearmint_amd/csrc/*.hip spearmint_amd/csrc/*.h include/spx.h | sha256sum | cut -d' ' -f1)
if [ "$want" = "$have" ]; then echo "gate log matches the sources ($have)"; grep -E "passed|failed|pytest rc" $log | tail -2; else echo "STALE gate log: $want vs $have"; exit 1; fi

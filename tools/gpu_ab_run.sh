#!/bin/bash
# run tools/acc_bench.py (or $AB_CMD) under every variant library named on the command line (build them with tools/build_variant.sh)
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo -n "$n: "; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python ${AB_CMD:-tools/acc_bench.py} 2>&1 | grep -v "^block" | tail -1
done

#!/bin/bash
# accumulate consumer variant libraries (tools/build_variant.sh): timing, and parity for those named with a trailing '+'
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  n=${a%+}
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  export POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  echo "== $n"
  [ "$a" != "$n" ] && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -W ignore -x -k "estep or accumulate or em_ or outlier" 2>&1 | tail -2
  timeout 300 python tools/acc_bench.py 2>&1 | tail -1
  timeout 300 python tools/estep_peaked_bench.py 2>&1 | tail -2 | head -1 | cut -c1-200
done

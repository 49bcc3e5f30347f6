#!/bin/bash
# accumulate consumer: fragment carry fix (no v_perm), cf folded into the chain + loop unrolled by two
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -W ignore -x -k "estep or accumulate or em_ or outlier or c4" 2>&1 | tail -5 | tee $O/tests.log
for n in acc_noperm default; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo -n "$n: "; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/acc_bench.py 2>&1 | tail -1
  echo -n "$n peaked: "; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/estep_peaked_bench.py 2>&1 | tail -2 | cut -c1-300
done

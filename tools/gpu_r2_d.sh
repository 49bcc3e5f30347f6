#!/bin/bash
# quick A/B of the accumulate pass + its parity tests
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -W ignore -x -k "estep or em_ or c4 or c2 or ill_cond or mstep" 2>&1 | tail -8 > $O/tests.log
timeout 300 python3 tools/acc_bench.py > $O/acc_f16.log 2>&1
timeout 600 python3 tools/estep_peaked_bench.py > $O/peaked_f16.log 2>&1
cat $O/tests.log $O/acc_f16.log $O/peaked_f16.log

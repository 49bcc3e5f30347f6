#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -W ignore -x -k "estep or em_ or c4 or c2 or ill_cond or mstep" 2>&1 | tail -4
echo -n "overlap: "; timeout 300 python3 tools/acc_bench.py | tail -1
echo -n "serial : "; PCL_ACC_OVERLAP=0 timeout 300 python3 tools/acc_bench.py | tail -1
echo -n "bf16 r1: "; PCL_ACC_BF16=1 timeout 300 python3 tools/acc_bench.py | tail -1
timeout 600 python3 tools/estep_peaked_bench.py

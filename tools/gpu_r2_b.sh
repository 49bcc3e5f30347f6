#!/bin/bash
# round 2, call B: the f16 producer/consumer accumulate kernel: parity tests, A/B timing against the round-1 kernel
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -W ignore -k "estep or em_ or accum or c4 or c2 or ill_cond or mstep or exchange or hmm" 2>&1 | tail -30 > $O/tests.log
timeout 300 python tools/acc_bench.py > $O/acc_f16.log 2>&1
PCL_ACC_BF16=1 timeout 300 python tools/acc_bench.py > $O/acc_bf16.log 2>&1
timeout 600 python tools/estep_peaked_bench.py > $O/peaked_f16.log 2>&1
tail -15 $O/tests.log; cat $O/acc_f16.log $O/acc_bf16.log $O/peaked_f16.log

#!/bin/bash
mkdir -p gpurun_out
PCL_SCORE_VARIANT=5 python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -15 > gpurun_out/v5_tests.log
python tools/accuracy_stress.py 5 > gpurun_out/v5_stress.log 2>&1
PCL_SCORE_VARIANT=5 CHECK=1 python tools/score_bench.py 1024 2048 1000 > gpurun_out/v5_bench.log 2>&1
cat gpurun_out/v5_tests.log gpurun_out/v5_stress.log; tail -3 gpurun_out/v5_bench.log

#!/bin/bash
mkdir -p gpurun_out
PCL_SCORE_VARIANT=6 python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -12 > gpurun_out/v6_tests.log
python tools/accuracy_stress.py 6 > gpurun_out/v6_stress.log 2>&1
PCL_SCORE_VARIANT=6 CHECK=1 python tools/score_bench.py 1024 2048 1000 > gpurun_out/v6_bench.log 2>&1
PCL_SCORE_VARIANT=5 CHECK=1 python tools/score_bench.py 1024 2048 1000 >> gpurun_out/v6_bench.log 2>&1
cat gpurun_out/v6_tests.log; tail -5 gpurun_out/v6_stress.log; grep -E "ms/launch|max abs" gpurun_out/v6_bench.log

#!/bin/bash
# conditioning guard: GPU tests, accuracy stress, short bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -15 > gpurun_out/cond_tests.log
python tools/accuracy_stress.py > gpurun_out/cond_stress.log 2>&1
python bench.py --steps 5 --warmup 2 --cpu-baseline 0 > gpurun_out/cond_bench.log 2>&1
cat gpurun_out/cond_tests.log gpurun_out/cond_stress.log; tail -c 1500 gpurun_out/cond_bench.log

#!/bin/bash
mkdir -p gpurun_out
PCL_SCORE_VARIANT=7 python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -12 > gpurun_out/v7_tests.log
sed -i 's/for v in (5, 4, 3, 1):/for v in (7, 5, 4, 3, 1):/' tools/accuracy_stress.py
python tools/accuracy_stress.py 7 > gpurun_out/v7_stress.log 2>&1
for v in 7 5 7 5; do PCL_SCORE_VARIANT=$v CHECK=1 python tools/score_bench.py 1024 2048 1000 2>&1 | tail -2 >> gpurun_out/v7_bench.log; done
cat gpurun_out/v7_tests.log; tail -5 gpurun_out/v7_stress.log; grep -E "ms/launch|max abs" gpurun_out/v7_bench.log

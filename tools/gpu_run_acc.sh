#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -25 > gpurun_out/tests.log
cat gpurun_out/tests.log
python bench.py --steps 3 --warmup 1 --cpu-baseline 0 2>gpurun_out/bench_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(d['extra'])"
tail -3 gpurun_out/bench_err.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_units.py -m gpu -q -W ignore -x 2>&1 | tail -3
timeout 900 python bench.py --cpu-baseline 0 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r2o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])
e=d['extra']
for k in ('estep_ms','accumulate_ms','accumulate_pruned'): print(k, e.get(k))
P

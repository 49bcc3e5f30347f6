#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -15 > $O/tests.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -6 $O/tests.log; tail -3 $O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['traffic'])
e=d['extra']
for k in ('estep_ms','accumulate_ms','strict_f32','estep_peaked','clock_power_under_scoring'): print(k, e.get(k))
print(d.get('roofline_estep'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P

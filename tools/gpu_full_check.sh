#!/bin/bash
# full -m gpu suite, smoke(), default bench line; results under gpurun_out/full
cd $GRAFT_REPO_ROOT
O=gpurun_out/full; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -25 > $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -8 $O/tests.log; tail -2 $O/smoke.log; tail -3 $O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['traffic'])
e=d['extra']
for k in sorted(e): print(k, e[k])
print(d.get('roofline_estep'))
print(d['cpu_baseline'])
P

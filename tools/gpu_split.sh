#!/bin/bash
# split states: the tests around them, what an EM iteration costs now, config 4 whole
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/split
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_accumulate.py tests/test_gpu_units.py -m gpu -q -W ignore -x > gpurun_out/split/t.log 2>&1; rc=$?
tail -8 gpurun_out/split/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/em_iter_probe.py > gpurun_out/split/em.log 2>&1 || { tail -5 gpurun_out/split/em.log; exit 1; }
cat gpurun_out/split/em.log
timeout -k 10 600 python bench.py --workload C4 > gpurun_out/split/c4.json 2> gpurun_out/split/c4.err || { tail -5 gpurun_out/split/c4.err; exit 1; }
python - <<P
import json
d=json.loads(open('gpurun_out/split/c4.json').read().strip().splitlines()[-1])
r=d['detail']
print('C4 value %.3f M, %.1f ms/iteration' % (d['value']/1e6, r['ms_per_iteration']))
print(json.dumps(r['second_iteration'])[:1500])
P

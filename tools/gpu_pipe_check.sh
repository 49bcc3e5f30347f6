cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipe
timeout 900 python -m pytest tests/test_gpu_units.py -x -q -W ignore -k "exchange or rccl or mstep or em_loop" 2>&1 | tail -8 | tee gpurun_out/pipe/tests.log
timeout 600 python bench.py --cpu-baseline 0 > gpurun_out/pipe/bench.json 2> gpurun_out/pipe/bench.err; tail -3 gpurun_out/pipe/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/pipe/bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value', d['value'], 'estep_ms', e['estep_ms'], 'acc', e['accumulate_ms'])
print('exchange', e['exchange']['per_rank'])
print('pipe', e.get('estep_pipelined'))
print('err', e.get('error'))
P

#!/bin/bash
# round 5: 8 hardware queues; the SoA decode kernel (parity + timing against round 4's); fresh loops beside their resident twins; C4 fresh phases
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_decode.py -m gpu -x -q > gpurun_out/r5f_dec_tests.txt 2>&1; echo "decode tests rc=$?" >> gpurun_out/r5f_dec_tests.txt
tail -5 gpurun_out/r5f_dec_tests.txt
grep -q "rc=0" gpurun_out/r5f_dec_tests.txt || exit 1
timeout -k 10 300 python3 tools/decode_fuzz.py 30 > gpurun_out/r5f_dec_fuzz.txt 2>&1; echo "fuzz rc=$?" >> gpurun_out/r5f_dec_fuzz.txt; tail -3 gpurun_out/r5f_dec_fuzz.txt
for n in default dec_r4; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n" >> gpurun_out/r5f_dec_bench.txt
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/c5_decode_bench.py 417 4096 20000 3 8192 >> gpurun_out/r5f_dec_bench.txt 2>&1; echo "rc=$?" >> gpurun_out/r5f_dec_bench.txt
done
cat gpurun_out/r5f_dec_bench.txt
for q in 8 4; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> gpurun_out/r5f_fresh.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 60 >> gpurun_out/r5f_fresh.txt 2>&1; echo "rc=$?" >> gpurun_out/r5f_fresh.txt
done
cat gpurun_out/r5f_fresh.txt
timeout -k 10 500 python3 bench.py --cpu-baseline 0 > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5f_bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'fresh', d.get('value_fresh_batches'), 'sustained', d.get('value_sustained'), 'pcie', d.get('value_pcie_inclusive'))
print('fresh', {k: v for k, v in d.get('fresh_batches', {}).items() if k != 'what'})
print('pcie', {k: v for k, v in d.get('pcie_inclusive', {}).items() if k != 'what'})
e=d.get('extra', {}); print('extra error', e.get('error'), 'estep_ms', e.get('estep_ms'))
c4=e.get('configs', {}).get('C4', {}); print('C4', {k: c4.get(k) for k in ('value','ms_per_iteration','phase_ms_rank0','fresh_batches','error')})
print('C4 second', c4.get('second_iteration', {}).get('ms'), c4.get('second_iteration', {}).get('kernel_ms_rank0'))
print('C5', {k: v for k, v in e.get('configs', {}).get('C5', {}).items() if k not in ('task',)})
print('C2', e.get('configs', {}).get('C2', {}).get('value'), 'C3', e.get('configs', {}).get('C3', {}).get('value'), 'C5shard', e.get('configs', {}).get('C5shard', {}).get('ms_per_step'))
PY
tail -3 gpurun_out/r5f_bench.err

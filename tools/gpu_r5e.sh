#!/bin/bash
# round 5: PDE (6,16) per-mixture decisions against no PDE; the remaining new tests; the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r5e_pde.txt
for n in default nopde; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n" >> gpurun_out/r5e_pde.txt
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-6 >> gpurun_out/r5e_pde.txt 2>&1; echo "rc=$?" >> gpurun_out/r5e_pde.txt
done
grep -o "^==.*\|iteration [0-9]\|E-step [0-9.]* ms\|'score_direct': [0-9.]*\|'score_subset': [0-9.]*\|'accumulate': [0-9.]*\|hash(B) [0-9a-f]* hash(acc) [0-9a-f]*" gpurun_out/r5e_pde.txt | paste -sd' ' | sed 's/== /\n== /g; s/iteration/\n  iteration/g'
timeout -k 10 600 python3 -m pytest tests/test_gpu_a_bench_ranks.py tests/test_gpu_parity.py -m gpu -q -k "config4 or third_em" > gpurun_out/r5e_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5e_tests.txt
tail -8 gpurun_out/r5e_tests.txt
timeout -k 10 500 python3 bench.py > gpurun_out/r5e_bench.json 2> gpurun_out/r5e_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5e_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'fresh', d.get('value_fresh_batches'), 'sustained', d.get('value_sustained'), 'pcie', d.get('value_pcie_inclusive'))
print({k: r[k] for k in ('frac','frac_of_f16_dense_peak','frac_executed','traffic','kernel_avg_ms','sustained_value','fresh_batches_value','pcie_inclusive_value','strict_f32_value','strict_f32_frac')})
print('fresh', {k: v for k, v in d.get('fresh_batches', {}).items() if k != 'what'})
print('pcie', {k: v for k, v in d.get('pcie_inclusive', {}).items() if k != 'what'})
c=d['cpu_baseline']; print('cpu', c['value'], c['value_leg'], c['vectorised_value'], c['gemm_value'], c['faithful_value'], c['cores'], c['leg_wall_s'])
e=d.get('extra', {}); print('extra error', e.get('error'), 'estep_ms', e.get('estep_ms'), 'timeline', e.get('timeline_s'))
c4=e.get('configs', {}).get('C4', {}); print('C4', {k: c4.get(k) for k in ('value','ms_per_iteration','fresh_batches','error')})
print('C4 second', c4.get('second_iteration', {}).get('ms'), 'C5', e.get('configs', {}).get('C5', {}).get('value'))
PY
tail -3 gpurun_out/r5e_bench.err

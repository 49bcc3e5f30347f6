#!/bin/bash
# round 5, the record: the default bench line (with the CPU legs), config 4 as EM iterations at both variance floors, smoke, and the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_line.json') if l.startswith('{')][-1])
r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'fresh', d.get('value_fresh_batches'), 'sustained', d.get('value_sustained'), 'pcie', d.get('value_pcie_inclusive'), 'strict', d.get('value_strict_f32'))
print({k: r[k] for k in ('frac','frac_of_f16_dense_peak','frac_executed','traffic','traffic_over_algorithmic','kernel_avg_ms','achieved','fb_kernel_avg_ms','fb_kernel_alone_ms')})
print('fresh', {k: v for k, v in d.get('fresh_batches', {}).items() if k != 'what'})
c=d['cpu_baseline']; print('cpu', c['value'], c['value_leg'], c['vectorised_value'], c['gemm_value'], c['faithful_value'], c['cores'], c['leg_wall_s'])
e=d.get('extra', {}); print('extra error', e.get('error'), 'estep_ms', e.get('estep_ms'), 'acc', e.get('accumulate_ms'), 'peaked', (e.get('estep_peaked') or {}).get('estep_local_ms'))
print('timeline', e.get('timeline_s'))
cf=e.get('configs', {})
for k in ('C2','C3','C5shard','C5','C5_ragged'):
    print(k, {q: cf.get(k, {}).get(q) for q in ('value','ms_per_step','score_kernel_ms','decode_kernel_ms','dp_kernel_ms','wall_s')})
c4=cf.get('C4', {}); print('C4', c4.get('value'), c4.get('ms_per_iteration'), 'fresh', (c4.get('fresh_batches') or {}).get('ms_per_iteration'), 'second', (c4.get('second_iteration') or {}).get('ms'))
print('shim', e.get('zero_change_route', {}).get('deferred_batch_shim'))
PY
for f in 1e-6 1e-3; do
  timeout -k 10 500 python3 bench.py --workload C4 --steps 1 --warmup 1 --iters 4 --c-covariance $f > gpurun_out/r05_c4_floor_$f.json 2> gpurun_out/r05_c4_floor_$f.err; echo "c4 $f rc=$?"
done
python3 - <<'PY'
import json
for f in ('1e-6','1e-3'):
    d=json.loads([l for l in open('gpurun_out/r05_c4_floor_%s.json' % f) if l.startswith('{')][-1])
    print(f, 'value', d['value'], d['ms_per_step'], 'resident', d['ms_per_step_resident_batches'])
    for e in d['detail']['em_iterations']:
        print('   ', e['iteration'], round(e['ms'],1), 'off-pipe', round(e['mixtures_off_the_matrix_pipe'],3), 'whole states off', e['states_off_the_matrix_pipe'], 'floor after', round(e['variances_at_the_floor_after'],3), 'loglik', round(e['loglik_mean_rank0'],1),
              {k: round(v,1) for k, v in e['kernel_ms_rank0'].items() if v > 0.5})
PY
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05_smoke.txt
POCCALA_PARITY_REPORT=1 timeout -k 10 1000 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 30" --args python3 -m pytest tests -m gpu -q > gpurun_out/r05_gpu_suite.txt 2>&1
grep -n "SIGSEGV\|^#[0-9]\|passed\|failed" gpurun_out/r05_gpu_suite.txt | head -40

#!/bin/bash
# round 6, the record: the bench line as the driver runs it, smoke, the whole GPU suite (parity report on request)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt; echo "bench rc=$?"
tail -1 gpurun_out/r06_bench_stdout.txt > gpurun_out/r06_bench_line.json; cp bench_full.json gpurun_out/r06_bench_full.json; wc -c gpurun_out/r06_bench_stdout.txt gpurun_out/r06_bench_line.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_line.json')); r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'final', d['final'], 'traffic', r['traffic'], r['traffic_over_algorithmic'], 'sha', r['kernel_code_sha16'])
print({k: r[k] for k in ('frac','frac_of_f16_dense_peak','frac_executed','kernel_avg_ms','sustained_value','fresh_batches_value','pcie_inclusive_value','strict_f32_value','value_em2_model','value_em3_model','score_kernel_ms_em3','coarse_kernel_ms_em3','estep_ms','accumulate_ms','c4_ms_per_iteration','c4_em_iteration_ms','c2_frames_per_s','c3_frames_per_s','c5_frames_per_s','c5_decode_kernel_ms')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'gpu/cpu', d['gpu_over_cpu'])
PY
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_smoke.txt
POCCALA_PARITY_REPORT=1 timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06_gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -22 gpurun_out/r06_gpu_suite.txt

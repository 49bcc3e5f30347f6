#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "A=1" "PCL_COMPACT_MAIN=0"; do
  env $cfg PCL_COARSE_STATS=1 timeout -k 10 400 python3 bench.py --workload C4 --steps 1 --warmup 1 --iters 5 --c-covariance 1e-6 --cpu-baseline 0 > gpurun_out/r06_c4_tmp.json 2> gpurun_out/r06_c4_tmp.err || { tail -5 gpurun_out/r06_c4_tmp.err; exit 1; }
  [ "$cfg" = "A=1" ] && cp gpurun_out/r06_c4_tmp.json gpurun_out/r06_c4_em_floor_1e-6.json
  python3 - "$cfg" <<'PY'
import json, sys
d=json.loads([l for l in open('gpurun_out/r06_c4_tmp.json') if l.startswith('{')][-1])
print('C4 full,', sys.argv[1])
for e in d['detail']['em_iterations']:
    print('   ', e['iteration'], round(e['ms'],1), 'off-pipe', round(e['mixtures_off_the_matrix_pipe'],3), 'whole', e['states_off_the_matrix_pipe'], 'split', e['split_states'], 'logP', round(e['loglik_mean_rank0'],2), {k: round(v,1) for k, v in e['kernel_ms_rank0'].items() if v > 0.5})
PY
done
timeout -k 10 200 python3 tools/coarse_fuzz.py 60 0 > gpurun_out/soak_coarse.txt 2>&1; echo "coarse rc=$? $(tail -1 gpurun_out/soak_coarse.txt)"
bash tools/gpu_r06_coarse_ab.sh

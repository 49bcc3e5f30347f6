#!/bin/bash
# round 6: states stay SPLIT at a high share of off-pipe mixtures (PCL_SPLIT_MAX) with / without the partial-distance test in the subset launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_split_ab.txt; : > $O
V=$GRAFT_REPO_ROOT/build_ab/lib_pdesub.so
run() { echo "== $1" >> $O; shift; env "$@" timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-6 2>&1 | sed -e 's/cond max.*E-step/E-step/' -e 's/; mean logP.*hash/ hash/' >> $O || exit 1; }
run "default (limit 0.5, no subset PDE)" A=1
run "limit 0.85, no subset PDE" PCL_SPLIT_MAX=0.85
run "limit 0.85, subset PDE" PCL_SPLIT_MAX=0.85 POCCALA_HIP_LIB=$V
run "limit 1.0, subset PDE" PCL_SPLIT_MAX=1.0 POCCALA_HIP_LIB=$V
cat $O
for cfg in "A=1" "PCL_SPLIT_MAX=0.85 POCCALA_HIP_LIB=$V"; do
  env $cfg timeout -k 10 400 python3 bench.py --workload C4 --steps 1 --warmup 1 --iters 3 --c-covariance 1e-6 --cpu-baseline 0 > gpurun_out/r06_c4_tmp.json 2> gpurun_out/r06_c4_tmp.err || exit 1
  python3 - "$cfg" <<'PY'
import json, sys
d=json.loads([l for l in open('gpurun_out/r06_c4_tmp.json') if l.startswith('{')][-1])
print('C4 full,', sys.argv[1].split('/')[-1])
for e in d['detail']['em_iterations']:
    print('   ', e['iteration'], round(e['ms'],1), 'off-pipe', round(e['mixtures_off_the_matrix_pipe'],3), 'whole', e['states_off_the_matrix_pipe'], 'split', e['split_states'], {k: round(v,1) for k, v in e['kernel_ms_rank0'].items() if v > 0.5})
PY
done

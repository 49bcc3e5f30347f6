#!/bin/bash
# round 4: deferred worker shim (tests + bench route), config 4 whole from the initial model, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 600 $O/tests.log python -m pytest -q -x -W ignore tests/test_gpu_dropin.py
tail -12 $O/tests.log
step 300 $O/bench_c4.json python bench.py --workload C4 --steps 1 --warmup 1
python - <<P
import json
d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['detail']['phase_ms_rank0']); print(d['detail']['kernel_ms_per_iteration_rank0']); print(d['detail']['second_iteration']['ms'], d['detail']['second_iteration']['states_off_the_matrix_pipe'])
P
step 900 $O/bench.json python bench.py
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','value_sustained','value_pcie_inclusive','value_strict_f32')})
e=d['extra']
print('zero_change', json.dumps(e.get('zero_change_route'))[:1500])
for k,v in e['configs'].items(): print(k, json.dumps(v)[:400])
print(e['timeline_s'])
print('roofline', {k: d['roofline'][k] for k in ('frac','kernel_avg_ms','fb_kernel_avg_ms','fb_kernel_alone_ms')})
P

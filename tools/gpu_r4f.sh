#!/bin/bash
# round 4: full -m gpu suite (bounds asserted), smoke, forward-backward alone, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 1500 $O/suite.log python -m pytest tests -m gpu -q -W ignore
tail -6 $O/suite.log
step 200 $O/smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -2 $O/smoke.log
for U in 128 1024; do step 200 $O/fb_$U.log python tools/fb_bench.py $U; cat $O/fb_$U.log; done
step 200 $O/c2.log python tools/c2_host_overhead.py; tail -5 $O/c2.log
step 900 $O/bench.json python bench.py
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','value_sustained','value_pcie_inclusive','value_strict_f32')})
e=d['extra']
for k,v in e['configs'].items(): print(k, json.dumps(v)[:600])
print(e['timeline_s'])
print('roofline', {k: d['roofline'][k] for k in ('frac','kernel_avg_ms','fb_kernel_avg_ms','fb_kernel_alone_ms')})
print('estep', e['estep_ms'], e['accumulate_ms'], e['exchange']['per_rank'][0]['derive_ms'], e['exchange']['per_rank'][0]['mstep_owned_ms'])
P

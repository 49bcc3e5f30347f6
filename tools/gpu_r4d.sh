#!/bin/bash
# round 4: the PCIe-inclusive loop leg by leg (three batches in rotation), config 4 whole with its phase breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
for legs in h2d,d2h,vit h2d,d2h d2h h2d vit none; do
  POCCALA_PCIE_LEGS=$legs step 300 $O/pcie_$legs.json python bench.py --cpu-baseline 0 --sustain 1 --extra 0 --steps 20
  python - <<P
import json
try:
    d=json.loads(open('$O/pcie_$legs.json').read().strip().splitlines()[-1])
    p=d.get('pcie_inclusive') or {}
    print('$legs', 'headline %.2f ms' % d['ms_per_step'], 'pcie loop', p.get('ms_per_step'), 'score', p.get('score_kernel_ms'), 'fb span', p.get('fb_span_ms'), p.get('results_intact'), p.get('error'))
except Exception as e: print('$legs', 'failed', e)
P
done
step 300 $O/bench_c4.json python bench.py --workload C4 --steps 2 --warmup 1
python - <<P
import json
d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['detail']['phase_ms_rank0']); print(d['detail']['kernel_ms_per_iteration_rank0'])
P

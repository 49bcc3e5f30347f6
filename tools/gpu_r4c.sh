#!/bin/bash
# round 4, third check: the merged forward-backward launches (3 instead of 6), C2 pipeline with and without the second stream,
# the PCIe-inclusive loop leg by leg, config 4 whole
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 600 $O/fb_tests.log python -m pytest tests/test_gpu_fb_linear.py tests/test_gpu_parity.py tests/test_gpu_units.py -q -x -W ignore -k "fb_linear or linear or bw or golden or baum or estep_end or hmm_acc or out_of_range or caller_logpi or impossible"
tail -5 $O/fb_tests.log
for U in 128 1024; do
  step 200 $O/fb_lin_$U.log python tools/fb_bench.py $U
  cat $O/fb_lin_$U.log
done
PCL_DP_STREAM=1 step 200 $O/c2_dp1.log python tools/c2_host_overhead.py
PCL_DP_STREAM=0 step 200 $O/c2_dp0.log python tools/c2_host_overhead.py
echo "--- C2 two streams"; tail -5 $O/c2_dp1.log; echo "--- C2 one stream"; tail -5 $O/c2_dp0.log
for legs in h2d,d2h,vit h2d,d2h d2h h2d vit none; do
  POCCALA_PCIE_LEGS=$legs step 300 $O/pcie_$legs.json python bench.py --cpu-baseline 0 --sustain 1 --extra 0 --steps 20
  python - <<P
import json
try:
    d=json.loads(open('$O/pcie_$legs.json').read().strip().splitlines()[-1])
    print('$legs', d['ms_per_step'], d.get('value_pcie_inclusive'), (d.get('pcie_inclusive') or {}).get('ms_per_step'))
except Exception as e: print('$legs', 'failed', e)
P
done
step 300 $O/bench_c4.json python bench.py --workload C4 --steps 2 --warmup 1
tail -c 2500 $O/bench_c4.json; echo

#!/bin/bash
# round 4: single-wave posterior workgroups -- tests, forward-backward alone, in-loop spans (kernel trace of the headline loop)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 600 $O/tests.log python -m pytest -q -x -W ignore tests/test_gpu_fb_linear.py tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_dropin.py -k "fb_linear or linear or bw or golden or baum or estep or hmm_acc or out_of_range or caller_logpi or impossible or worker or c2_full or cabi"
tail -4 $O/tests.log
for U in 128 1024; do step 200 $O/fb_$U.log python tools/fb_bench.py $U; cat $O/fb_$U.log; done
step 200 $O/c2.log python tools/c2_host_overhead.py; tail -5 $O/c2.log
step 400 $O/bench.json python bench.py --cpu-baseline 0 --extra 0 --sustain 3 --steps 20
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; p=d.get('pcie_inclusive') or {}
print('value %.3f M, step %.3f ms, score kernel %.3f ms, fb span %.3f ms (alone %.3f), sustained %.3f M, pcie %.3f ms/step fb span %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_avg_ms'], r['fb_kernel_avg_ms'], r['fb_kernel_alone_ms'], (d.get('value_sustained') or 0)/1e6, p.get('ms_per_step', 0), p.get('fb_span_ms')))
P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --cpu-baseline 0 --extra 0 --sustain 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<P
import csv, glob
for f in glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hmm_' in r['Name'] or 'split16' in r['Name']:
            print(r['Name'][:60], r['Calls'], 'avg %.3f ms' % (float(r['AverageNs'])/1e6), 'min %.3f max %.3f' % (float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
P
find $O/trace -name "*.csv" -size +1M -delete

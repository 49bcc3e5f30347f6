#!/bin/bash
# the clock the accumulate consumer holds (GRBM_GUI_ACTIVE / duration) and its matrix-pipe occupancy, for the full kernel and with parts
# switched off (diagnostic builds, results wrong): is the pass energy bound?   usage: gpu_acc_clock.sh default acc_NOP2 ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/acc_clock; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  export POCCALA_HIP_LIB=$R/$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 $R/tools/acc_bench.py > $O/t_$n.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p_$n -- python3 $R/tools/acc_bench.py > $O/p_$n.log 2>&1
  python3 - $O $n <<'P'
import csv, glob, sys, collections
O, n = sys.argv[1], sys.argv[2]
ms = [float(r['AverageNs']) / 1e6 for f in glob.glob('%s/t_%s/**/*kernel_stats.csv' % (O, n), recursive=True) for r in csv.DictReader(open(f)) if 'acc16_consumer' in r['Name']][0]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('%s/p_%s/**/*counter_collection.csv' % (O, n), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'acc16_consumer' in r['Kernel_Name']:
            agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
cyc = agg['GRBM_GUI_ACTIVE'][0] / agg['GRBM_GUI_ACTIVE'][1] / 8
mf = agg['SQ_VALU_MFMA_BUSY_CYCLES'][0] / agg['SQ_VALU_MFMA_BUSY_CYCLES'][1] / 1024
print('%-10s consumer %.2f ms/dispatch, %.3g cycles -> clock %.2f GHz, matrix pipe %.0f %% busy' % (n, ms, cyc, cyc / ms / 1e6, 100 * mf / cyc))
P
done

#!/bin/bash
# the clock the scoring kernel holds and its matrix-pipe occupancy on random operands and on all-zero operands (ZERO=1): power bound?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/score_clock; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for z in "" 1; do
  export ZERO=$z; [ -z "$z" ] && unset ZERO
  n=${z:-0}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 $R/tools/score_bench.py 1024 2048 1000 > $O/t_$n.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p_$n -- python3 $R/tools/score_bench.py 1024 2048 1000 > $O/p_$n.log 2>&1
  python3 - $O $n <<'P'
import csv, glob, sys, collections
O, n = sys.argv[1], sys.argv[2]
ms = [float(r['AverageNs']) / 1e6 for f in glob.glob('%s/t_%s/**/*kernel_stats.csv' % (O, n), recursive=True) for r in csv.DictReader(open(f)) if 'gmm_score_split16' in r['Name']][0]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('%s/p_%s/**/*counter_collection.csv' % (O, n), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gmm_score_split16' in r['Kernel_Name']:
            agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
cyc = agg['GRBM_GUI_ACTIVE'][0] / agg['GRBM_GUI_ACTIVE'][1] / 8
mf = agg['SQ_VALU_MFMA_BUSY_CYCLES'][0] / agg['SQ_VALU_MFMA_BUSY_CYCLES'][1] / 1024
print('%s operands: %.2f ms/launch, %.3g cycles -> clock %.2f GHz, matrix pipe %.0f %% busy' % ('all-zero' if n == '1' else 'random', ms, cyc, cyc / ms / 1e6, 100 * mf / cyc))
P
done

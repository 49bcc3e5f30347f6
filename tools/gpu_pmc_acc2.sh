#!/bin/bash
# rocprofv3 evidence for the accumulate pass (tools/acc_bench.py): one trace pass + PMC passes of <= 4 counters, each under a timeout
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_acc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/acc_bench.py > $O/trace.log 2>&1
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -- python3 $R/tools/acc_bench.py > $O/pmc$i.log 2>&1 || echo "group $i ($grp) failed or timed out" >> $O/failures.log
done
cd $O
python3 - <<'P' > $O/summary.txt
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    print('## --kernel-trace --stats:', f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print('%-72s calls=%s avg_ns=%s pct=%s' % (r['Name'][:72], r['Calls'], r['AverageNs'], r['Percentage']))
print('## --pmc passes (<= 4 counters per pass), per-dispatch averages')
for f in sorted(glob.glob('pmc*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:58], r['Counter_Name'])
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        if 'acc' in k[0]: print('%-60s %-26s per-dispatch=%.6g n=%d' % (k[0], k[1], v[0] / v[1], v[1]))
P
cat $O/summary.txt; cat $O/failures.log 2>/dev/null
find . -name "*.csv" -size +2M -delete

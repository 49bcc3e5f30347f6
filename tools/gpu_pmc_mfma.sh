cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 $R/tools/score_bench.py 512 2048 50 > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/b -- python3 $R/tools/score_bench.py 512 2048 50 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c -- python3 $R/tools/score_bench.py 512 2048 50 > $OUT/c.log 2>&1
cd $OUT; tail -1 a.log; tail -1 b.log
python3 - <<'P'
import csv, glob, collections
for tag in ('a','b'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'gmm_score' not in r['Kernel_Name']: continue
            agg[r['Counter_Name']][0] += float(r['Counter_Value']); agg[r['Counter_Name']][1] += 1
        for k, v in sorted(agg.items()): print('%-28s per-dispatch=%.6g (n=%d)' % (k, v[0]/v[1], v[1]))
for f in glob.glob('c/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gmm_score' in r['Name']: print('avg ns', r['AverageNs'], 'calls', r['Calls'])
P

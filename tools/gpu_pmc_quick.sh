# clock and matrix-pipe occupancy of the scoring kernel: GRBM_GUI_ACTIVE (sum over 8 XCDs) and SQ_VALU_MFMA_BUSY_CYCLES
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/score_bench.py 1024 2048 1000 > $OUT/trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 $R/tools/score_bench.py 1024 2048 1000 > $OUT/pmc.log 2>&1
cd $OUT; for f in $(find . -name "*kernel_stats.csv"); do head -5 $f; done
python3 - <<'P'
import csv, glob, collections
for f in glob.glob('pmc/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:50], r['Counter_Name'])
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        if 'score' in k[0]: print('%-52s %-26s per-dispatch=%.6g n=%d' % (k[0], k[1], v[0] / v[1], v[1]))
P
tail -n 1 trace.log pmc.log

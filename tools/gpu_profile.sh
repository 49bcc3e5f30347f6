# rocprofv3 evidence for bench.py's roofline block (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --cpu-baseline 0 --extra 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_clk -- python3 $R/bench.py $ARGS > $OUT/bench_clk.log 2>&1
cd $OUT && find . -name "*.csv" | head -30; for f in $(find . -name "*kernel_stats.csv"); do echo == $f; head -12 $f; done
python3 - <<'P'
import csv, glob, collections
for tag in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_clk'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'][:60], r['Counter_Name'])
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
        print('==', f)
        for k, v in sorted(agg.items()):
            print('%-62s %-22s sum=%.6g n=%d per-dispatch=%.6g' % (k[0], k[1], v[0], v[1], v[0] / v[1]))
P

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmca
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/acc_bench.py > $OUT/trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 $R/tools/acc_bench.py > $OUT/pmc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -- python3 $R/tools/acc_bench.py > $OUT/pmc2.log 2>&1
cd $OUT; for f in $(find . -name "*kernel_stats.csv"); do head -4 $f | cut -c1-60,330-420; done
python3 - <<'P'
import csv, glob, collections
for f in glob.glob('pmc*/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:50], r['Counter_Name'])
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        if 'accumulate' in k[0]: print('%-52s %-26s per-dispatch=%.6g n=%d' % (k[0], k[1], v[0] / v[1], v[1]))
P

#!/bin/bash
# kernel trace of two EM iterations on the C4 shard (the second has split states)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/em_iter_probe.py 1024 ${ITERS:-2} > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
cut -c1-90,200- $f | head -5
python3 - <<P
import csv,glob
f=glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print('%-70s calls=%s avg_us=%.1f total_ms=%.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
P
find $O/trace -name "*.csv" -size +1M -delete

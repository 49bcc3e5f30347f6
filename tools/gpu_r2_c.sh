#!/bin/bash
# round 2, call C: trace of the accumulate pass (both kernels), quick timings
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/tools/acc_bench.py > $O/acc_f16.log 2>&1
timeout 600 python3 $R/tools/estep_peaked_bench.py > $O/peaked_f16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/acc_bench.py > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_peaked -- python3 $R/tools/estep_peaked_bench.py > $O/trace_peaked.log 2>&1
cd $O; cat acc_f16.log peaked_f16.log; for f in $(find . -name "*kernel_stats.csv"); do echo == $f; python3 - $f <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-70s calls=%s avg_ns=%s pct=%s' % (r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage']))
P
done
find . -name "*.csv" -size +3M -delete

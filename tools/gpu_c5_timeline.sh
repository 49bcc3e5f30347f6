#!/bin/bash
# kernel timeline of the streamed C5 run: do the scoring and decode kernels overlap?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
n=${1:-default}
if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
export POCCALA_HIP_LIB=$R/$lib
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$n -- python3 $R/tools/c5_decode_bench.py ${C5_ARGS:-417 4096 20000 3 8192} > $O/run_$n.log 2>&1
f=$(find $O/trace_$n -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in rows if 'decode' in r['Kernel_Name'] or 'score_split16' in r['Kernel_Name']]
ev.sort()
t0 = ev[0][0]
for s, e, n, q in ev[-int(__import__("os").environ.get("NEV", "24")):]:
    print('%10.2f %10.2f %8.2f ms  q=%s %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
P
tail -4 $O/run_$n.log | cut -c1-300

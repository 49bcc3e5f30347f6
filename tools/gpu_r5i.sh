#!/bin/bash
# round 5: subset PDE variants (two E-steps per iteration, the second timed); what sits in the gaps between scoring launches; the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r5i_pde.txt
for n in default sub3 sub2 sub2r4 nosub2; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n" >> gpurun_out/r5i_pde.txt
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/em_iter_probe.py 1024 3 1e-6 >> gpurun_out/r5i_pde.txt 2>&1; echo "rc=$?" >> gpurun_out/r5i_pde.txt
done
grep -o "^==.*\|iteration [0-9]\|E-step [0-9.]* ms\|'score': [0-9.]*\|'score_direct': [0-9.]*\|'score_subset': [0-9.]*\|'accumulate': [0-9.]*\|hash(B) [0-9a-f]* hash(acc) [0-9a-f]*" gpurun_out/r5i_pde.txt | paste -sd' ' | sed 's/== /\n== /g; s/iteration/\n  iteration/g'
cd /tmp && export TMPDIR=/tmp
PROBE_TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5i_trace -- python3 $GRAFT_REPO_ROOT/tools/fresh_batch_probe.py C4shard 30 > $GRAFT_REPO_ROOT/gpurun_out/r5i_trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py gpurun_out/r5i_trace > gpurun_out/r5i_gaps.txt 2>&1; cat gpurun_out/r5i_gaps.txt
find gpurun_out/r5i_trace -name "*.csv" -size +4M -delete
timeout -k 10 500 python3 bench.py --cpu-baseline 0 > gpurun_out/r5i_bench.json 2> gpurun_out/r5i_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5i_bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'fresh', d.get('value_fresh_batches'), 'sustained', d.get('value_sustained'), 'pcie', d.get('value_pcie_inclusive'))
print('fresh', {k: v for k, v in d.get('fresh_batches', {}).items() if k != 'what'})
print('pcie', {k: v for k, v in d.get('pcie_inclusive', {}).items() if k != 'what'})
e=d.get('extra', {}); print('extra error', e.get('error'), 'estep_ms', e.get('estep_ms'), 'pipelined', (e.get('estep_pipelined') or {}).get('estep_ms'))
c4=e.get('configs', {}).get('C4', {}); print('C4', {k: c4.get(k) for k in ('value','ms_per_iteration','phase_ms_rank0','error')}); print('C4 fresh', {k: v for k, v in c4.get('fresh_batches', {}).items() if k != 'what'})
print('C4 second', c4.get('second_iteration', {}).get('ms'), c4.get('second_iteration', {}).get('kernel_ms_rank0'))
PY
tail -3 gpurun_out/r5i_bench.err

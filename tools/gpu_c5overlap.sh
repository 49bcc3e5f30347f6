#!/bin/bash
# config 5 streamed: can the token passing run BESIDE the scoring if the scoring leaves registers (LDS padding: 2 workgroups per CU) and the decode workgroups are small (256 threads)?
cd $GRAFT_REPO_ROOT
run() { # tag, lib, pad
  if [ -n "$2" ]; then export POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$2.so; else unset POCCALA_HIP_LIB; fi
  export PCL_SCORE_LDS_PAD_KB=$3
  timeout -k 10 400 python bench.py --workload C5 > gpurun_out/c5o_$1.json 2> gpurun_out/c5o_$1.err || { tail -3 gpurun_out/c5o_$1.err; exit 1; }
  python - <<P
import json
d=json.loads(open('gpurun_out/c5o_$1.json').read().strip().splitlines()[-1]); r=d['detail']
print('%-22s value %.3f M  wall %.3f s  score %.1f ms/chunk  decode %.1f ms/chunk | ragged %.3f M' % ('$1', d['value']/1e6, r['wall_s'], r['score_kernel_ms_per_chunk'], r['decode_kernel_ms_per_chunk'], d['ragged']['value']/1e6))
P
}
run pad34 "" 34
run pad40 "" 40
run w5 dec_w5 0
run w5_pad34 dec_w5 34
run w6_pad34 dec_w6 34

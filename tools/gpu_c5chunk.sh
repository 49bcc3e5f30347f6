#!/bin/bash
# config 5 whole: how the stream's chunk size moves the throughput (decode is one workgroup per utterance)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5chunk
for per in 139 417 834 1668; do
  timeout -k 10 400 python bench.py --workload C5 --c5-chunk $per > gpurun_out/c5chunk/c5_$per.json 2> gpurun_out/c5chunk/c5_$per.err || { tail -5 gpurun_out/c5chunk/c5_$per.err; exit 1; }
  python - <<P
import json
d=json.loads(open('gpurun_out/c5chunk/c5_$per.json').read().strip().splitlines()[-1])
r=d['detail']; g=d['ragged']
print($per, 'value %.3f M  wall %.3f s  score %.1f ms/chunk decode %.1f ms/chunk  tokens %.0f  pinned %.0f MB | ragged %.3f M' % (d['value']/1e6, r['wall_s'], r['score_kernel_ms_per_chunk'], r['decode_kernel_ms_per_chunk'], r['live_tokens_mean'], r['pinned_host_bytes']/1e6, g['value']/1e6))
P
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for occ in 0 2; do
  POCCALA_STREAM_OCCUPANCY=$occ timeout -k 10 400 python bench.py --workload C5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['detail']; g=d['ragged']; print('occupancy $occ: uniform %.3f M (score %.1f decode %.1f) | ragged %.3f M (score %.1f decode %.1f)' % (d['value']/1e6, r['score_kernel_ms_per_chunk'], r['decode_kernel_ms_per_chunk'], g['value']/1e6, g['score_kernel_ms_per_chunk'], g['decode_kernel_ms_per_chunk']))" || exit 1
done; done

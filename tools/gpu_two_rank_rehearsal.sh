#!/bin/bash
# the N > 1 control flow on one device (host transport; C2's statistics fit its 256 MiB limit): self-spawned ranks
cd $GRAFT_REPO_ROOT
export POCCALA_SHARE_DEVICE=1
timeout 900 python bench.py --gpus 2 --workload C2 --steps 5 --warmup 1 2>/tmp/e1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']
print(d['value'], d['n_gpus'], d['config']['transport'], d['config']['rccl_nranks'], 'estep_ms', e.get('estep_ms'), 'err', e.get('error'))
print([ (r['rank'], round(r['exchange_ms'],2), round(r['reduce_scatter_ms'],2), round(r['mstep_owned_ms'],2), round(r['all_gather_ms'],2)) for r in e['exchange']['per_rank']])"
tail -2 /tmp/e1

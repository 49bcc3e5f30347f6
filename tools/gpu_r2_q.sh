#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload C5shard --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-1500
timeout 600 python tools/c5_decode_bench.py 417 4096 20000 3 8192 2>&1 | grep "^resident\|^tree" | cut -c1-300
POCCALA_SHARE_DEVICE=1 timeout 600 python bench.py --workload C5shard --gpus 2 --steps 2 --warmup 1 --utts 100 2>&1 | tail -1 | cut -c1-400

#!/bin/bash
# C5: decode tests, the resident shard, and a stream of 8 full-size chunks through decode_stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q -W ignore -x 2>&1 | tail -3
timeout 600 python tools/c5_decode_bench.py 417 4096 20000 3 8192 2>&1 | tee $O/c5_shard.log | cut -c1-330
timeout 900 python tools/c5_decode_bench.py 3336 4096 20000 8 8192 2>&1 | grep -v "^tree\|^utterance\|^decode ~" | tee $O/c5_stream8.log | cut -c1-330

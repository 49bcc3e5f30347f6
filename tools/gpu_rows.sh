#!/bin/bash
# compaction by rows against by segments: same statistics bit for bit, time per pass
cd $GRAFT_REPO_ROOT
for v in 0 1; do echo "PCL_ACC_ROWS=$v"; PCL_ACC_ROWS=$v ACC_PASSES=6 timeout -k 10 200 python tools/acc_bench.py 2>&1 | tail -2 || exit 1; done
timeout -k 10 900 python -m pytest tests/test_gpu_accumulate.py tests/test_gpu_parity.py -m gpu -q -W ignore -x 2>&1 | tail -3

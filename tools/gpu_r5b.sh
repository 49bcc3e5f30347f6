#!/bin/bash
# round 5: the fresh-batch sweep after the non-blocking create / deferred destroy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 40 > gpurun_out/r5b_fresh.txt 2>&1; echo "fresh rc=$?" >> gpurun_out/r5b_fresh.txt
timeout -k 10 200 python3 tools/fresh_batch_probe.py C2 400 >> gpurun_out/r5b_fresh.txt 2>&1; echo "fresh C2 rc=$?" >> gpurun_out/r5b_fresh.txt
PCL_DESTROY_SYNC=1 timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 40 >> gpurun_out/r5b_fresh.txt 2>&1; echo "fresh (sync destroy) rc=$?" >> gpurun_out/r5b_fresh.txt
cat gpurun_out/r5b_fresh.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not c5_full and not c4_full" > gpurun_out/r5b_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5b_tests.txt
tail -15 gpurun_out/r5b_tests.txt

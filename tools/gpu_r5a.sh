#!/bin/bash
# round 5, first measurements: where a fresh batch per step stands today; EM iterations at the reference's variance floor
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 40 > gpurun_out/r5a_fresh.txt 2>&1; echo "fresh rc=$?" >> gpurun_out/r5a_fresh.txt
timeout -k 10 200 python3 tools/fresh_batch_probe.py C2 400 >> gpurun_out/r5a_fresh.txt 2>&1; echo "fresh C2 rc=$?" >> gpurun_out/r5a_fresh.txt
timeout -k 10 400 python3 tools/em_iter_probe.py 1024 5 1e-6 > gpurun_out/r5a_em_1e-6.txt 2>&1; echo "rc=$?" >> gpurun_out/r5a_em_1e-6.txt
timeout -k 10 400 python3 tools/em_iter_probe.py 1024 5 1e-3 > gpurun_out/r5a_em_1e-3.txt 2>&1; echo "rc=$?" >> gpurun_out/r5a_em_1e-3.txt
cat gpurun_out/r5a_fresh.txt gpurun_out/r5a_em_1e-6.txt gpurun_out/r5a_em_1e-3.txt

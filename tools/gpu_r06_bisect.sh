#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in tests/test_gpu_lifecycle.py tests/test_gpu_em_shaped.py; do
  n=$(basename $f .py)
  MALLOC_CHECK_=3 timeout -k 10 400 python3 -X faulthandler -m pytest $f -x -q > gpurun_out/r06_bisect_$n.txt 2>&1; echo "$f rc=$?"
  tail -25 gpurun_out/r06_bisect_$n.txt
done

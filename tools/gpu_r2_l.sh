#!/bin/bash
# accumulate: image budget (number of state groups) and overlap
cd $GRAFT_REPO_ROOT
for mb in 2048 4096 8192 16384; do
  for ov in 1 0; do
    echo -n "PCL_ACC_IMAGE_MB=$mb PCL_ACC_OVERLAP=$ov: "; PCL_ACC_IMAGE_MB=$mb PCL_ACC_OVERLAP=$ov timeout 300 python tools/acc_bench.py 2>&1 | tail -1
  done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for mb in 2048 3072 6144; do for div in 12 6; do echo "PCL_ACC_IMAGE_MB=$mb PCL_ACC_FIRST_DIV=$div: $(PCL_ACC_IMAGE_MB=$mb PCL_ACC_FIRST_DIV=$div ACC_PASSES=6 timeout -k 10 200 python tools/acc_bench.py 2>&1 | tail -1 | cut -c1-50)" || exit 1; done; done

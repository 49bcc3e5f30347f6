#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 12 6 24; do echo "PCL_ACC_FIRST_DIV=$d"; PCL_ACC_FIRST_DIV=$d ACC_PASSES=6 timeout -k 10 200 python tools/acc_bench.py 2>&1 | tail -3 || exit 1; done

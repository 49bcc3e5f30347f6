#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in default sub_r2b4 sub_r2b3 sub_r3b4 sub_r4b2; do
  if [ $v = default ]; then unset POCCALA_HIP_LIB; else export POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$v.so; fi
  echo "$v: $(timeout -k 10 300 python tools/em_iter_probe.py 1024 2 2>&1 | tail -1 | grep -o "'score_subset': [0-9.]*")" || exit 1
done

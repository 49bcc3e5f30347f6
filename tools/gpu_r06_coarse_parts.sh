#!/bin/bash
# round 6: the coarse pass's time by parts (tools/coarse_time_probe.py), on the C4 shard's model after N EM iterations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_coarse_parts.txt; : > $O
for n in ${NS:-2 1}; do
  timeout -k 10 200 python3 tools/coarse_time_probe.py make $n >> $O 2>&1 || exit 1
  for np in 1 3; do PCL_COARSE_PASSES=$np timeout -k 10 100 python3 tools/coarse_time_probe.py time >> $O 2>&1; done
  for v in ${VARIANTS:-cexp1 cexp2 cexp4}; do POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$v.so timeout -k 10 100 python3 tools/coarse_time_probe.py time >> $O 2>&1; done
done
cat $O

#!/bin/bash
# round 6: the coarse kernel at 2 (default) and 3 waves per SIMD; tools/em_iter_probe.py on the C4 shard, kernel ms per batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_coarse_ab.txt; : > $O
for lib in "$GRAFT_REPO_ROOT/poccala_amd/libpoccala_hip.so" "$GRAFT_REPO_ROOT/build_ab/lib_coarse3.so"; do
  for rep in 1 2; do
    echo "== $(basename $lib) run $rep" >> $O
    POCCALA_HIP_LIB=$lib PCL_COARSE_STATS=1 timeout -k 10 300 python3 tools/em_iter_probe.py 1024 3 1e-6 2>&1 | sed -e 's/cond max.*E-step/E-step/' -e 's/; mean logP.*hash/ hash/' | grep -o "iteration [0-9].*off-pipe mixtures [0-9.]*%\|'score': [0-9.]*\|'score_coarse': [0-9.]*\|exact_pairs_per_pass': [0-9]*" | paste -sd' ' >> $O
  done
done
cat $O

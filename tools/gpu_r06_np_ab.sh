#!/bin/bash
# round 6: the coarse pass with one product per term (default) against the two-piece operands' three (PCL_COARSE_PASSES=3);
# tools/em_iter_probe.py on the C4 shard, kernel ms per batch and the pairs evaluated in direct form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_np_ab.txt; : > $O
for np in ${NPS:-1 3}; do
  for rep in 1 2; do
    echo "== PCL_COARSE_PASSES=$np run $rep" >> $O
    PCL_COARSE_PASSES=$np PCL_COARSE_STATS=1 timeout -k 10 300 python3 tools/em_iter_probe.py 1024 ${ITERS:-5} 1e-6 2>&1 | sed -e 's/cond max.*E-step/E-step/' -e 's/; mean logP.*hash/ hash/' | grep -o "iteration [0-9].*off-pipe mixtures [0-9.]*%\|'score': [0-9.]*\|'score_coarse': [0-9.]*\|exact_pairs_per_pass': [0-9]*\|hash(B) [0-9a-f]*" | paste -sd' ' | sed 's/iteration/\niteration/g' >> $O
  done
done
cat $O

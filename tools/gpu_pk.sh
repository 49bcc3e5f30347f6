#!/bin/bash
# A/B of the direct-form scoring kernel variants (packed f32 pairs of frames against the scalar form)
cd $GRAFT_REPO_ROOT
for v in np_r3 pk_r4 pk_r2 pk_r4g2; do
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$v.so PCL_SCORE_VARIANT=1 CHECK=1 timeout -k 10 120 python tools/score_bench.py 256 2048 50 2>&1 | tail -2 || exit 1
done

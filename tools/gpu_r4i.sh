#!/bin/bash
# round 4: the multi-wave scaled forward-backward (N up to 256): tests, fuzz, timing against the log-domain kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"; tail -4 $log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; exit 1; fi
    return 0
}
step 300 $O/tests.log python -m pytest -q -x -W ignore tests/test_gpu_fb_linear.py
step 300 $O/fuzz.log python tools/fb_linear_fuzz.py 0 120
for L in 21 40 60 84; do for U in 128 1024; do
  step 120 $O/fb_${L}_${U}.log python tools/fb_bench.py $U $L
  PCL_FB_LINEAR=0 step 120 $O/fblog_${L}_${U}.log python tools/fb_bench.py $U $L
done; done

#!/bin/bash
# round 4: the scaled forward-backward (tests + timing against the log-domain kernels), then the whole -m gpu suite with the
# parity bookkeeping in discovery mode (every violation of the tightened bounds is recorded, none stops the run).
# A step that times out or is killed ends the script (no further GPU step after a hang); a failing test does not.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
step() {   # step <seconds> <log> <command...>
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 600 $O/fb_tests.log python -m pytest tests/test_gpu_fb_linear.py -q -x -W ignore
tail -15 $O/fb_tests.log
step 300 $O/fb_golden.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -q -W ignore -k "bw or golden or baum or estep_end or hmm_acc"
tail -8 $O/fb_golden.log
for U in 128 1024; do
  PCL_FB_LINEAR=1 step 200 $O/fb_lin_$U.log python tools/fb_bench.py $U
  PCL_FB_LINEAR=0 step 200 $O/fb_log_$U.log python tools/fb_bench.py $U
  cat $O/fb_lin_$U.log $O/fb_log_$U.log
done
POCCALA_PARITY_SOFT=1 step 1500 $O/suite.log python -m pytest tests -m gpu -q -W ignore -s
grep -c "PARITY VIOLATION" $O/suite.log; grep "PARITY VIOLATION" $O/suite.log | sort | uniq -c | sort -rn | head -40
tail -12 $O/suite.log

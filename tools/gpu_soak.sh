#!/bin/bash
# round 4 soak: randomised parity of the new forward-backward, more seeds of the suite's fuzz tests, decoder fuzz, streamed decode + EM loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"; tail -3 $log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; exit 1; fi
    return 0
}
step 500 $O/fb_fuzz.log python tools/fb_linear_fuzz.py 0 150
step 500 $O/parity_soak.log python tools/parity_soak.py 100 40
step 500 $O/decode_fuzz.log python tools/decode_fuzz.py
step 500 $O/soak_stream.log python tools/soak_stream_em.py
step 700 $O/split_fuzz.log python tools/split_fuzz.py 100 60

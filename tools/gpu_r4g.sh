#!/bin/bash
# round 4: changed tests, config 4 whole over two ranks on one device (rehearsal transport), forward-backward profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 600 $O/tests.log python -m pytest -q -x -W ignore tests/test_gpu_parity.py::test_cabi_error_codes tests/test_gpu_decode.py::test_decode_stream_equals_chunk_by_chunk tests/test_gpu_fb_linear.py tests/test_gpu_dropin.py
tail -5 $O/tests.log
POCCALA_SHARE_DEVICE=1 step 500 $O/c4_two_ranks.json python bench.py --workload C4 --gpus 2 --steps 1 --warmup 1
tail -c 1800 $O/c4_two_ranks.json; echo
PARTS=fb bash tools/gpu_profile.sh > $O/prof.log 2>&1
tail -30 $O/prof.log | cut -c1-300

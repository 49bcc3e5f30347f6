#!/bin/bash
# round 4, second check: configs 4 and 5 at their stated size (tests + bench lines), the exchange at world 4 / 8 (threads), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "rc=$rc $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hung or was killed: stopping"; tail -5 $log; exit 1; fi
    return 0
}
step 900 $O/tests_new.log python -m pytest -q -x -W ignore -s tests/test_gpu_units.py::test_exchange_world_4_and_8_equals_single_rank tests/test_gpu_parity.py::test_c4_full_size_one_statistics_block tests/test_gpu_decode.py::test_c5_full_corpus_streamed "tests/test_gpu_dropin.py::test_worker_flow_matches_reference"
tail -25 $O/tests_new.log
step 600 $O/bench.json python bench.py
tail -c 600 $O/bench.json; echo
step 300 $O/bench_c4.json python bench.py --workload C4 --steps 2 --warmup 1
tail -c 1500 $O/bench_c4.json; echo
step 300 $O/bench_c5.json python bench.py --workload C5
tail -c 1500 $O/bench_c5.json; echo

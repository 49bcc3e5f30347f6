#!/bin/bash
# round 6: a list of test files (default: the new ones), one pytest process
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TESTS:-"tests/test_gpu_lifecycle.py tests/test_gpu_em_shaped.py tests/test_gpu_a_bench_ranks.py"}
timeout -k 10 ${LIMIT:-1000} python3 -m pytest $T -x -q -s --durations=12 > gpurun_out/r06_tests.txt 2>&1; echo "tests rc=$?"
tail -${TAIL:-60} gpurun_out/r06_tests.txt

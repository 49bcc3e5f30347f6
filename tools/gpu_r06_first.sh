#!/bin/bash
# round 6, first call: the bench line as the driver runs it (stdout must END in the compact line), then the new tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --gpus 1 --steps 25 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt; echo "bench rc=$?"
wc -c gpurun_out/r06_bench_stdout.txt; tail -c 3000 gpurun_out/r06_bench_stdout.txt; echo; tail -5 gpurun_out/r06_bench_stderr.txt
cp bench_full.json gpurun_out/r06_bench_full.json
timeout -k 10 900 python3 -m pytest tests/test_gpu_lifecycle.py tests/test_gpu_em_shaped.py tests/test_gpu_a_bench_ranks.py -x -q -s --durations=8 > gpurun_out/r06_new_tests.txt 2>&1; echo "tests rc=$?"
tail -40 gpurun_out/r06_new_tests.txt

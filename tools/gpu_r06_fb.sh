#!/bin/bash
# round 6: the forward-backward kernels after the barrier / EXEC changes -- parity (golden alpha/beta, tripwires, fuzz) and timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_fb_linear.py tests/test_gpu_fuzz_hmm.py -x -q --durations=6 > gpurun_out/r06_fb_tests.txt 2>&1; echo "fb tests rc=$?"; tail -12 gpurun_out/r06_fb_tests.txt
timeout -k 10 200 python3 tools/fb_linear_fuzz.py 0 80 > gpurun_out/r06_fb_fuzz.txt 2>&1; echo "fb_linear_fuzz rc=$?"; tail -4 gpurun_out/r06_fb_fuzz.txt
for U in 128 1024; do timeout -k 10 120 python3 tools/fb_bench.py $U > gpurun_out/r06_fb_bench_$U.txt 2>&1; echo "fb_bench $U rc=$?"; tail -3 gpurun_out/r06_fb_bench_$U.txt; done
timeout -k 10 120 python3 tools/fb_bench.py 1024 21 > gpurun_out/r06_fb_bench_65.txt 2>&1; echo "fb_bench N=65 rc=$?"; tail -3 gpurun_out/r06_fb_bench_65.txt
timeout -k 10 120 python3 tools/fb_bench.py 128 40 > gpurun_out/r06_fb_bench_122.txt 2>&1; echo "fb_bench N=122 rc=$?"; tail -3 gpurun_out/r06_fb_bench_122.txt

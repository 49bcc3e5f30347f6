#!/bin/bash
# round 6: the E-step fuzz seeds the soak failed (statistics `acc`, before the coarse pass's pairs were evaluated in the accumulate pass's own f32 arithmetic) + the suites around
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for sd in 328 334 393 404 463 478 510 596 758 771 822 866 878 887 896 980 1536 1546; do timeout -k 10 60 python3 tests/test_gpu_fuzz_estep.py 1 $sd 2>&1 | grep "f32\|cases"; done > gpurun_out/r06_refuzz.txt 2>&1
grep -c "f32 ok" gpurun_out/r06_refuzz.txt; grep "FAILED" gpurun_out/r06_refuzz.txt | cut -c1-220
timeout -k 10 300 python3 tests/test_gpu_fuzz_estep.py 700 300 > gpurun_out/soak_estep.txt 2>&1; echo "estep rc=$? $(tail -1 gpurun_out/soak_estep.txt)"
timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 60 1500 > gpurun_out/soak_estep_big.txt 2>&1; echo "estep big rc=$? $(tail -1 gpurun_out/soak_estep_big.txt)"
timeout -k 10 900 python3 -m pytest tests/test_gpu_coarse.py tests/test_gpu_parity.py tests/test_gpu_accumulate.py tests/test_gpu_fuzz_estep.py tests/test_gpu_em_shaped.py tests/test_gpu_units.py tests/test_gpu_decode.py -x -q > gpurun_out/r06_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06_tests.txt
timeout -k 10 200 python3 tools/coarse_fuzz.py 60 0 > gpurun_out/soak_coarse.txt 2>&1; echo "coarse rc=$? $(tail -1 gpurun_out/soak_coarse.txt)"

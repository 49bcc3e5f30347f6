#!/bin/bash
# round 6: the coarse pass over the off-pipe mixtures -- parity first (the tests that reach split states), then what it costs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "split or third_em or conditioning or fixup or flag or far_above" > gpurun_out/r06_coarse_tests.txt 2>&1; echo "parity subset rc=$?"; tail -5 gpurun_out/r06_coarse_tests.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz_estep.py tests/test_gpu_em_shaped.py -x -q -s > gpurun_out/r06_coarse_tests2.txt 2>&1; echo "fuzz + em-shaped rc=$?"; tail -12 gpurun_out/r06_coarse_tests2.txt
O=gpurun_out/r06_coarse_probe.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-6 2>&1 | sed -e 's/cond max.*E-step/E-step/' -e 's/; mean logP.*hash/ hash/' >> $O || exit 1; }
run "coarse pass (default)" PCL_COARSE_STATS=1
run "PCL_COARSE=0 (round 5)" PCL_COARSE=0
cat $O

#!/bin/bash
# round 5: fresh-batch sweep with batch-own completion events + staged descriptor uploads; partial-distance elimination A/B; new tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 60 > gpurun_out/r5c_fresh.txt 2>&1; echo "fresh rc=$?" >> gpurun_out/r5c_fresh.txt
timeout -k 10 200 python3 tools/fresh_batch_probe.py C2 400 >> gpurun_out/r5c_fresh.txt 2>&1; echo "fresh C2 rc=$?" >> gpurun_out/r5c_fresh.txt
cat gpurun_out/r5c_fresh.txt
for n in default nopde pde2_8 pde6_16 pde3_0; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n" >> gpurun_out/r5c_pde.txt
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-6 >> gpurun_out/r5c_pde.txt 2>&1; echo "rc=$?" >> gpurun_out/r5c_pde.txt
done
echo "== default at 1e-3" >> gpurun_out/r5c_pde.txt
timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-3 >> gpurun_out/r5c_pde.txt 2>&1
grep -o "^==.*\|iteration [0-9]\|E-step [0-9.]* ms\|'score_direct': [0-9.]*\|'score_subset': [0-9.]*\|hash(B) [0-9a-f]* hash(acc) [0-9a-f]*" gpurun_out/r5c_pde.txt | paste -sd' ' | sed 's/== /\n== /g; s/iteration/\n  iteration/g'
timeout -k 10 1100 python3 -m pytest tests/test_gpu_a_bench_ranks.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or third_em or split_states or ill_conditioned or variants" > gpurun_out/r5c_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5c_tests.txt
tail -25 gpurun_out/r5c_tests.txt

#!/bin/bash
# round 5: where the fresh-batch step's last 0.8 ms go; partial-distance elimination with one decision per group of 4; the rest of the new tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 60 > gpurun_out/r5d_fresh.txt 2>&1; echo "fresh rc=$?" >> gpurun_out/r5d_fresh.txt
cat gpurun_out/r5d_fresh.txt
rm -f gpurun_out/r5d_pde.txt
for n in default nopde pde6_14 pde8_0 pde3_9; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n" >> gpurun_out/r5d_pde.txt
  POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/em_iter_probe.py 1024 4 1e-6 >> gpurun_out/r5d_pde.txt 2>&1; echo "rc=$?" >> gpurun_out/r5d_pde.txt
done
grep -o "^==.*\|iteration [0-9]\|E-step [0-9.]* ms\|'score_direct': [0-9.]*\|'score_subset': [0-9.]*\|'accumulate': [0-9.]*\|hash(B) [0-9a-f]* hash(acc) [0-9a-f]*" gpurun_out/r5d_pde.txt | paste -sd' ' | sed 's/== /\n== /g; s/iteration/\n  iteration/g'
timeout -k 10 1000 python3 -m pytest tests/test_gpu_a_bench_ranks.py tests/test_gpu_parity.py -m gpu -q -k "bench or dying or watchdog or third_em or split_states or ill_conditioned or variants" > gpurun_out/r5d_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5d_tests.txt
tail -25 gpurun_out/r5d_tests.txt

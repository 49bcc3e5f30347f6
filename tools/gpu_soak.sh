#!/bin/bash
# round 5: the randomised harnesses at length (one call, ~18 min): E-step against the oracle, HMMs against the oracle, the token passing against its
# restatement, a sweep against its unhurried twin.  Each under its own timeout; joined with && so that nothing runs after a failure or a kill.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=${SOAK_SEED:-300}       # (E-step fuzz: seeds below 1000 are the small draws, from 1000 the big ones; HMM fuzz: from 5000 the long utterances)
P=${SOAK_PART:-12}      # 1 = the randomised harnesses (~18 min), 2 = lifecycle + the full-size tests at full depth (~6 min); a gpurun call lasts at most 20 min: run the parts in two calls
if [[ $P == *1* ]]; then
ES=$((S % 300))            # (the small draws are the seeds below 1000: 700 of them from S mod 300 -- a seed of 20000 used to ask for a negative count and ran none)
timeout -k 10 300 python3 tests/test_gpu_fuzz_estep.py 700 $ES > gpurun_out/soak_estep.txt 2>&1; echo "estep rc=$? $(tail -1 gpurun_out/soak_estep.txt)"
timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 60 $((S + 1200)) > gpurun_out/soak_estep_big.txt 2>&1; echo "estep big rc=$? $(tail -1 gpurun_out/soak_estep_big.txt)"
timeout -k 10 200 python3 tests/test_gpu_fuzz_hmm.py 500 $S > gpurun_out/soak_hmm.txt 2>&1; echo "hmm rc=$? $(tail -1 gpurun_out/soak_hmm.txt)"
timeout -k 10 200 python3 tests/test_gpu_fuzz_hmm.py 40 $((S + 5100)) > gpurun_out/soak_hmm_long.txt 2>&1; echo "hmm long rc=$? $(tail -1 gpurun_out/soak_hmm_long.txt)"
timeout -k 10 200 python3 tools/decode_fuzz.py 150 $S > gpurun_out/soak_decode.txt 2>&1; echo "decode rc=$? $(tail -1 gpurun_out/soak_decode.txt)"
timeout -k 10 420 python3 tools/sweep_fuzz.py 70 60 $S > gpurun_out/soak_sweep.txt 2>&1; echo "sweep rc=$? $(tail -1 gpurun_out/soak_sweep.txt)"
grep -h "FAILED\|differs\|mismatch" gpurun_out/soak_*.txt | grep -v " 0 mismatch" | head -20
fi
if [[ $P == *2* ]]; then
# round 6: the context / batch lifecycle under load (three seeds x 200 teardowns), and the full-size tests at their full depth (POCCALA_SOAK:
# 24 utterances of the C4 shard and of C3 against the oracle, both ends of the C5 shard, the f64 statistics at M = 2048)
for sd in 1 2 3; do timeout -k 10 300 python3 tools/lifecycle_stress.py --iters 200 --seed $sd > gpurun_out/soak_lifecycle_$sd.txt 2>&1; echo "lifecycle seed $sd rc=$? $(tail -1 gpurun_out/soak_lifecycle_$sd.txt)"; done
timeout -k 10 300 python3 tools/coarse_fuzz.py 120 0 > gpurun_out/soak_coarse.txt 2>&1; echo "coarse rc=$? $(tail -1 gpurun_out/soak_coarse.txt)"
POCCALA_SOAK=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_accumulate.py tests/test_gpu_em_shaped.py -q -k "c4_shard or c3_deep or c5_shard" > gpurun_out/soak_fullsize.txt 2>&1; echo "full-size rc=$? $(tail -1 gpurun_out/soak_fullsize.txt)"
fi

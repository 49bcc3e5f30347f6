#!/bin/bash
# round 5: the randomised harnesses on the alternative routes (strict-f32 and direct-form scoring, the log-domain forward-backward, one stream, no split states)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 3 1; do
  PCL_SCORE_VARIANT=$v timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_v$v.txt 2>&1; echo "variant $v rc=$? $(tail -1 gpurun_out/knob_estep_v$v.txt)"
done
PCL_FB_LINEAR=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_hmm.py 200 0 > gpurun_out/knob_hmm_log.txt 2>&1; echo "fb log-domain rc=$? $(tail -1 gpurun_out/knob_hmm_log.txt)"
PCL_FB_LINEAR=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 100 0 > gpurun_out/knob_estep_log.txt 2>&1; echo "estep fb log-domain rc=$? $(tail -1 gpurun_out/knob_estep_log.txt)"
PCL_DP_STREAM=0 PCL_FEWER_MARKERS=0 PCL_ZERO_ASYNC=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 100 0 > gpurun_out/knob_estep_1s.txt 2>&1; echo "one stream rc=$? $(tail -1 gpurun_out/knob_estep_1s.txt)"
PCL_SPLIT_MAX=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_nosplit.txt 2>&1; echo "no split states rc=$? $(tail -1 gpurun_out/knob_estep_nosplit.txt)"
# round 6: the round 4-5 routes beside the coarse pass / the compacted main layout, states without an on-pipe mixture through the coarse pass, and the
# BASELINE mixture counts (seeds 6000..: 1024 / 2048 / 4096 mixtures per state) on the default route
PCL_COARSE=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_nocoarse.txt 2>&1; echo "PCL_COARSE=0 rc=$? $(tail -1 gpurun_out/knob_estep_nocoarse.txt)"
PCL_COMPACT_MAIN=0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_nocompact.txt 2>&1; echo "PCL_COMPACT_MAIN=0 rc=$? $(tail -1 gpurun_out/knob_estep_nocompact.txt)"
PCL_COARSE_SPLIT_MAX=1.0 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_split1.txt 2>&1; echo "PCL_COARSE_SPLIT_MAX=1.0 rc=$? $(tail -1 gpurun_out/knob_estep_split1.txt)"
PCL_COARSE_PASSES=3 timeout -k 10 200 python3 tests/test_gpu_fuzz_estep.py 150 0 > gpurun_out/knob_estep_np3.txt 2>&1; echo "PCL_COARSE_PASSES=3 rc=$? $(tail -1 gpurun_out/knob_estep_np3.txt)"
PCL_COARSE_PASSES=3 timeout -k 10 200 python3 tools/coarse_fuzz.py 40 500 > gpurun_out/knob_coarse_np3.txt 2>&1; echo "PCL_COARSE_PASSES=3 coarse fuzz rc=$? $(tail -1 gpurun_out/knob_coarse_np3.txt)"
timeout -k 10 300 python3 tests/test_gpu_fuzz_estep.py 100 6000 > gpurun_out/knob_estep_bigM.txt 2>&1; echo "seeds 6000.. rc=$? $(tail -1 gpurun_out/knob_estep_bigM.txt)"
grep -h FAILED gpurun_out/knob_*.txt | head -20

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
bash tools/gpu_profile.sh > gpurun_out/profile_summary.log 2>&1
grep -E "gmm_score|hmm_fb" gpurun_out/profile_summary.log | grep -E "FETCH|WRITE|Calls|calls|kernel_stats|^\"" | head; grep -A6 "kernel_stats.csv" gpurun_out/profile_summary.log | head -12
tail -1 gpurun_out/prof/bench_trace.log | cut -c1-400

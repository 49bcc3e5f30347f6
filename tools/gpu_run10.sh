cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh > gpurun_out/profile_summary.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-300

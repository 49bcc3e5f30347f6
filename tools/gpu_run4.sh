cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh > gpurun_out/profile_summary.log 2>&1
tail -60 gpurun_out/profile_summary.log
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; tail -2 gpurun_out/bench_full.log

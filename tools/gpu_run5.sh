cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -40 gpurun_out/pytest_gpu.log

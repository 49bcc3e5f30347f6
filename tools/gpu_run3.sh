cd $GRAFT_REPO_ROOT
python tools/debug_acc.py > gpurun_out/debug_acc.log 2>&1; cat gpurun_out/debug_acc.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
python tools/score_bench.py 256 2048 50 2>&1 | tail -1

cd $GRAFT_REPO_ROOT
for v in 1 3; do PCL_SCORE_VARIANT=$v CHECK=1 python tools/score_bench.py 256 2048 50 2>&1 | tail -2; done
PCL_SCORE_VARIANT=3 python tools/score_bench.py 1024 2048 1000 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log

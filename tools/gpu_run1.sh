cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.log 2>&1
nproc > gpurun_out/host.log; free -g >> gpurun_out/host.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu && timeout 120 ./gpurun_out/ubench_valu > gpurun_out/ubench.log 2>&1
timeout 300 python bench.py --workload C2 --steps 3 --cpu-baseline 0 > gpurun_out/bench_c2.log 2>&1
timeout 600 python bench.py --steps 2 --cpu-baseline 0 > gpurun_out/bench_c4.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/ubench.log; tail -3 gpurun_out/bench_c2.log; tail -3 gpurun_out/bench_c4.log

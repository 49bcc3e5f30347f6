cd $GRAFT_REPO_ROOT
POCCALA_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload C2 --steps 2 --cpu-baseline 0 > gpurun_out/bench_torchrun1.log 2>&1; echo "exit $?"; tail -8 gpurun_out/bench_torchrun1.log | cut -c1-200

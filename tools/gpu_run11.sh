cd $GRAFT_REPO_ROOT
POCCALA_DEVICE=0 POCCALA_NO_RCCL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --workload C2 --steps 3 --warmup 1 > gpurun_out/bench_2rank_rehearsal.log 2>&1; echo "exit $?"; grep -o '{"metric.*' gpurun_out/bench_2rank_rehearsal.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['utterances_total'], d['extra']['estep_frames_per_s'])"
timeout 600 python bench.py --workload C2 --steps 3 2>&1 | grep -o '{"metric.*' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:60])"

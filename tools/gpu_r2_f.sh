#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
O=gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -30 > $O/tests.log
timeout 600 python3 tools/estep_peaked_bench.py > $O/peaked.log 2>&1
POCCALA_HANG_DUMP=100 POCCALA_SHARE_DEVICE=1 timeout 400 python bench.py --gpus 2 --workload C2 --steps 5 --warmup 1 --cpu-baseline 0 > $O/bench_world2_shared.json 2> $O/bench_world2_shared.err
tail -12 $O/tests.log; cat $O/peaked.log; head -c 400 $O/bench_world2_shared.json; echo; tail -40 $O/bench_world2_shared.err

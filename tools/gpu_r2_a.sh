#!/bin/bash
# round 2, call A: new GPU tests, the whole GPU suite, a short bench, and the N > 1 path rehearsed on one device
mkdir -p gpurun_out/r2a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_units.py -m gpu -q -W ignore -x 2>&1 | tail -40 > $O/units.log
timeout 1500 python -m pytest tests -m gpu -q -W ignore 2>&1 | tail -40 > $O/tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
POCCALA_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --workload C2 --steps 5 --warmup 1 --cpu-baseline 0 > $O/bench_world2_shared.json 2> $O/bench_world2_shared.err
POCCALA_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 1 --cpu-baseline 0 > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
tail -5 $O/units.log; tail -5 $O/tests.log; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err; head -c 300 $O/bench_world2_shared.json; echo; tail -3 $O/bench_world2_shared.err; tail -3 $O/bench_rccl_world1.err

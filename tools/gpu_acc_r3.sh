#!/bin/bash
# round-3 accumulate kernel check on the GPU box: parity at small and real M, then A/B timings of the variant libraries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== parity (small)"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "estep or em_iteration or ill_conditioned or outlier or c4_shard_full or sharded" 2>&1 | tail -8
echo "== parity (real M)"; timeout 900 python -m pytest tests/test_gpu_accumulate.py -q -m gpu -s 2>&1 | tail -12
echo "== units/dropin"; timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_dropin.py tests/test_gpu_decode.py -q -m gpu 2>&1 | tail -8
echo "== flat A/B"; bash tools/gpu_ab_run.sh default $ABV
echo "== peaked"; for n in default $ABV; do if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi; echo -n "$n: "; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/estep_peaked_bench.py 2>&1 | head -1; done
} > gpurun_out/r3_acc.log 2>&1
tail -60 gpurun_out/r3_acc.log

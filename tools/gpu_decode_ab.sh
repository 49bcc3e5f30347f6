#!/bin/bash
# the token-passing kernels: parity tests (both kernels), randomised soak, then the C5 shard timing under every variant library named
# STAMPLIB=name: phase stamps of that library (built with -DPCL_DEC_STAMPS) on 8 utterances alone and on the full shard
cd $GRAFT_REPO_ROOT
O=gpurun_out/dec; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -W ignore 2>&1 | tail -5 | tee $O/tests_lr.log
PCL_DEC_GENERAL=1 timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -W ignore -k "bit_for_bit or properties" 2>&1 | tail -3 | tee $O/tests_general.log
timeout 600 python tools/decode_fuzz.py ${FUZZ:-60} ${SEED:-11} 2>&1 | tail -4 | tee $O/fuzz.log
for n in "$@"; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n"; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/c5_decode_bench.py 417 4096 20000 3 8192 2>&1 | grep -E "resident|streaming" | tee -a $O/ab.log
done
if [ -n "$STAMPLIB" ]; then
  echo "== stamps, 8 utterances"; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$STAMPLIB.so timeout 300 python tools/c5_decode_bench.py 8 256 20000 1 8192 2>&1 | grep -E "stamps" | head -3 | tee -a $O/ab.log
  echo "== stamps, full shard"; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$STAMPLIB.so timeout 300 python tools/c5_decode_bench.py 417 256 20000 1 8192 2>&1 | grep -E "stamps" | head -3 | tee -a $O/ab.log
fi

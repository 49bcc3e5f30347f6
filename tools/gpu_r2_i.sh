#!/bin/bash
# decode kernel: parity tests, then C5 shard timing (resident, streamed) under the default / stamps builds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q -W ignore -x 2>&1 | tail -12 | tee $O/tests.log
for n in default dec_stamps; do
  if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi
  echo "== $n"; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python tools/c5_decode_bench.py 417 4096 20000 ${CH:-3} 8192 2>&1 | grep -v "^tree" | tee $O/c5_$n.log | cut -c1-400 | grep -v "stamps" | tail -8
  grep stamps $O/c5_$n.log | head -2
done

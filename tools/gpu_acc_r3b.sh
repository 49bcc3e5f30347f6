#!/bin/bash
# accumulate A/B: variant libraries, then the image budget (state groups) under the default library, then a kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== flat A/B"; bash tools/gpu_ab_run.sh default $ABV
echo "== peaked"; for n in default $ABV; do if [ $n = default ]; then lib=poccala_amd/libpoccala_hip.so; else lib=build_ab/lib_$n.so; fi; echo -n "$n: "; POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/estep_peaked_bench.py 2>&1 | head -1; done
echo "== image budget (flat)"; for mb in 256 512 1024 4096 8192; do echo -n "PCL_ACC_IMAGE_MB=$mb "; PCL_ACC_IMAGE_MB=$mb timeout 300 python tools/acc_bench.py | tail -1; done
echo "== no overlap"; PCL_ACC_OVERLAP=0 timeout 300 python tools/acc_bench.py | tail -1
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_accumulate.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_acc -- python3 $GRAFT_REPO_ROOT/tools/acc_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== kernel stats"; f=$(find gpurun_out/prof_acc -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-170
find gpurun_out/prof_acc -name "*.csv" -size +1M -delete
} > gpurun_out/r3_accb.log 2>&1
tail -60 gpurun_out/r3_accb.log

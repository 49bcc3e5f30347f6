#!/bin/bash
# where does pcl_destroy crash?  the failing selection under rocgdb (batch mode, backtrace on the fault)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 25" -ex "info threads" --args python3 -m pytest tests/test_gpu_accumulate.py tests/test_gpu_units.py tests/test_gpu_parity.py -m gpu -x -q -k "not c5_ and not c3_ and not c2_" > gpurun_out/r5h_gdb.txt 2>&1
echo "gdb rc=$?"
grep -n "SIGSEGV\|^#[0-9]" gpurun_out/r5h_gdb.txt | head -40
tail -5 gpurun_out/r5h_gdb.txt

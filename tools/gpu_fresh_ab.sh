#!/bin/bash
# round 5: fewer markers on the main queue (A/B in one run), the sweep tests, then the whole GPU suite under rocgdb (a backtrace if pcl_destroy faults again)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r5j_fresh.txt
for m in 1 0 1; do
  echo "== PCL_FEWER_MARKERS=$m" >> gpurun_out/r5j_fresh.txt
  PCL_FEWER_MARKERS=$m timeout -k 10 300 python3 tools/fresh_batch_probe.py C4shard 60 >> gpurun_out/r5j_fresh.txt 2>&1; echo "rc=$?" >> gpurun_out/r5j_fresh.txt
done
cat gpurun_out/r5j_fresh.txt
timeout -k 10 1000 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 30" --args python3 -m pytest tests -m gpu -x -q > gpurun_out/r5j_gdb.txt 2>&1
echo "gdb rc=$?"
grep -n "SIGSEGV\|^#[0-9]\|passed\|failed" gpurun_out/r5j_gdb.txt | head -40

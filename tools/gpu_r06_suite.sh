#!/bin/bash
# the whole GPU suite as the driver runs it, with durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r06_suite.txt 2>&1; echo "suite rc=$?"
tail -60 gpurun_out/r06_suite.txt

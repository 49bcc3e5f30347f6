#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -25 > gpurun_out/tests.log
cat gpurun_out/tests.log

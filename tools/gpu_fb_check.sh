#!/bin/bash
# forward-backward with the table-based two-term log-sum-exp: golden / oracle parity, then timing
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_dropin.py -m gpu -q -W ignore -x -k "not c5 and not c3_deep and not score_variants and not fuzz" 2>&1 | tail -3
timeout 300 python tools/fb_bench.py 2>&1 | tail -6
timeout 300 python bench.py --workload C2 --cpu-baseline 0 --extra 0 --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d['ms_per_step'])"

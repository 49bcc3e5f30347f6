#!/usr/bin/env python3
"""More seeds of the randomised parity tests than the suite runs (scoring fuzz, E-step fuzz, label batches / per-unit
accumulators): usage: parity_soak.py [first seed] [count]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from poccala_amd import Engine
import test_gpu_parity as tp
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = Engine(0)
bad = 0
for seed in range(first, first + count):
    for fn in (tp.test_score_fuzz_default_variant, tp.test_estep_fuzz_default_variant):
        try:
            fn(eng, seed)
        except Exception:
            bad += 1
            print('FAILED %s seed %d' % (fn.__name__, seed)); traceback.print_exc(limit=2)
print('%d seeds x 2 tests, %d failures' % (count, bad))
sys.exit(1 if bad else 0)

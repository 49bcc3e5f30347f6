#!/usr/bin/env python3
"""MFCC front-end throughput: 1024 signals x ~3.8 s at 16 kHz (300 frames each) -> (300, 39) features."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine
from poccala_amd.StatisticalModel.AudioProcessing import mfcc_batch
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
n = 400 + 299 * 200
sigs = [np.round(2000 * rng.standard_normal(n)) for _ in range(U)]
eng = Engine(0); eng.enable_timing(True)
out = mfcc_batch(sigs, 16000, d1=True, d2=True, engine=eng); eng.kernel_time('mfcc')
t0 = time.perf_counter()
out = mfcc_batch(sigs, 16000, d1=True, d2=True, engine=eng)
wall = time.perf_counter() - t0
ms, k = eng.kernel_time('mfcc')
F = sum(len(o) for o in out)
flop = F * 257 * 400 * 4
print('MFCC: %d signals, %d frames: kernels %.2f ms (%.2f TFLOP/s f64 DFT), %.3g frames/s on the device; %.1f ms wall incl. PCIe + host prep (%.3g frames/s)'
      % (U, F, ms, flop / ms / 1e9, F / ms * 1e3, wall * 1e3, F / wall))

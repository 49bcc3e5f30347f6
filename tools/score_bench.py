#!/usr/bin/env python3
"""Score-only timing of the GMM scoring kernel (kernel A/B harness; not the headline bench).
usage: score_bench.py [U] [M] [units]   env POCCALA_HIP_LIB / PCL_SCORE_VARIANT select the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
units = int(sys.argv[3]) if len(sys.argv) > 3 else 50
D, T, L = 39, 300, 20
mean, var, w, trans = synth.make_model(units, M, D)
frames, lens, begin = synth.make_frames(U, T, D)
labels = synth.make_labels(U, L, units)
if os.environ.get('ZERO'):      # power experiment: trivial operands (all-zero frames, identical unit-variance mixtures)
    frames[:] = 0; mean[:] = 0; var[:] = 1
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w)
eng.load_frames(frames)
b, n = make_sentence_batch(eng, labels, lens, begin, trans)
b.score(PCL_F32); eng.sync(); eng.kernel_time('score')
reps = 5
for _ in range(reps):
    b.score(PCL_F32)
ms, k = eng.kernel_time('score')
ms /= k
pairs = U * T * 3 * L
flop = pairs * M * (3 * D + 4)
ref = None
if os.environ.get('CHECK'):
    from oracle import poccala_oracle as po
    B = b.get('B')[0]
    j = labels[0][0] * 3
    ref = po.gmm_point(frames[:T].astype(np.float64), mean[j], var[j], w[j])
    print('max abs err row1:', np.abs(B[1] - ref).max())
print('%s variant=%s U=%d M=%d units=%d: %.3f ms/launch  %.2f TFLOP/s algorithmic (%.1f%% of 157.3)  %.2f Mframes/s score-only'
      % (os.path.basename(os.environ.get('POCCALA_HIP_LIB', 'default')), os.environ.get('PCL_SCORE_VARIANT', '1'), U, M, units, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573, U * T / ms / 1e3))

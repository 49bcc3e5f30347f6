#!/usr/bin/env python3
"""BASELINE config 5 scoring shape on one GPU: all J = 549 states (XIF_tone) x 4096 mixtures for every frame
of a shard of the 1M-frame corpus (1/8 = 417 utterances x 300 frames).  Decode-side scoring throughput only:
the reference's decoder (Decoder.py) is dead code, see DESIGN.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C5shard']
U = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
mean, var, w, _ = synth.make_model(c['units'], c['M'], c['D'])
frames, lens, begin = synth.make_frames(U, c['T'], c['D'])
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_frames(frames)
J = c['units'] * 3
b = eng.batch([J + 2] * U, lens, begin)
b.set_states([np.concatenate([[-1], np.arange(J), [-2]]).astype(np.int32)] * U)
b.score(PCL_F32); eng.sync(); eng.kernel_time('score')
for _ in range(3):
    b.score(PCL_F32)
ms, k = eng.kernel_time('score'); ms /= k
F = int(lens.sum())
flop = F * J * c['M'] * (3 * c['D'] + 4)
print('C5 shard: %d frames x %d states x %d mixtures: %.1f ms/launch, %.1f TFLOP/s algorithmic (%.3f of 157.3), %.3g frames/s decode-scoring per GPU'
      % (F, J, c['M'], ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, F / ms * 1e3))

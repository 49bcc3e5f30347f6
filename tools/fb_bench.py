#!/usr/bin/env python3
"""Forward-backward alone (no scoring beside it): 1024 sentence HMMs of 62 states x 300 frames, random emissions.
env PCL_FB_ONE_WAVE=1 selects the round-1 one-wave kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, synth
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20           # units per utterance: N = 3 L + 2 states (20 -> 62: one wave)
mean, var, w, trans = synth.make_model(50, 4, 13)
labels = synth.make_labels(U, L, 50)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans))
frames, lens, begin = synth.make_frames(U, 300, 13)
eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
b.score(1)
for fix in (False, True):
    for _ in range(60):                        # (a kernel this short, launched after host work, otherwise runs at idle clocks)
        b.forward_backward(fix_pi=fix)
    eng.sync(); eng.kernel_time('fb')
    for _ in range(20):
        b.forward_backward(fix_pi=fix)
    ms, k = eng.kernel_time('fb')
    print('%s N=%d U=%d fix_pi=%s: %.3f ms per launch (%d passes), logP[0] = %.10f' % ('one-wave' if os.environ.get('PCL_FB_ONE_WAVE') else ('log-domain' if os.environ.get('PCL_FB_LINEAR') == '0' else 'scaled-linear'), 3 * L + 2, U, fix, ms / k, b.get('npass')[0], b.get('logp')[0]))

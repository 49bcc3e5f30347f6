#!/usr/bin/env python3
"""Accumulate-only timing (kernel A/B harness): C4 shard, score + forward-backward once, then N accumulate passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch
U, M, units, D, T, L = 1024, 2048, 1000, 39, 300, 20
mean, var, w, trans = synth.make_model(units, M, D)
frames, lens, begin = synth.make_frames(U, T, D)
labels = synth.make_labels(U, L, units)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_frames(frames)
b, n = make_sentence_batch(eng, labels, lens, begin, trans)
b.score(PCL_F32); b.forward_backward(fix_pi=False)
eng.stats_zero(); b.accumulate(PCL_F32); eng.sync(); eng.kernel_time('accumulate')
PASSES = int(os.environ.get('ACC_PASSES', '4')) - 1            # (+ the warm-up pass above: ACC_PASSES passes in the trace)
for _ in range(PASSES):
    eng.stats_zero(); b.accumulate(PCL_F32)
ms, k = eng.kernel_time('accumulate')
st = eng.stats_download()
print('%s: accumulate %.2f ms/pass   sum acc = %.6f (frames x states occupancy), alpha_acc sum = %.6f' % (
    os.path.basename(os.environ.get('POCCALA_HIP_LIB', 'default')), ms / k, st['acc'].sum(), st['alpha_acc'].sum()))

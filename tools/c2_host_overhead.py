#!/usr/bin/env python3
"""C2 (128 utterances, 256-mix): is the score + forward-backward step bound by the host's launch path?  Wall time of the timed loop
against the time the host needs to issue it (the loop returns before the GPU is done), and per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C2']
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'])
frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'])
labels = synth.make_labels(c['U'], c['L'], c['units'])
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
bs = [eng.label_batch(labels, lens, begin) for _ in range(2)]
for b in bs:
    b.score(PCL_F32); b.forward_backward()
eng.sync()
for rep in range(3):
    n = 200
    t0 = time.perf_counter()
    for k in range(n):
        b = bs[k & 1]
        b.score(PCL_F32)
        b.forward_backward()
    t_issue = time.perf_counter() - t0
    eng.sync()
    t_all = time.perf_counter() - t0
    print('steps %d: issued in %.3f ms/step, done in %.3f ms/step' % (n, t_issue / n * 1e3, t_all / n * 1e3))
eng.enable_timing(True)
for k in range(20):
    b = bs[k & 1]; b.score(PCL_F32); b.forward_backward()
eng.sync()
for name in ('score', 'fb'):
    ms, k = eng.kernel_time(name)
    print(name, ms / max(k, 1), k)

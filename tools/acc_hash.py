#!/usr/bin/env python3
"""hash of the E-step statistics and per-unit accumulators of a ragged problem (for A/B runs of accumulate variants: same bits or not)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, xxhash
from poccala_amd import Engine, PCL_F32, synth
units, M, D = 30, 300, 39
mean, var, w, trans = synth.make_model(units, M, D, seed=5)
var[::3, ::7] *= 0.004                                     # tight mixtures: split states too
frames, lens, begin = synth.make_frames(200, 120, D, seed=6, ragged=True)
labels = [list(l) + [int(l[0])] for l in synth.make_labels(200, 4, units, seed=7)]      # a unit named twice: duplicate rows
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
for peaked in (False, True):
    b.score(PCL_F32); b.forward_backward(); eng.stats_zero(); b.accumulate(PCL_F32); b.accumulate_hmm()
    st = eng.stats_download()
    h = xxhash.xxh3_128()
    for k in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
        h.update(np.ascontiguousarray(st[k]).tobytes())
    print('pass', int(peaked), h.hexdigest(), float(st['acc'].sum()))
    eng.em_exchange(1e-3, update_transitions=True); b.refresh_transitions()
    hm = xxhash.xxh3_128()
    for a in eng.model_download() + tuple(eng.model_conditioning()[:1]) + tuple(eng.model_split_info()[:1]):
        hm.update(np.ascontiguousarray(a).tobytes())
    b.score(PCL_F32)
    hm.update(np.ascontiguousarray(np.concatenate([x.ravel() for x in b.get('B')])).tobytes())
    print('model + conditioning + off-pipe counts + next scores after the M-step', hm.hexdigest())

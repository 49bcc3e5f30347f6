#!/usr/bin/env python3
"""E-step timing on features SAMPLED FROM THE MODEL (each frame drawn from a mixture of the state its utterance
passes through), i.e. with the peaked posteriors of real aligned speech, next to the bench's random features
(flat posteriors, nothing pruned).  Shows what the exact posterior-underflow compaction of gmm_accumulate.hip buys."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch

c = dict(synth.CONFIGS['C4shard'])
U = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'])
labels = synth.make_labels(U, c['L'], c['units'])
T, D, L = c['T'], c['D'], c['L']
rng = np.random.default_rng(5)
frames = np.empty((U * T, D), dtype=np.float32)
per = T // (3 * L)                      # frames per emitting state on the sampled path
for u, lab in enumerate(labels):
    states = np.repeat(np.asarray(lab)[:, None] * 3 + np.arange(3)[None, :], per).reshape(-1)[:T]
    states = np.concatenate([states, np.full(T - len(states), states[-1])])
    mix = rng.integers(0, c['M'], size=T)
    frames[u * T:(u + 1) * T] = mean[states, mix] + np.sqrt(var[states, mix]) * rng.standard_normal((T, D))
lens = np.full(U, T, dtype=np.int32); begin = np.arange(U, dtype=np.int64) * T
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w)
for name, fr in (('model-sampled features', frames), ('random features (bench)', synth.make_frames(U, T, D)[0])):
    eng.load_frames(fr)
    b, n = make_sentence_batch(eng, labels, lens, begin, trans)
    for rep in range(3):
        if rep == 1: eng.kernel_time('accumulate')        # the first pass allocates the work lists
        eng.stats_zero(); eng.sync(); t0 = time.perf_counter()
        b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); eng.sync()
        dt = time.perf_counter() - t0
    acc_ms = eng.kernel_time('accumulate')[0] / 2
    lg = b.get('lgamma')
    kept = np.mean([np.mean(l[1:-1] >= -150 * np.log(2)) for l in lg[:64]])
    print('%-26s E-step %.1f ms (%.2f M frames/s), accumulate %.1f ms, surviving (frame,state) pairs %.1f %%'
          % (name, dt * 1e3, U * T / dt / 1e6, acc_ms, 100 * kept))
    b.close()

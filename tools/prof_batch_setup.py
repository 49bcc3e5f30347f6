#!/usr/bin/env python3
"""Host cost of making a resident label batch at the C4 shard's size (1024 utterances x 300 frames, 20 labels): frames upload,
pcl_batch_create_labels, first score (tile lists), first accumulate (work lists)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C4shard']
t0 = time.perf_counter()
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'], seed=1000)
labels = synth.make_labels(c['U'], c['L'], c['units'], seed=2000)
t1 = time.perf_counter(); print('synthetic data %.2f s' % (t1 - t0))
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.sync()
t2 = time.perf_counter(); print('model + units upload (incl. derive) %.3f s' % (t2 - t1))
for rep in range(3):
    ta = time.perf_counter()
    eng.load_frames(frames); eng.sync()
    tb = time.perf_counter()
    b = eng.label_batch(labels, lens, begin); eng.sync()
    tc = time.perf_counter()
    b.score(PCL_F32); eng.sync()
    td = time.perf_counter()
    b.forward_backward(); eng.sync()
    te = time.perf_counter()
    eng.stats_zero(); b.accumulate(PCL_F32); b.accumulate_hmm(); eng.sync()
    tf = time.perf_counter()
    b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); b.accumulate_hmm(); eng.sync()
    tg = time.perf_counter()
    print('rep %d: frames upload %.1f ms, label_batch %.1f ms, first score %.1f ms, first fb %.1f ms, first accumulate %.1f ms, steady E-step (no M-step) %.1f ms'
          % (rep, (tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3, (te - td) * 1e3, (tf - te) * 1e3, (tg - tf) * 1e3))
    b.close()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
b = eng.label_batch(labels, lens, begin); eng.sync()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)

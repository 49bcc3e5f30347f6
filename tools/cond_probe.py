#!/usr/bin/env python3
"""Per-MIXTURE conditioning of the centred expansion after EM iterations on the C4 shard: would routing single mixtures (instead of
whole states) to the direct-form kernel keep most of the work on the matrix pipe?  cond_m = log2e sum_d (mu_md - c_d)^2 / (2 var_md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C4shard']
U = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
frames, lens, begin = synth.make_frames(U, c['T'], c['D'], seed=1000)
labels = synth.make_labels(U, c['L'], c['units'], seed=2000)
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
LOG2E = 1.4426950408889634
for it in range(4):
    m_, v_, w_ = eng.model_download()
    J = m_.shape[0]
    cen = m_.mean(axis=1, keepdims=True)
    cm = LOG2E * ((m_ - cen) ** 2 * (0.5 / v_)).sum(axis=2)          # (J, M)
    live = w_ > 0
    q = np.percentile(cm[live], [10, 50, 90, 99, 100])
    frac = {t: float(np.mean(cm[live] <= t)) for t in (96, 400, 1600, 6400)}
    # precision-weighted centre: does it help?
    pw = (m_ / v_).sum(axis=1, keepdims=True) / (1.0 / v_).sum(axis=1, keepdims=True)
    cm2 = LOG2E * ((m_ - pw) ** 2 * (0.5 / v_)).sum(axis=2)
    print('iteration %d: live mixtures %.1f%%; cond_m percentiles 10/50/90/99/100 = %s; share of live mixtures with cond_m <= 96/400/1600/6400: %s; '
          'states with every mixture <= 96: %d of %d; precision-weighted centre: median %.1f max %.1f; var median %.3g min %.3g'
          % (it, 100 * live.mean(), np.round(q, 1), {k: round(v, 4) for k, v in frac.items()}, int((np.where(live, cm, 0).max(axis=1) <= 96).sum()), J,
             np.median(cm2[live]), cm2[live].max(), np.median(v_[live]), v_.min()), flush=True)
    eng.stats_zero(); b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); b.accumulate_hmm()
    eng.em_exchange(1e-3, update_transitions=True)
    b.refresh_transitions()

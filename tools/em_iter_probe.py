#!/usr/bin/env python3
"""What an EM iteration does to the kernels' time on the C4 shard: conditioning of the re-estimated model (states above the
threshold leave the matrix pipe), scoring / forward-backward / accumulate times per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C4shard']
U = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
C_COV = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3          # the reference's driver passes 1e-6 (init.py:30 -> Controller.py:151)
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
frames, lens, begin = synth.make_frames(U, c['T'], c['D'], seed=1000)
labels = synth.make_labels(U, c['L'], c['units'], seed=2000)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
names = ('score', 'score_coarse', 'score_subset_fixup', 'derive_coarse', 'score_subset', 'score_direct', 'score_fixup', 'fb', 'accumulate', 'acc_consume', 'acc_subset', 'mstep', 'derive')
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    cond, cmax = eng.model_conditioning()
    n_off, lim = eng.model_split_info()
    m_, v_, w_ = eng.model_download()
    eng.sync()
    for k in names: eng.kernel_time(k)
    eng.stats_zero(); b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); b.accumulate_hmm(); eng.sync()      # (lazy buffers, tile lists, clocks)
    for k in names: eng.kernel_time(k)
    t0 = time.perf_counter()
    eng.stats_zero(); b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); b.accumulate_hmm(); eng.sync()
    t1 = time.perf_counter()
    kt = {k: round(eng.kernel_time(k)[0], 3) for k in names}
    lp = b.get('logp'); st = eng.stats_download(moments=False)
    if os.environ.get('PCL_COARSE_STATS'):
        kt['exact_pairs_per_pass'] = eng.coarse_pairs() // 2      # (two E-steps since the last reset)
    import xxhash
    hb = xxhash.xxh3_64(np.ascontiguousarray(np.concatenate([x.ravel() for x in b.get('B')[::32]])).tobytes()).hexdigest()
    hs = xxhash.xxh3_64(np.ascontiguousarray(st['acc']).tobytes()).hexdigest()
    print('iteration %d: off-pipe mixtures %.1f%% (limit %d per state: %d split states, %d whole states off); cond max %.1f, states above %.0f: %d of %d; var min %.3g (floored %.2f%%), weights == 0: %.2f%%; E-step %.1f ms %s; mean logP %.2f; zero-occupancy mixtures %.2f%%; hash(B) %s hash(acc) %s'
          % (it, 100.0 * n_off.sum() / (len(n_off) * c['M']), lim, int(((n_off > 0) & (n_off <= lim)).sum()), int((n_off > lim).sum()), cond.max(), cmax, int((cond > cmax).sum()), len(cond), v_.min(), 100 * np.mean(v_ <= C_COV * 1.0000001), 100 * np.mean(w_ == 0), (t1 - t0) * 1e3, kt, lp.mean(), 100 * np.mean(st['acc'] == 0), hb, hs), flush=True)
    eng.em_exchange(C_COV, update_transitions=True)
    b.refresh_transitions()

#!/usr/bin/env python3
"""Where the coarse pass's time goes (round 6): `make N` leaves the C4 shard's model after N EM iterations under /tmp (the GPU box's own
disk); `time` loads it and times the scoring launches -- run under POCCALA_HIP_LIB=build_ab/lib_cexpK.so (tools/build_variant.sh with
-DPCL_COARSE_EXP=K: 1 = no direct-form evaluation, 2 = no matrix-pipe products, 4 = nothing passes) the differences are the parts.
Those variants compute wrong likelihoods on purpose: timing only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
c = synth.CONFIGS['C4shard']
mode = sys.argv[1]
frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'], seed=1000)
labels = synth.make_labels(c['U'], c['L'], c['units'], seed=2000)
eng = Engine(0); eng.enable_timing(True)
if mode == 'make':
    mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
    eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
    b = eng.label_batch(labels, lens, begin)
    for it in range(int(sys.argv[2])):
        eng.stats_zero(); b.score(PCL_F32); b.forward_backward(); b.accumulate(PCL_F32); b.accumulate_hmm()
        eng.em_exchange(1e-6, update_transitions=True); b.refresh_transitions()
    m_, v_, w_ = eng.model_download()
    np.save('/tmp/em_model_mean.npy', m_); np.save('/tmp/em_model_var.npy', v_); np.save('/tmp/em_model_w.npy', w_)
    np.save('/tmp/em_model_trans.npy', np.stack(trans))
    print('model after %d iterations saved' % int(sys.argv[2]))
else:
    eng.load_model(np.load('/tmp/em_model_mean.npy'), np.load('/tmp/em_model_var.npy'), np.load('/tmp/em_model_w.npy'))
    eng.load_units(np.load('/tmp/em_model_trans.npy')); eng.load_frames(frames)
    b = eng.label_batch(labels, lens, begin)
    names = ('score', 'score_coarse', 'score_subset', 'score_direct', 'score_fixup')
    b.score(PCL_F32); eng.sync()
    for k in names: eng.kernel_time(k)
    for rep in range(3): b.score(PCL_F32)
    eng.sync()
    n_off, lim = eng.model_split_info()
    print(os.path.basename(os.environ.get('POCCALA_HIP_LIB', 'regular')), 'PASSES=' + os.environ.get('PCL_COARSE_PASSES', '1'),
          'off-pipe %.1f%%' % (100.0 * n_off.sum() / (len(n_off) * c['M'])), {k: round(eng.kernel_time(k)[0] / 3, 3) for k in names})
b.close(); eng.close()

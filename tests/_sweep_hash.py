"""Helper of tests/test_gpu_sweep.py (run as a subprocess under different environment knobs): a short corpus sweep -- batches made and
dropped on the way, one statistics block, the exchange (one rank: the M-step), a second E-step on the new model -- and one line of hashes
of everything it produced.  The knobs only move WHERE and WHEN things are queued: the line must not change."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xxhash  # noqa: E402
from poccala_amd import Engine, PCL_F32, synth  # noqa: E402

units, M, D, U, T, L = 10, 64, 39, 40, 90, 5
mean, var, w, trans = synth.make_model(units, M, D, seed=901)
trans = [synth.random_left_right_transmat(np.random.default_rng(903 + u)) for u in range(units)]      # every unit its own matrix from the start
frames, lens, begin = synth.make_frames(2 * U, T, D, seed=902, ragged=True)
eng = Engine(0)
eng.load_model(mean, var, w)
eng.load_units(np.stack(trans))
eng.load_frames(frames)
h = xxhash.xxh3_128()
DBG = bool(os.environ.get('SWEEP_DEBUG'))


def dbg(tag, a):
    if DBG:
        print('  ', tag, xxhash.xxh3_64(np.ascontiguousarray(a).tobytes()).hexdigest())

for it in range(2):
    eng.stats_zero()
    live = []
    for k in range(6):
        half = k % 2
        labels = synth.make_labels(U, L, units, seed=910 + k)
        b = eng.label_batch(labels, lens[U * half:U * (half + 1)], begin[U * half:U * (half + 1)])
        b.score(PCL_F32)
        b.forward_backward(fix_pi=False)
        b.accumulate(PCL_F32)
        b.accumulate_hmm()
        if DBG and os.environ.get('SWEEP_B'):
            dbg('it%d B[%d]' % (it, k), np.concatenate([x.ravel() for x in b.get('B')]))
            dbg('it%d lgamma[%d]' % (it, k), np.concatenate([x.ravel() for x in b.get('lgamma')]))
        res = b.result_buffers(('logp',), slot=k % 3)
        b.fetch_async(res)
        live.append((b, res))
        if len(live) > 2:
            old, r = live.pop(0)
            old.fetch_wait()
            h.update(np.ascontiguousarray(r['logp']).tobytes())
            dbg('it%d logp' % it, r['logp'])
            old.close()
    for b, r in live:
        b.fetch_wait()
        h.update(np.ascontiguousarray(r['logp']).tobytes())
        dbg('it%d logp(tail)' % it, r['logp'])
        b.close()
    st = eng.stats_download()
    for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
        h.update(np.ascontiguousarray(st[key]).tobytes())
        dbg('it%d %s' % (it, key), st[key])
    ks, ga = eng.hmm_acc_download()
    h.update(ks.tobytes()); h.update(ga.tobytes())
    dbg('it%d ksai_acc' % it, ks); dbg('it%d gamma_acc' % it, ga)
    eng.em_exchange(1e-3, update_transitions=True)
    for a in eng.model_download():
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(eng.units_download().tobytes())
eng.close()
print('SWEEPHASH', h.hexdigest())

#!/usr/bin/env python3
"""Round 5: the forward-backward results of a batch were seen to differ from run to run in the second EM iteration of a sweep (same
emissions, same transitions).  After one EM iteration on a small model: the same label batch created / scored / run 60 times in one
process -- fresh batch each time, and one resident batch re-run -- every result compared with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
units, M, D, U, T, L = 10, 64, 39, 40, 90, 5
mean, var, w, trans = synth.make_model(units, M, D, seed=901)
frames, lens, begin = synth.make_frames(2 * U, T, D, seed=902, ragged=True)
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
mk = lambda k: eng.label_batch(synth.make_labels(U, L, units, seed=910 + k), lens[U * (k % 2):U * (k % 2 + 1)], begin[U * (k % 2):U * (k % 2 + 1)])
if not os.environ.get('SKIP_EM'):
    eng.stats_zero()
    for k in range(6):
        b = mk(k); b.score(PCL_F32); b.forward_backward(fix_pi=False); b.accumulate(PCL_F32); b.accumulate_hmm(); b.close()
    eng.em_exchange(1e-3, update_transitions=True)
print('unit transitions after the M-step: min nonzero %.3g, zeros %d of %d' % (eng.units_download()[eng.units_download() > 0].min(), int((eng.units_download() == 0).sum()), eng.units_download().size))

def run(b):
    b.score(PCL_F32); b.forward_backward(fix_pi=False)
    return dict(B=np.concatenate([x.ravel() for x in b.get('B')]), lg=np.concatenate([x.ravel() for x in b.get('lgamma')]), logp=b.get('logp'),
                npass=b.get('npass'), alpha=np.concatenate([x.ravel() for x in b.get('alpha')]), beta=np.concatenate([x.ravel() for x in b.get('beta')]),
                q=b.get('qtrace'))

def cmp(tag, ref, got, n_off):
    bad = {}
    for key in ref:
        a, c = ref[key], got[key]
        if not np.array_equal(a, c, equal_nan=True):
            fin = np.isfinite(a) & np.isfinite(c)
            d = np.abs(a[fin] - c[fin])
            bad[key] = (int((a != c).sum()), float(d.max()) if d.size else None)
    if bad:
        lo = np.flatnonzero(ref['logp'] != got['logp'])
        print('  %s: differs: %s; utterances with another ln P(O): %s; npass ref %s got %s' % (tag, bad, lo[:8], ref['npass'][lo[:4]], got['npass'][lo[:4]]))
    return bool(bad)

for k in (1, 2):
    b0 = mk(k); ref = run(b0)
    nbad = 0
    for i in range(60):
        b = mk(k); got = run(b); b.close()
        nbad += cmp('fresh batch %d, labels %d' % (i, k), ref, got, 0)
    print('labels %d: %d of 60 fresh batches differ from the first' % (k, nbad))
    nbad = 0
    for i in range(60):
        nbad += cmp('resident re-run %d, labels %d' % (i, k), ref, run(b0), 0)
    print('labels %d: %d of 60 re-runs of the resident batch differ' % (k, nbad))
    b0.close()
eng.close()

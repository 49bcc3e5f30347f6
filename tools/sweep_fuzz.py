#!/usr/bin/env python3
"""Randomised corpus sweep against an unhurried twin (round 5).  Engine A runs a sweep the way a loader would: label batches of random
shape (utterances, label length, ragged frames) made on the fly, a random choice of score / forward-backward (free or locked pi) / Viterbi /
accumulate / results on their way, dropped at a random distance behind (sometimes with their work still in flight), now and then an M-step
(GMM and transitions) in between.  Engine B does the same batches one at a time with a device sync around every step.  Every result of
A -- ln P(O), ln gamma, Viterbi paths, pass counts, the statistics block, the per-unit accumulators, the re-estimated model -- must
equal B's bit for bit.   usage: sweep_fuzz.py [seeds] [steps per seed] [first seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, PCL_F64, synth


def run_seed(seed, steps, verbose=False):
    rng = np.random.default_rng(seed)
    units = int(rng.integers(3, 14))
    M = int(rng.choice([8, 33, 64, 96]))
    D = int(rng.choice([13, 26, 39]))
    mean, var, w, _ = synth.make_model(units, M, D, seed=seed)
    trans = [synth.random_left_right_transmat(rng) for _ in range(units)]
    F = 6000
    frames = rng.standard_normal((F, D)).astype(np.float32)
    A, B = Engine(0), Engine(0)
    for e in (A, B):
        e.load_model(mean, var, w)
        e.load_units(np.stack(trans))
        e.load_frames(frames)
        e.stats_zero()
    P = PCL_F32
    live = []                                   # (batch on A, expected results from B, pending fetch buffers)
    bad = []

    def check(tag, got, want):
        for k in want:
            if k in got and not np.array_equal(got[k], want[k], equal_nan=True):
                bad.append('%s: %s differs (seed %d)' % (tag, k, seed))

    def collect(b, ops):
        out = {}
        if 'fb' in ops:
            out['logp'] = b.get('logp')
            out['npass'] = b.get('npass')
            out['lgamma'] = np.concatenate([x.ravel() for x in b.get('lgamma')])
        if 'vit' in ops:
            out['path'] = np.concatenate(b.get('path'))
            out['point'] = b.get('point')
        return out

    for step in range(steps):
        U = int(rng.integers(1, 50))
        L = int(rng.integers(1, 9)) if rng.random() < 0.85 else int(rng.integers(20, 30))      # now and then sentence HMMs of more than 64 states
        lens = rng.integers(1 if rng.random() < 0.1 else 8, 60, size=U).astype(np.int32)         # (one-frame utterances too)
        begin = rng.integers(0, F - 60, size=U).astype(np.int64)          # utterances may overlap in the frame matrix: they only read it
        labels = [rng.integers(0, units, size=L) for _ in range(U)]
        if rng.random() < 0.5:
            labels_a = np.stack(labels).astype(np.int32)                  # the (U, L) fast path
        else:
            labels_a = labels
        ops = {'score'}
        if rng.random() < 0.85:
            ops.add('fb')
        if rng.random() < 0.5:
            ops.add('vit')
        if 'fb' in ops and rng.random() < 0.6:
            ops.add('acc')
        fix_pi = bool(rng.random() < 0.3)
        P = PCL_F64 if rng.random() < 0.15 else PCL_F32                                          # the float64 parity mode now and then
        # ---- B: unhurried
        B.sync()
        bb = B.label_batch(labels, lens, begin)
        bb.score(P); B.sync()
        if 'fb' in ops:
            bb.forward_backward(fix_pi=fix_pi); B.sync()
        if 'vit' in ops:
            bb.viterbi(); B.sync()
        if 'acc' in ops:
            bb.accumulate(P); bb.accumulate_hmm(); B.sync()
        want = collect(bb, ops)
        bb.close(); B.sync()
        # ---- A: on the fly
        ba = A.label_batch(labels_a, lens, begin)
        ba.score(P)
        order = [o for o in ('fb', 'vit') if o in ops]
        if len(order) == 2 and rng.random() < 0.5:
            order.reverse()
        for o in order:
            if o == 'fb':
                ba.forward_backward(fix_pi=fix_pi)
            else:
                ba.viterbi()
        if 'acc' in ops:
            ba.accumulate(P); ba.accumulate_hmm()
        bufs = None
        if 'fb' in ops and rng.random() < 0.5:
            want_names = ['logp', 'lgamma'] + (['path', 'point'] if 'vit' in ops else [])
            bufs = ba.result_buffers(tuple(want_names), slot=step % 6)
            ba.fetch_async(bufs)
        live.append((ba, want, bufs, ops, step))
        # drop some
        while len(live) > int(rng.integers(1, 5)):
            b, wnt, bf, op, st = live.pop(0)
            mode = rng.random()
            if mode < 0.25:
                b.close()                                             # with its work (possibly) still in flight, nothing read
                continue
            if bf is not None:
                b.fetch_wait()
                got = dict(logp=bf['logp'].copy(), lgamma=np.concatenate([v.ravel() for v in b.lgamma_views(bf['lgamma'])]))
                if 'path' in bf:
                    got['path'] = bf['path'].copy(); got['point'] = bf['point'].copy()
                check('step %d (fetched)' % st, got, wnt)
            check('step %d' % st, collect(b, op), wnt)
            b.close()
        # an EM iteration now and then (both engines: same statistics in the same order -> same model)
        if rng.random() < 0.12:
            for b, wnt, bf, op, st in live:
                if bf is not None:
                    b.fetch_wait()
                check('step %d (before the M-step)' % st, collect(b, op), wnt)
                b.close()
            live = []
            sa, sb = A.stats_download(), B.stats_download()
            for k in sa:
                if not np.array_equal(sa[k], sb[k]):
                    bad.append('statistics %s differ before an M-step at step %d (seed %d)' % (k, step, seed))
            c_cov = float(rng.choice([1e-3, 1e-6]))
            A.em_exchange(c_cov, update_transitions=True)
            B.em_exchange(c_cov, update_transitions=True)
            if rng.random() < 0.3:                                   # a model upload in mid-sweep (the statistics block and the scratch are rebuilt)
                mm, vv, ww = B.model_download()
                A.load_model(mm, vv, ww); B.load_model(mm, vv, ww)
            for x, y, nm in zip(A.model_download() + (A.units_download(),), B.model_download() + (B.units_download(),), ('mean', 'var', 'weight', 'transitions')):
                if not np.array_equal(x, y):
                    bad.append('re-estimated %s differ at step %d (seed %d)' % (nm, step, seed))
            A.stats_zero(); B.stats_zero()
    for b, wnt, bf, op, st in live:
        if bf is not None:
            b.fetch_wait()
        check('step %d (end)' % st, collect(b, op), wnt)
        b.close()
    sa, sb = A.stats_download(), B.stats_download()
    for k in sa:
        if not np.array_equal(sa[k], sb[k]):
            bad.append('final statistics %s differ (seed %d)' % (k, seed))
    ha, hb = A.hmm_acc_download(), B.hmm_acc_download()
    if not (np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1])):
        bad.append('per-unit accumulators differ (seed %d)' % seed)
    A.close(); B.close()
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    total = []
    for s in range(first, first + n):
        bad = run_seed(s, steps)
        total += bad
        print('seed %d: %s' % (s, 'ok' if not bad else '; '.join(bad[:6])), flush=True)
    print('%d seeds x %d steps, %d mismatches' % (n, steps, len(total)))
    sys.exit(1 if total else 0)

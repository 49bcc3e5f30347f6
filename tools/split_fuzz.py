#!/usr/bin/env python3
"""Randomised check of split states: random models with a random share of tight mixtures per state (none, a few, more than
one accumulate slice of 256, just below / above the limit), zero-weight tight mixtures, outlier frames (flagged tiles: the fix-up and the
subset launch meet), any feature dimension.  Scores against the float64 oracle with the suite's f32 allowance; E-step statistics against a
context that never splits (PCL_SPLIT_MAX=0: whole states in direct form) with the f32 contract (1e-4 relative + 1e-6 of the largest).
usage: split_fuzz.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch
import test_gpu_parity as tp
from oracle import poccala_oracle as po

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = Engine(0)
os.environ['PCL_SPLIT_MAX'] = '0'
ref_eng = Engine(0)
del os.environ['PCL_SPLIT_MAX']
bad = 0
S = 5
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.choice([13, 26, 39, 20, 33, 45]))
    units = int(rng.integers(2, 5))
    M = int(rng.choice([7, 40, 64, 130, 300, 700]))
    mean, var, w, trans = synth.make_model(units, M, D, seed=seed)
    J = mean.shape[0]
    share = rng.choice([0.0, 0.03, 0.2, 0.38, 0.45, 0.9], J)
    tight = rng.random((J, M)) < share[:, None]
    var[tight] = rng.uniform(1e-3, 2e-2, (int(tight.sum()), D))
    if tight.any() and seed % 3 == 0:                      # a tight mixture with weight 0
        j, m = np.argwhere(tight)[0]
        w[j, m] = 0.0
        w[j] /= w[j].sum()
    U, L, PER = int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(3, 12))
    labels = [list(rng.integers(0, units, L)) for _ in range(U)]
    TU = L * (S - 2) * PER
    lens = np.full(U, TU, dtype=np.int64)
    begin = np.arange(U, dtype=np.int64) * TU
    st = np.concatenate([np.repeat([unit * (S - 2) + k for unit in lab for k in range(S - 2)], PER) for lab in labels])
    comp = rng.integers(0, M, len(st))
    for i in range(0, len(st), 2):                         # every other frame sits on a tight mixture of its state, if there is one
        t = np.flatnonzero(tight[st[i]] & (w[st[i]] > 0))
        if len(t):
            comp[i] = rng.choice(t)
    x = (mean[st, comp] + np.sqrt(var[st, comp]) * rng.standard_normal((len(st), D))).astype(np.float32)
    if seed % 2:
        x[rng.integers(0, len(x)), rng.integers(0, D)] = 3.0e4      # an outlier: its tile is flagged and rescored in direct form
    try:
        out = []
        for e in (eng, ref_eng):
            e.load_model(mean, var, w)
            e.load_frames(x)
            b, n = make_sentence_batch(e, labels, lens, begin, trans)
            b.score(PCL_F32)
            b.forward_backward(fix_pi=False)
            e.stats_zero()
            b.accumulate(PCL_F32)
            out.append((b.get('B'), b.get('logp'), e.stats_download()))
            b.close()
        n_off, limit = eng.model_split_info()
        # scores of utterance 0 against the oracle
        lab = labels[0]
        xx = x[:TU].astype(np.float64)
        rows = [unit * (S - 2) + k for unit in lab for k in range(S - 2)]
        with np.errstate(divide='ignore'):
            ref = np.stack([po.gmm_point(xx, mean[j], var[j], w[j]) for j in rows])
        bound = tp.f32_evaluation_bound(mean[rows], var[rows], w[rows], x[:TU])
        got = out[0][0][0][1:-1]
        tp.assert_f32_class(got, ref, bound, what='seed %d D=%d M=%d off-pipe %s (limit %d):' % (seed, D, M, n_off.tolist(), limit))
        np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-4)
        for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
            a, r = out[0][2][key], out[1][2][key]
            scale = np.abs(r).max()
            at = tp.cov_acc_atol(out[1][2]['acc'], mean, var, scale * 2e-6) if key == 'cov_acc' else scale * 2e-6
            err = np.abs(a - r) - (1e-4 * np.abs(r) + at)
            assert (err <= 0).all(), (key, float(err.max()), float(scale))
    except Exception:
        import traceback
        bad += 1
        print('FAILED seed %d (D=%d M=%d units=%d)' % (seed, D, M, units)); traceback.print_exc(limit=3)
print('%d seeds, %d failures' % (count, bad))
sys.exit(1 if bad else 0)

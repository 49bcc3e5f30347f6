#!/usr/bin/env python3
"""What bounds the error of cov_acc (and mean_acc) of the f32-class accumulate pass?  The pass forms RAW moments about the state's
centre c_j on the matrix pipe, S2 = sum g x'^2, S1 = sum g x', S0 = sum g (x' = x - c_j), and shifts them to the mixture's mean in
float64: cov = S2 - 2 d S1 + d^2 S0, d = mu - c_j.  Each raw moment carries a relative error eta (two-piece f16 operands, f32
accumulation), so |d cov_acc[m,k]| ~ eta * acc[m] * (d_k^2 + var_k) -- an ABSOLUTE error that a bound relative to cov_acc[m,k]
itself cannot describe when the mixture sits far from the centre in units of its own width.  This tool runs the suite's randomised
E-step case (tests/test_gpu_parity.py:_estep_fuzz) for many seeds and prints the largest
    kappa = (|d cov_acc| - 1e-4 |cov_acc|)+ / (acc (d^2 + var))      and the same for mean_acc against acc (|d| + sd)
usage: cov_error_probe.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import poccala_oracle as po
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch
S = 5
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
eng = Engine(0)
worst_c = worst_m = worst_used = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(500 + seed)
    D = int(rng.choice([13, 26, 39]))
    units, M, U, L, PER = int(rng.integers(2, 6)), int(rng.integers(2, 70)), int(rng.integers(2, 7)), int(rng.integers(1, 4)), int(rng.integers(2, 6))
    mean, var, w, trans = synth.make_model(units, M, D, seed=seed)
    mean = mean * 2.0
    labels = [list(rng.integers(0, units, L)) for _ in range(U)]
    TU = L * (S - 2) * PER
    lens = np.full(U, TU, dtype=np.int64)
    begin = np.arange(U, dtype=np.int64) * TU
    st = np.concatenate([np.repeat([unit * (S - 2) + k for unit in lab for k in range(S - 2)], PER) for lab in labels])
    comp = rng.integers(0, M, len(st))
    x = (mean[st, comp] + np.sqrt(var[st, comp]) * rng.standard_normal((len(st), D))).astype(np.float32)
    eng.load_model(mean, var, w)
    eng.load_frames(x)
    b, n = make_sentence_batch(eng, labels, lens, begin, trans)
    b.score(PCL_F32); b.forward_backward(fix_pi=False); eng.stats_zero(); b.accumulate(PCL_F32)
    got = eng.stats_download()
    cond, cmax = eng.model_conditioning()
    b.close()
    model = {u: dict(trans=trans[u], gmms=[(mean[u * 3 + k], var[u * 3 + k], w[u * 3 + k]) for k in range(3)]) for u in range(units)}
    J = mean.shape[0]
    ref = dict(acc=np.zeros((J, M)), mean_acc=np.zeros((J, M, D)), cov_acc=np.zeros((J, M, D)))
    for u, lab in enumerate(labels):
        xx = x[begin[u]:begin[u] + lens[u]].astype(np.float64)
        _, accs, _ = po.estep_utterance(xx, list(lab), model)
        for pos, unit in enumerate(lab):
            for k in range(S - 2):
                a = accs[pos].gmm[k]
                for key in ref:
                    ref[key][unit * (S - 2) + k] += np.exp(a[key])
    c = mean.mean(axis=1, keepdims=True).astype(np.float32).astype(np.float64)       # the state's expansion centre (model_derive.hip)
    d2 = (mean - c) ** 2
    acc = ref['acc'][:, :, None]
    ok = (acc > 1e-12) & (cond[:, None, None] <= cmax)
    ec = np.maximum(np.abs(got['cov_acc'] - ref['cov_acc']) - 1e-4 * np.abs(ref['cov_acc']), 0.0)
    kc = float((ec / np.where(ok, acc * (d2 + var), np.inf)).max())
    em = np.maximum(np.abs(got['mean_acc'] - ref['mean_acc']) - 1e-4 * np.abs(ref['mean_acc']), 0.0)
    km = float((em / np.where(ok, acc * (np.sqrt(d2) + np.sqrt(var)), np.inf)).max())
    scale = np.abs(ref['cov_acc']).max()
    used = float((np.abs(got['cov_acc'] - ref['cov_acc']) / (1e-6 * scale + 1e-4 * np.abs(ref['cov_acc']))).max())
    worst_c, worst_m, worst_used = max(worst_c, kc), max(worst_m, km), max(worst_used, used)
    if kc > 5e-7 or used > 1.0:
        print('seed %d: D=%d M=%d cond max %.1f: kappa_cov %.2e kappa_mean %.2e; used of (1e-4, 1e-6 max) %.2f' % (seed, D, M, cond.max(), kc, km, used))
print('%d seeds: worst kappa_cov %.2e, kappa_mean %.2e, worst fraction of the plain (1e-4, 1e-6 max) bound %.2f' % (count, worst_c, worst_m, worst_used))

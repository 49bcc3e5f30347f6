#!/usr/bin/env python3
"""VERDICT r3 next #8, the bounded experiment, decided on the CPU before any kernel work: how many 32-mixture x 32-frame tiles of
the scoring kernel could a coarse first pass skip?

The kernel (csrc/gmm_score_split.hip) spends 15 MFMAs per tile: a1 x1 (5), a1 x2 (5), a2 x1 (5).  The proposal: after the
a1 x1 pass alone, skip the other 10 when EVERY value of the tile sits more than 40 (log2) below its lane's running reference
(a true earlier maximum of that frame), so that the skipped mass stays below 2^-24 of the sum even if all 2048 mixtures were
skipped at the threshold (2048 * 2^-40 = 2^-29).  This script evaluates the exact exponents in float64 (an optimistic stand-in
for the coarse pass: no margin for its own error) on the bench's corpora and counts skippable tiles:
  flat    features ~ N(0,1) against the synthetic model (the headline's data)
  peaked  features sampled from the model along the label (aligned speech), scored against the state they were drawn from and
          against other states of the utterance
Keep rule of the verdict: >= 5 % of the tiles on flat data.  Prints the rates; profiles/r04_score_skip.txt holds the output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import synth

LOG2E = 1.4426950408889634
M, D, J_SAMPLE, FRAMES = 2048, 39, 12, 512
mean, var, w, _ = synth.make_model(40, M, D, seed=1)
rng = np.random.default_rng(0)


def exponents(x, j):
    """log2 of w_m N(x; mu_m, var_m) with the reference's constant (util.py:29), (frames, M)."""
    d2 = ((x[:, None, :] - mean[j][None]) ** 2 / var[j][None]).sum(-1)
    return LOG2E * (np.log(w[j])[None] - 0.5 * D * np.log(2 * np.pi) - 0.5 * var[j].sum(-1)[None] - 0.5 * d2)


def tiles(v, thresholds=(40.0, 30.0, 24.0)):
    """v (frames, M): mixture tiles of 32 in order; lane reference = running maximum over the tiles before; frame tiles of 32."""
    F = v.shape[0] // 32 * 32
    v = v[:F].reshape(F // 32, 32, M // 32, 32)                       # (frame tile, frame, mixture tile, mixture)
    tmax = v.max(-1)                                                  # (ft, f, mt)
    ref = np.maximum.accumulate(tmax, axis=2)
    ref_before = np.concatenate([np.full_like(ref[:, :, :1], -np.inf), ref[:, :, :-1]], axis=2)
    gap = (ref_before - tmax).min(axis=1)                             # (ft, mt): the smallest distance below the reference over the tile's 32 frames
    final = ref[:, :, -1:]
    out = {}
    for t in thresholds:
        skip = gap > t
        mass = np.where(skip[:, None, :, None], np.exp2(v - final[..., None]), 0.0).sum((2, 3))     # skipped mass per frame, relative to 2^max
        tot = np.exp2(v - final[..., None]).sum((2, 3))
        out[t] = (float(skip.mean()), float((mass / tot).max()))
    # per-entry: how many single (frame, mixture) values sit below the threshold -- what a per-lane skip could reach at best
    ent = {t: float(((final[..., None] - v) > t).mean()) for t in thresholds}
    return out, ent


def report(name, vs):
    agg, ent = {}, {}
    for v in vs:
        o, e = tiles(v)
        for t in o:
            agg.setdefault(t, []).append(o[t])
            ent.setdefault(t, []).append(e[t])
    for t in sorted(agg, reverse=True):
        print('%-8s threshold 2^-%d: skippable tiles %.3f %% (worst skipped mass %.1e of a frame\'s sum); single entries below the threshold %.1f %%'
              % (name, int(t), 100 * np.mean([a[0] for a in agg[t]]), max(a[1] for a in agg[t]), 100 * np.mean(ent[t])))


flat = [exponents(rng.standard_normal((FRAMES, D)), j) for j in range(J_SAMPLE)]
report('flat', flat)
own, other = [], []
for j in range(J_SAMPLE):
    mix = rng.integers(0, M, FRAMES)
    x = mean[j, mix] + np.sqrt(var[j, mix]) * rng.standard_normal((FRAMES, D))
    own.append(exponents(x, j))
    other.append(exponents(x, (j + 7) % 40))
report('peaked/own', own)
report('peaked/other', other)
print('decision: the rule asks for >= 5 % of the tiles on flat data; a tile can be skipped only when all 32 frames x 32 mixtures are far below their references at once')

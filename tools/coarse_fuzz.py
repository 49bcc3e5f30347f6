#!/usr/bin/env python3
"""Randomised soak of the coarse pass over off-pipe mixtures (csrc/gmm_score_coarse.hip) against the oracle:

    python tools/coarse_fuzz.py [count] [first seed]

Every case draws a model (2-8 states, 32-512 mixtures -- one case in eight 2048 --, D in {13, 26, 39}; per state a random share 0-100 %
of mixtures with variances log-uniform in [1e-6, 0.2]; zero weights), frames (on tight means exactly, near them, from the broad
mixtures, noise; in one case of four a block carrying an offset that leaves the f16 range: the flag + direct-form fallback), and a
context (the default split limit, or PCL_COARSE_SPLIT_MAX=1 so that states WITHOUT an on-pipe mixture go through the pass: threshold
from -inf).  ln b of every (frame, state) against oracle.gmm_point within 5e-5 + 5e-6 |ln b| + the analytical f32 input bound; the count
of exactly evaluated pairs is reported.  Exit code 1 on any failure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import poccala_oracle as po
import test_gpu_coarse as tc

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
worst = 0.0
tot_exact = tot_pairs = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    J = int(rng.integers(2, 9))
    M = 2048 if seed % 8 == 7 else int(rng.choice([32, 48, 64, 96, 160, 256, 512]))
    D = int(rng.choice([13, 26, 39]))
    if M == 2048:
        J = 2
    all_tight = bool(seed % 2)
    shares = [float(rng.choice([0.0, rng.uniform(0.02, 0.3), rng.uniform(0.3, 0.94), 1.0 if all_tight else 0.9])) for _ in range(J)]
    mean, var, w, tight = tc.tight_model(77000 + seed, J, M, D, shares)
    per = int(rng.choice([64, 150, 300])) if M < 2048 else 96
    x, own = tc.frames_for(rng, mean, var, tight, per)
    if seed % 4 == 2:
        lo = int(rng.integers(0, max(1, len(x) - 40)))
        x[lo:lo + 40] += float(rng.choice([30.0, 60.0, -45.0]))
    env = dict(PCL_COARSE_STATS=1)
    if all_tight:
        env['PCL_COARSE_SPLIT_MAX'] = 1.0
    eng = tc._engine(**env)
    try:
        got = tc.score_all(eng, mean, var, w, x)
        n_off, limit = eng.model_split_info()
        exact = eng.coarse_pairs()
    finally:
        eng.close()
    with np.errstate(divide='ignore'):
        ref = np.stack([po.gmm_point(x.astype(np.float64), mean[j], var[j], w[j]) for j in range(J)])
    bound = tc._bound(mean, var, w, x)
    fin = np.isfinite(ref)
    ok = np.array_equal(np.isfinite(got), fin)
    used = 0.0
    if ok:
        allow = 5e-5 + 5e-6 * np.abs(ref[fin]) + bound[fin]
        used = float((np.abs(got[fin] - ref[fin]) / allow).max())
        ok = used <= 1.0
    split = int(((n_off > 0) & (n_off <= limit)).sum())
    pairs = int(n_off[n_off <= limit].sum()) * len(x)
    tot_exact += exact; tot_pairs += pairs
    worst = max(worst, used)
    print('seed %d: J=%d M=%d D=%d frames=%d split states %d (off-pipe %s of limit %d) exact pairs %d of %d: %s (%.2f of the allowance)'
          % (seed, J, M, D, len(x), split, n_off.tolist(), limit, exact, pairs, 'ok' if ok else 'FAILED', used), flush=True)
    bad += not ok
print('%d random cases, %d failures; worst share of the allowance used %.2f; %d of %d off-pipe pairs evaluated exactly (%.3f %%)'
      % (count, bad, worst, tot_exact, tot_pairs, 100.0 * tot_exact / max(tot_pairs, 1)))
sys.exit(1 if bad else 0)

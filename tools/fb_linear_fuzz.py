#!/usr/bin/env python3
"""Randomised soak of the scaled linear-domain forward-backward (csrc/hmm_fb_linear.inc) against the log-domain kernels it
replaced (PCL_FB_LINEAR=0) on the same batches: random numbers of states (2..256: one to four wavefronts per chain) and frames (1..400), self-loop probabilities
from 1e-6 to 1 - 1e-6, emissions from mild to thousands of nats apart with ln 0 sprinkled in, pi free / locked, thresholds
that end the pass loop anywhere.  usage: fb_linear_fuzz.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from poccala_amd import Engine
import test_gpu_fb_linear as tf

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eng = Engine(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    U = int(rng.integers(1, 24))
    scale = float(rng.choice([1.0, 8.0, 60.0, 700.0]))
    As, pis, Bs, sizes = [], [], [], []
    for _ in range(U):
        n, t = int(rng.integers(2, rng.choice([65, 65, 129, 257]))), int(rng.integers(1, 401))
        a = np.zeros((n, n))
        for i in range(n - 1):
            x = float(rng.choice([rng.uniform(0.05, 0.95), 1e-6, 1 - 1e-6]))
            a[i, i], a[i, i + 1] = x, 1.0 - x
        a[n - 1, n - 1] = float(rng.choice([0.0, 1.0]))
        p = rng.dirichlet(np.ones(n))
        b = -40.0 * rng.uniform(0.5, 3.0) + scale * rng.standard_normal((n, t))
        b[rng.random((n, t)) < rng.choice([0.0, 0.01, 0.2])] = -np.inf
        As.append(a); pis.append(p); Bs.append(b); sizes.append((n, t))
    fix = bool(rng.integers(0, 2))
    thr = float(rng.choice([0.64, 0.64, 1e9, -1.0, 1e-9]))
    res = {}
    try:
        for linear in (True, False):
            os.environ['PCL_FB_LINEAR'] = '1' if linear else '0'
            b = eng.batch([s[0] for s in sizes], [s[1] for s in sizes])
            with np.errstate(divide='ignore'):
                b.set_transitions([np.log(a) for a in As], [np.log(p) for p in pis])
            b.set_emissions(Bs)
            b.forward_backward(fix_pi=fix, threshold=thr)
            res[linear] = {k: b.get(k) for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi', 'logp', 'npass', 'qtrace')}
            b.close()
        # pass counts can differ only where Q - Q_prev sits within rounding of the threshold: none of these thresholds does
        # (pi after up to 16 re-estimations: exp() of differences of logarithms ~1e3 in size -- both kernels carry ~1e-12 of absolute
        #  rounding there, 1e-9 relative on a pi of 1e-100 after many passes)
        # (seed 121: 16 passes with a free pi, ln gamma ~ -24 apart by 1.27e-9 = 5e-11 relative -- the pi feedback carries the two
        #  chains' roundings from pass to pass; the absolute allowance is 4e-9 for that)
        tf.compare(res[True], res[False], rtol=1e-11, atol=4e-9, pi_rtol=1e-7)
    except Exception as e:                        # noqa
        bad += 1
        print('FAILED seed %d (U=%d scale=%g fix=%s thr=%g): %s' % (seed, U, scale, fix, thr, str(e)[:300]))
os.environ.pop('PCL_FB_LINEAR', None)
print('%d random batches, %d failures' % (count, bad))
sys.exit(1 if bad else 0)

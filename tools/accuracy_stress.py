#!/usr/bin/env python3
"""Accuracy of the f32 scoring kernels on harsher models than the bench's: mixture means spread over several
standard deviations inside a state, small variances, a large common offset (MFCC c0-like).  Reports max |d ln b|
against the float64 oracle for the MFMA kernel (per-state centred expansion) and the VALU kernel (direct form)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(variant):
    os.environ['PCL_SCORE_VARIANT'] = str(variant)
    from oracle import poccala_oracle as po
    from poccala_amd import Engine, PCL_F32
    rng = np.random.default_rng(0)
    M, D, T = 256, 39, 200
    eng = Engine(0)
    for name, spread, vlo, vhi, offset in (('bench-like', 1.0, 0.5, 2.0, 0.0), ('spread 3 sigma', 3.0, 0.5, 2.0, 0.0),
                                           ('spread 5, small var', 5.0, 0.05, 0.5, 0.0), ('offset 60', 1.0, 0.5, 2.0, 60.0),
                                           ('offset 60, spread 5, small var', 5.0, 0.05, 0.5, 60.0)):
        mean = rng.standard_normal((1, M, D)) * spread + offset
        var = rng.uniform(vlo, vhi, (1, M, D))
        w = rng.dirichlet(np.ones(M))[None]
        # frames drawn from the mixture itself (so that some component explains each frame)
        comp = rng.integers(0, M, T)
        x = (mean[0, comp] + np.sqrt(var[0, comp]) * rng.standard_normal((T, D))).astype(np.float32)
        eng.load_model(mean, var, w); eng.load_frames(x)
        b = eng.batch([3], [T], [0]); b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
        b.score(PCL_F32)
        got = b.get('B')[0][1]
        ref = po.gmm_point(x.astype(np.float64), mean[0], var[0], w[0])
        print('variant %d  %-32s max |d ln b| = %.3g   (|ln b| ~ %.0f)' % (variant, name, np.abs(got - ref).max(), np.abs(ref).mean()))
        b.close()

if len(sys.argv) > 1:
    run(int(sys.argv[1]))
else:
    for v in (1, 3, 7):
        subprocess.run([sys.executable, __file__, str(v)])

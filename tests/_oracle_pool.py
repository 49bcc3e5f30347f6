"""Run the float64 oracle for several utterances / state blocks on the host cores of the GPU box (spawned workers: the
pytest process has initialised HIP and must not fork).  Test infrastructure only."""
import multiprocessing as mp
import os

import numpy as np


def _label_job(args):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import poccala_oracle as po
    x, lab, model, fix_code, want_bw = args
    _, a, bref, pi = po.score_label(x, list(lab), model)
    if not want_bw:
        return bref, None, None, a, pi
    bw = po.baum_welch(a, pi, [bref], fix_code=fix_code)
    l = bw['alpha'][0] + bw['beta'][0]
    lg = l - po.lse(l, axis=0)[None, :]
    return bref, float(bw['logp'][0]), lg, a, pi


def _rows_job(args):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import poccala_oracle as po
    x, gmms = args
    out = np.empty((len(gmms), x.shape[0]))
    for k, (m, v, w) in enumerate(gmms):
        for s in range(0, x.shape[0], 20):                       # bounds the (T,M,D) temporary
            out[k, s:s + 20] = po.gmm_point(x[s:s + 20], m, v, w)
    return out


def workers():
    return max(1, min(64, (os.cpu_count() or 2) - 2))


def label_jobs(jobs):
    """jobs: [(x (T,D) f64, label, model dict restricted to the label's units, fix_code, want_bw)] -> list of
    (B_ref, logp, lgamma_ref, A, pi)."""
    with mp.get_context('spawn').Pool(min(workers(), len(jobs))) as pool:
        return pool.map(_label_job, jobs, chunksize=1)


def state_rows(x, gmms, block=8):
    """ln b_j(o_t) of many states for one utterance, the states split over the pool."""
    jobs = [(x, gmms[i:i + block]) for i in range(0, len(gmms), block)]
    with mp.get_context('spawn').Pool(min(workers(), len(jobs))) as pool:
        return np.concatenate(pool.map(_rows_job, jobs, chunksize=1), axis=0)

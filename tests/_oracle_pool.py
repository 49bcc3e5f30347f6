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
    if want_bw == 'xi':                            # + the un-normalised ln xi (N,N) and ln gamma (N,) of the final pass (quirk Q5)
        return bref, float(bw['logp'][0]), lg, a, pi, bw['ksai'], bw['gamma']
    if want_bw == 'xi+bound':                      # ... + the analytical f32 input-rounding bound of every emitting row (models EM has sharpened)
        rows = [model[u]['gmms'][k] for u in lab for k in range(len(model[u]['gmms']))]
        return bref, float(bw['logp'][0]), lg, a, pi, bw['ksai'], bw['gamma'], f32_evaluation_bound_rows(rows, x)
    return bref, float(bw['logp'][0]), lg, a, pi


def f32_evaluation_bound_rows(gmms, x, t_chunk=20):
    """tests/test_gpu_parity.py:f32_evaluation_bound for a list of (mean, var, w) states, in frame chunks (bounds the (T,M,D)
    temporaries at M = 2048): what an f32 evaluation of the Gaussian exponent may lose per (state, frame) -- parameters and frames
    are f32 roundings, every standardised residual carries up to 2^-23 (|x_d| + |mu_d|) / sigma_d, weighted with the mixture
    posteriors.  Returns (len(gmms), T)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros((len(gmms), x.shape[0]))
    for j, (mean, var, w) in enumerate(gmms):
        sig = np.sqrt(var)
        with np.errstate(divide='ignore'):
            lw = np.log(w)
        for s in range(0, x.shape[0], t_chunk):
            xs = x[s:s + t_chunk]
            z = (xs[:, None, :] - mean[None]) / sig[None]
            comp = lw[None] - 0.5 * (z ** 2).sum(-1) - 0.5 * var.sum(-1)[None]       # (the log-sum-exp weights: quirk Q1's constant, -1/2 sum var, per mixture)
            comp -= comp.max(1, keepdims=True)
            post = np.exp(comp)
            post /= post.sum(1, keepdims=True)
            per = 2.0 ** -23 * (((np.abs(xs)[:, None, :] + np.abs(mean)[None]) / sig[None]) * np.abs(z)).sum(-1)
            out[j, s:s + t_chunk] = (post * per).sum(1)
    return out


def _rows_job(args):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import poccala_oracle as po
    x, gmms = args
    out = np.empty((len(gmms), x.shape[0]))
    for k, (m, v, w) in enumerate(gmms):
        for s in range(0, x.shape[0], 20):                       # bounds the (T,M,D) temporary
            out[k, s:s + 20] = po.gmm_point(x[s:s + 20], m, v, w)
    return out


def _bw_units_job(args):
    """Baum-Welch on GIVEN emissions + LHMM.update_acc's slicing (LHMM.py:473-500) + add_acc (:149-161) merged per unit inside
    the utterance: {unit: (ksai_acc (S-2,S), gamma_acc (S-2,))}, log domain."""
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import poccala_oracle as po
    a, pi, b, label, s = args
    e = s - 2
    bw = po.baum_welch(a, pi, [b])
    kv, gv = bw['ksai'][1:-1, :], bw['gamma'][1:-1]
    out = {}
    for pos, unit in enumerate(label):
        k, g = kv[pos * e:pos * e + e, pos * e:pos * e + s], gv[pos * e:pos * e + e]
        if unit in out:
            out[unit] = (po.logaddexp_q4(out[unit][0], k), po.logaddexp_q4(out[unit][1], g))
        else:
            out[unit] = (k.copy(), g.copy())
    return out, float(bw['logp'][0]), int(bw['n_pass'])


def bw_unit_jobs(jobs):
    """jobs: [(A, pi, B (N,T), label, S)] -> [({unit: (ksai_acc, gamma_acc)}, logp, n_pass)]: the HMM half of the E-step on given emissions."""
    with mp.get_context('spawn').Pool(min(workers(), len(jobs))) as pool:
        return pool.map(_bw_units_job, jobs, chunksize=4)


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


def _acc_job(args):
    """A13 for ONE occurrence of a state (Clustering.GMM.update_acc, Clustering.py:653-680) through the oracle, the
    mixtures in slices (every mixture's accumulators are independent of the others; bounds the (M,D,T) temporaries).
    Returns the occurrence's contribution in the LINEAR domain: acc (M,), alpha_acc, mean_acc (M,D), cov_acc (M,D)."""
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import poccala_oracle as po
    x, l_value, b_value, mean, var, w, step = args
    m, d = mean.shape
    out = dict(acc=np.zeros(m), alpha_acc=0.0, mean_acc=np.zeros((m, d)), cov_acc=np.zeros((m, d)))
    for lo in range(0, m, step):
        hi = min(m, lo + step)
        a = dict(acc=np.full(hi - lo, -np.inf), alpha_acc=-np.inf, mean_acc=np.full((hi - lo, d), -np.inf),
                 cov_acc=np.full((hi - lo, d), -np.inf))
        po.gmm_update_acc(a, l_value, b_value, x, mean[lo:hi], var[lo:hi], w[lo:hi])
        with np.errstate(all='ignore'):
            out['acc'][lo:hi] = np.exp(a['acc'])
            out['mean_acc'][lo:hi] = np.exp(a['mean_acc'])
            out['cov_acc'][lo:hi] = np.exp(a['cov_acc'])
            out['alpha_acc'] = float(np.exp(a['alpha_acc']))
    return out


def state_statistics(jobs, step=128):
    """jobs: [(x (T,D) f64, ln gamma_t(j) (T,), ln b_j(o_t) (T,), mean, var, w)] -- the occurrences of one or more
    states; returns one linear-domain contribution per job (see _acc_job)."""
    jobs = [tuple(j) + (step,) for j in jobs]
    with mp.get_context('spawn').Pool(min(workers(), len(jobs))) as pool:
        return pool.map(_acc_job, jobs, chunksize=1)

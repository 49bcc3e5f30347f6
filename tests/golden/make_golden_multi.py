#!/usr/bin/env python3
"""Golden vectors for an LHMM that holds SEVERAL utterances (datasize > 1): the merge paths of
LHMM.__maximization (StatisticalModel/LHMM.py:454-466) and __expectation (:412-422).  Not on the per-utterance hot
path (AcousticModel always builds one embedded LHMM per utterance) but part of the class surface.  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import RecLog, import_reference  # noqa: E402


def main():
    _, util, LHMM, Clustering, AcousticModel = import_reference()
    rng = np.random.default_rng(1111)
    out = {}
    for tag, fix in (('free', 0), ('pifixed', 1)):
        n = 8
        a = np.zeros((n, n))
        a[0, 1] = 1.0
        for j in range(1, n - 1):
            a[j, j], a[j, j + 1] = 0.6, 0.4
        pi = np.ones(n) / n
        ts = [23, 31, 17]
        bs = []
        for t in ts:
            b = rng.standard_normal((n, t)) * 2 - 15
            b[0] = 0.0
            b[-1] = -np.inf
            bs.append(b)
        unit = LHMM({i: 'u' for i in range(n)}, n, RecLog(), transmat=a.copy(), probmat=[np.zeros((n, 1))], fix_code=6)
        log = RecLog()
        h = LHMM({i: 'u' for i in range(n)}, n, log, transmat=a.copy(), probmat=[b.copy() for b in bs], pi=pi.copy(),
                 hmm_list=[unit], fix_code=fix | 2)          # pdf locked: no GMMs behind this HMM
        h.add_data([np.zeros((t, 1)) for t in ts])
        h.add_T(list(ts))
        h.baulm_welch()
        qs = [float(m.split(':')[1]) for (c, m) in log.msgs if m.startswith('HMM 当前似然度')]
        out['A_' + tag] = a
        out['pi0_' + tag] = pi
        for k, b in enumerate(bs):
            out['B%d_%s' % (k, tag)] = b
        out['q_trace_' + tag] = np.array(qs)
        out['n_pass_' + tag] = np.int64(len(qs))
        out['pi_' + tag] = h.pi.copy()
        out['ksai_' + tag] = h._LHMM__ksai.copy()
        out['gamma_' + tag] = h._LHMM__gamma.copy()
        out['ksai_acc_' + tag] = unit.ksai_acc.copy()
        out['gamma_acc_' + tag] = unit.gamma_acc.copy()
        for k in range(len(ts)):
            out['alpha%d_%s' % (k, tag)] = h._LHMM__result_f[k].copy()
    np.savez_compressed(os.path.join(HERE, 'G11_multi_utterance.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', v))


if __name__ == '__main__':
    main()

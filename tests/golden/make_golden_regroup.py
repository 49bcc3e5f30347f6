#!/usr/bin/env python3
"""Golden vectors for the step after forced alignment, made by RUNNING the reference:
AcousticModel.discriminate (AcousticModel.py:937-955), the private __eq_segment in both modes (:587-627) and
__get_gmmdata (:629-644), called through their mangled names on a reference AcousticModel instance.
Build container only.  Writes tests/golden/G12_regroup.npz (inputs + the reference's outputs, no source text).

    python tests/golden/make_golden_regroup.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, RecLog   # noqa: E402


def main():
    scratch, util, LHMM, Clustering, AcousticModel = import_reference()
    am = AcousticModel(RecLog(), 'XIF_tone', processes=1, console=False, state_num=5)
    saved = []
    # __save_data pickles to disk; capture what it is given instead (the split itself is what is pinned)
    setattr(am, '_AcousticModel__save_data', lambda unit, unit_data: saved.append((unit, np.array(unit_data))))
    eq_segment = getattr(am, '_AcousticModel__eq_segment')
    get_gmmdata = getattr(am, '_AcousticModel__get_gmmdata')
    rng = np.random.default_rng(12)
    out = {}
    # mode 'g': one block -> S-2 slices (lengths incl. shorter than the number of states)
    lens = [1, 2, 3, 4, 5, 7, 10, 31]
    out['g_lens'] = np.array(lens)
    for n in lens:
        data = rng.standard_normal((n, 3))
        sl = eq_segment(data, 3, mode='g')
        out['g_data_%d' % n] = data
        out['g_sizes_%d' % n] = np.array([len(s) for s in sl])
        for k, s in enumerate(sl):
            out['g_slice_%d_%d' % (n, k)] = np.asarray(s).reshape(-1, 3)
    # mode 'e': equal split of an utterance over its label (frames beyond chunk * L are dropped)
    data = rng.standard_normal((23, 3))
    label = ['a', 'b', 'a', 'c']
    del saved[:]
    eq_segment(data, label, mode='e')
    out['e_data'] = data
    out['e_label'] = np.array(label)
    out['e_units'] = np.array([u for u, _ in saved])
    for i, (_, d) in enumerate(saved):
        out['e_block_%d' % i] = d
    # discriminate + __get_gmmdata on name sequences as AcousticModel.viterbi returns them
    cases = [['a'] * 4 + ['b'] * 5 + ['a'] * 3 + ['c'] * 2,          # a unit twice, apart
             ['a'] * 3 + ['a'] * 4 + ['b'] * 6,                        # the same unit twice in a row: one run
             ['b'] * 1 + ['c'] * 2 + ['d'] * 9]
    out['n_cases'] = np.array(len(cases))
    for ci, seq in enumerate(cases):
        data = rng.standard_normal((len(seq), 3))
        out['d_seq_%d' % ci] = np.array(seq)
        out['d_data_%d' % ci] = data
        for unit in sorted(set(seq)):
            runs = AcousticModel.discriminate(unit, np.array(seq))
            out['d_nruns_%d_%s' % (ci, unit)] = np.array(len(runs))
            blocks = []
            for ri, loc in enumerate(runs):
                out['d_loc_%d_%s_%d' % (ci, unit, ri)] = np.asarray(loc)
                blocks.append(data[loc])
            g = get_gmmdata(blocks)
            for k in range(3):
                out['d_gmm_%d_%s_%d' % (ci, unit, k)] = np.asarray(g[k]).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, 'G12_regroup.npz'), **out)
    print('G12_regroup.npz:', len(out), 'arrays')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Does the REFERENCE read what the build writes?  (f1's stated purpose: "so a reference
multi_embedded_training_2 could consume them", AcousticModel.py:918-935.)

Container-side, like make_golden.py (the reference cannot travel).  No GPU:

  1. The BUILD's host classes write a parameter tree for three units -- poccala_amd AcousticModel.init_unit / save_parameter
     (LHMM.save_parameter + GMM.save_parameter) -- and, twice, the accumulator files of a batch: AcousticModel.save_batch_acc fed the
     ORACLE's E-step statistics of two different utterance sets (linear GMM sums, log-domain per-unit xi / gamma sums: exactly what
     estep_batch hands it from the device).
  2. The imported REFERENCE reads that tree with its own code: per unit a reference LHMM with reference GMM states,
     LHMM.init_parameter / GMM.init_parameter (LHMM.py:243-254, Clustering.py:297-312), LHMM.init_acc / GMM.init_acc -- the merge of
     the two files (LHMM.py:256-290, Clustering.py:314-367) --, LHMM.update_param (LHMM.py:509-524 -> Clustering.py:682-693) at the
     driver's variance floor.
  3. What the reference ends up with -- A, w, mu, sigma^2 of every unit -- is fixture G17, with the inputs that made it.
     tests/test_oracle_golden.py holds the oracle's M-step on the summed statistics to it, and the build's own classes reading the
     same tree (tests/test_host_logic.py).

    python tests/golden/check_tree_with_reference.py       # rewrites tests/golden/G17_tree.npz"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

S, M, D = 5, 3, 5
UNITS = ['a1', 'b', 'zh']
C_COV = 1e-6                      # init.py:30 -> Controller.py:151


def oracle_batch(po, rng, model, n_utt):
    """the oracle's E-step over n_utt utterances: (stats of the (unit, k) states in sorted-unit order, linear; {unit: (ksai_acc, gamma_acc)} log)"""
    e = S - 2
    J = len(UNITS) * e
    stats = dict(acc=np.zeros((J, M)), alpha_acc=np.zeros(J), mean_acc=np.zeros((J, M, D)), cov_acc=np.zeros((J, M, D)))
    hk = {u: np.full((e, S), -np.inf) for u in UNITS}
    hg = {u: np.full((e,), -np.inf) for u in UNITS}
    for _ in range(n_utt):
        lab = list(rng.choice(UNITS, 3))
        x = rng.standard_normal((30, D))
        bw, accs, _ = po.estep_utterance(x, lab, model)
        for pos, unit in enumerate(lab):
            ui = UNITS.index(unit)
            hk[unit] = po.logaddexp_q4(hk[unit], accs[pos].ksai_acc)
            hg[unit] = po.logaddexp_q4(hg[unit], accs[pos].gamma_acc)
            for k in range(e):
                for key in stats:
                    stats[key][ui * e + k] += np.exp(accs[pos].gmm[k][key])
    return stats, {u: (hk[u], hg[u]) for u in UNITS}


def main():
    import make_golden
    scratch, util, RLHMM, RClustering, RAcousticModel = make_golden.import_reference()
    from oracle import poccala_oracle as po
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel as BuildAM
    rng = np.random.default_rng(1717)
    tree = os.environ['parameters_file_path']           # (where the imported reference looks: PARAMETERS_FILE_PATH is read at import)
    out = {'units': np.array(UNITS), 'c_covariance': np.float64(C_COV)}
    try:
        # ---- 1. the build writes
        am = BuildAM(None, 'XIF_tone', parameters_path=tree, state_num=S, mix_level=M, dct_num=D, delta_1=False, delta_2=False)
        unit_hmms, model = {}, {}
        for ui, u in enumerate(UNITS):
            h = am.init_unit(u)
            tr = np.zeros((S, S))
            tr[0][1] = 1.
            for j in range(1, S - 1):
                p = rng.uniform(0.3, 0.7)
                tr[j][j], tr[j][j + 1] = p, 1 - p
            h.transmat[:] = tr
            gm = []
            for k in range(S - 2):
                mean, var, w = make_golden.rand_gmm(rng, M, D)
                g = h.profunction[1 + k]
                g.alpha, g.mean, g.covariance = w.copy(), mean.copy(), make_golden.diag_cov(var)
                gm.append((mean, var, w))
                out['mean_%d_%d' % (ui, k)], out['var_%d_%d' % (ui, k)], out['w_%d_%d' % (ui, k)] = mean, var, w
            out['trans_%d' % ui] = tr
            am.save_parameter(u, h)
            unit_hmms[u] = h
            model[u] = dict(trans=tr, gmms=gm)
        for bi in range(2):
            stats, hacc = oracle_batch(po, rng, model, 4 + bi)
            for key, val in stats.items():
                out['batch%d_%s' % (bi, key)] = val
            for ui, u in enumerate(UNITS):
                out['batch%d_ksai_%d' % (bi, ui)], out['batch%d_gamma_%d' % (bi, ui)] = hacc[u]
            am.save_batch_acc(stats, hacc, unit_hmms)
        # ---- 2. the reference reads: its own multi_embedded_training_2 (AcousticModel.py:918-935) = init_unit -> init_parameter (:227-241)
        #         -> __cal_hmm (:519-530: __init_acc merges every accumulator file of the unit, LHMM.py:256-290 + Clustering.py:314-367;
        #         LHMM.update_param :509-524 -> GMM.update_param Clustering.py:682-693) -> __save_parameter.  Beside it, the same steps by
        #         hand on a second set of reference objects, to record what it read and merged on the way.
        ram = RAcousticModel(make_golden.RecLog(), 'XIF_tone', processes=1, console=False, state_num=S, mix_level=M, dct_num=D,
                             delta_1=False, delta_2=False)
        for ui, u in enumerate(UNITS):
            path = am.unit_path(u)
            h = ram.init_unit(u, new_log=True, fix_code=0)
            ram.init_parameter(u, h)
            gmms = [h.profunction[1 + k] for k in range(S - 2)]
            # what it read back IS what the build wrote
            assert np.array_equal(h.transmat, out['trans_%d' % ui])
            for k, g in enumerate(gmms):
                assert np.array_equal(np.array(g.mean), out['mean_%d_%d' % (ui, k)]) and np.array_equal(np.array(g.alpha), out['w_%d_%d' % (ui, k)])
                assert np.array_equal(np.array([np.diagonal(c) for c in g.covariance]), out['var_%d_%d' % (ui, k)])
            ram._AcousticModel__init_acc(u, h)                      # the merge of the two batches' files
            out['ref_ksai_acc_%d' % ui], out['ref_gamma_acc_%d' % ui] = h.ksai_acc.copy(), h.gamma_acc.copy()
            for k, g in enumerate(gmms):
                out['ref_acc_%d_%d' % (ui, k)] = np.array(g.acc)
                out['ref_alpha_acc_%d_%d' % (ui, k)] = np.float64(g.alpha_acc)
                out['ref_mean_acc_%d_%d' % (ui, k)] = np.array(g.mean_acc)
        for ui, u in enumerate(UNITS):
            ram.log = make_golden.RecLog()
            ram.multi_embedded_training_2(u, True, False, False, C_COV, ui, 0, len(UNITS), 0)
            # ... and what it SAVED, read straight from the files
            path = am.unit_path(u)
            out['new_trans_%d' % ui] = np.load(path + '/HMM/transmat.npy')
            for k in range(S - 2):
                gp = path + '/GMM_%d' % k
                out['new_w_%d_%d' % (ui, k)] = np.load(gp + '/GMM_weight.npy')
                out['new_mean_%d_%d' % (ui, k)] = np.load(gp + '/GMM_means.npy')
                cov = np.load(gp + '/GMM_covariance.npy')
                out['new_var_%d_%d' % (ui, k)] = np.array([np.diagonal(c) for c in cov])
        np.savez_compressed(os.path.join(HERE, 'G17_tree.npz'), **out)
        print('the reference read the build\'s tree: %d units, 2 accumulator files each; G17_tree.npz written' % len(UNITS))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()

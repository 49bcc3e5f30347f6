#!/usr/bin/env python3
"""Golden G16: the reference's own E-step (cal_observation_pro -> embedded -> LHMM(probmat).baulm_welch -> update_acc -> update_param) on
the ill-conditioned models the randomised GPU tests draw (tests/test_gpu_fuzz_estep.py): mixtures at the 1e-6 variance floor ('tight'),
variances over four decades inside a state ('wide'), weights down to 1e-12 with means far from the state's centre ('skewed'); frames sampled
from the model along the label.  So that "held to the oracle" on such draws means "held to the reference".  Runs in the build container
only (imports /root/reference through make_golden.import_reference); data only.

    python tests/golden/make_golden_kinds.py      # rewrites tests/golden/G16_kinds.npz"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import RecLog, diag_cov, import_reference  # noqa: E402

S = 5
E = S - 2


def main():
    warnings.simplefilter('ignore')
    scratch, util, LHMM, Clustering, AcousticModel = import_reference()
    GMM = Clustering.GMM
    am = AcousticModel(RecLog(), 'XIF_tone', processes=1, console=False, state_num=5)
    out = {}
    for ci, kind in enumerate(['tight', 'wide', 'skewed']):
        rng = np.random.default_rng(1700 + ci)
        label = ['b', 'a1']
        m, d, t = 9, 13, 30
        unit_params = {}
        for u in label:
            params = []
            for k in range(E):
                mean = rng.standard_normal((m, d))
                var = rng.uniform(0.5, 2.0, (m, d))
                w = rng.dirichlet(np.ones(m))
                if kind == 'tight':
                    hit = rng.random(m) < 0.35
                    var[hit] = 1e-6 * rng.uniform(1.0, 3.0, size=(int(hit.sum()), d))
                elif kind == 'wide':
                    var *= 10.0 ** rng.uniform(-2, 2, size=(m, 1))
                else:
                    w = w * 10.0 ** rng.uniform(-12, 0, size=m)
                    w /= w.sum()
                    mean += 6.0 * rng.standard_normal((1, d))
                params.append((mean, var, w))
            unit_params[u] = params
        # frames along the label, sampled from the model
        st = np.repeat([(u, k) for u in label for k in range(E)], t // (E * len(label)), axis=0)
        x = np.empty((t, d))
        for i in range(t):
            u, k = st[min(i, len(st) - 1)]
            mean, var, w = unit_params[u][int(k)]
            mix = int(rng.integers(0, m))
            x[i] = mean[mix] + np.sqrt(var[mix]) * rng.standard_normal(d)
        tag = kind
        out[tag + '_x'] = x
        out[tag + '_label'] = np.array(label)
        names = sorted(unit_params)
        out[tag + '_unit_names'] = np.array(names)
        hmm_list = []
        for u in label:
            trans = np.zeros((S, S))
            trans[0][1] = 1.
            for j in range(1, S - 1):
                trans[j][j] = 0.5
                trans[j][j + 1] = 0.5
            gmms = [GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(), covariance=diag_cov(var), gmm_id=k)
                    for k, (mean, var, w) in enumerate(unit_params[u])]
            prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
            h = LHMM({i: u for i in range(S)}, S, RecLog(), transmat=trans.copy(), profunc=prof, fix_code=0)
            with np.errstate(all='ignore'):
                h.cal_observation_pro([x], [t])
            h.clear_data()
            hmm_list.append(h)
        for ui, u in enumerate(names):
            for k, (mean, var, w) in enumerate(unit_params[u]):
                out['%s_mean_%d_%d' % (tag, ui, k)] = mean
                out['%s_var_%d_%d' % (tag, ui, k)] = var
                out['%s_w_%d_%d' % (tag, ui, k)] = w
        states, A, B, pi = am.embedded(list(label), hmm_list, 0, 15)
        out[tag + '_emb_B'] = B.copy()
        elog = RecLog()
        with np.errstate(all='ignore'):
            embed = LHMM(states, S, elog, transmat=A, probmat=[B], pi=pi, hmm_list=hmm_list, fix_code=0)
            embed.add_data([x])
            embed.add_T([t])
            embed.baulm_welch(show_q=False)
        qs = [float(msg.split(':')[1]) for (c, msg) in elog.msgs if msg.startswith('HMM 当前似然度')]
        out[tag + '_q_trace'] = np.array(qs)
        out[tag + '_logp'] = np.float64(util.log_sum_exp(embed._LHMM__result_f[0][:, -1]))
        for pos, h in enumerate(hmm_list):
            out['%s_ksai_acc_%d' % (tag, pos)] = np.array(h.ksai_acc, dtype=np.float64)
            out['%s_gamma_acc_%d' % (tag, pos)] = np.array(h.gamma_acc, dtype=np.float64)
            for k in range(E):
                g = h.profunction[1 + k]
                out['%s_acc_%d_%d' % (tag, pos, k)] = np.array(g.acc, dtype=np.float64)
                out['%s_alpha_acc_%d_%d' % (tag, pos, k)] = np.float64(g.alpha_acc)
                out['%s_mean_acc_%d_%d' % (tag, pos, k)] = np.array(g.mean_acc, dtype=np.float64)
                out['%s_cov_acc_%d_%d' % (tag, pos, k)] = np.array(g._GMM__covariance_acc, dtype=np.float64)
        with np.errstate(all='ignore'):
            for pos, h in enumerate(hmm_list):
                h.fix_code = 0
                h.update_param(c_covariance=1e-6)
                for k in range(E):
                    g = h.profunction[1 + k]
                    out['%s_new_w_%d_%d' % (tag, pos, k)] = np.array(g.alpha)
                    out['%s_new_mean_%d_%d' % (tag, pos, k)] = np.array(g.mean)
                    out['%s_new_var_%d_%d' % (tag, pos, k)] = np.array([np.diagonal(c) for c in g.covariance])
        print(kind, 'q', qs, 'logp', float(out[tag + '_logp']), 'B range', np.nanmin(B[np.isfinite(B)]), np.nanmax(B[np.isfinite(B)]))
    np.savez_compressed(os.path.join(HERE, 'G16_kinds.npz'), **out)


if __name__ == '__main__':
    main()

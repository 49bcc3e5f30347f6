#!/usr/bin/env python3
"""Golden G15: what the REFERENCE does on the inputs where the library documents a deviation -- a one-frame
utterance, a two-frame utterance, an utterance too short for its label -- by running its own worker sequence
(cal_observation_pro -> embedded -> LHMM(probmat).baulm_welch -> update_acc), recorded as data: arrays with their NaNs, or the
exception's type name when the reference raises.  Runs in the build container only (imports /root/reference through
make_golden.import_reference); nothing of the reference's source text is stored.

    python tests/golden/make_golden_edges.py      # rewrites tests/golden/G15_edges.npz"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import RecLog, diag_cov, import_reference, rand_gmm  # noqa: E402

S = 5


def main():
    warnings.simplefilter('ignore')
    scratch, util, LHMM, Clustering, AcousticModel = import_reference()
    GMM = Clustering.GMM
    am = AcousticModel(RecLog(), 'XIF_tone', processes=1, console=False, state_num=5)
    out = {}
    cases = [('t1_l1', ['b'], 1), ('t2_l1', ['b'], 2), ('t3_l1', ['b'], 3), ('t1_l2', ['b', 'a1'], 1), ('t2_l4', ['b', 'a1', 'b', 'ing2'], 2),
             ('p0', ['b', 'a1'], 12)]        # p0: a frame no state can emit (its column of B set to ln 0): P(O) = 0
    for ci, (tag, label, t) in enumerate(cases):
        rng = np.random.default_rng(1500 + ci)
        m, d = 3, 5
        x = rng.standard_normal((t, d))
        out[tag + '_x'] = x
        out[tag + '_label'] = np.array(label)
        unit_params, hmm_list = {}, []
        for u in label:
            if u not in unit_params:
                unit_params[u] = [rand_gmm(np.random.default_rng(7000 + 10 * ci + len(unit_params) * 3 + k), m, d) for k in range(S - 2)]
            params = unit_params[u]
            trans = np.zeros((S, S))
            trans[0][1] = 1.
            for j in range(1, S - 1):
                trans[j][j] = 0.5
                trans[j][j + 1] = 0.5
            gmms = [GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(), covariance=diag_cov(var), gmm_id=k)
                    for k, (mean, var, w) in enumerate(params)]
            prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
            h = LHMM({i: u for i in range(S)}, S, RecLog(), transmat=trans.copy(), profunc=prof, fix_code=0)
            h.cal_observation_pro([x], [t])
            h.clear_data()
            hmm_list.append(h)
        names = sorted(unit_params)
        out[tag + '_unit_names'] = np.array(names)
        for ui, u in enumerate(names):
            for k, (mean, var, w) in enumerate(unit_params[u]):
                out['%s_mean_%d_%d' % (tag, ui, k)] = mean
                out['%s_var_%d_%d' % (tag, ui, k)] = var
                out['%s_w_%d_%d' % (tag, ui, k)] = w
        states, A, B, pi = am.embedded(list(label), hmm_list, 0, 15)
        if tag == 'p0':
            B[1:-1, 5] = -np.inf
        out[tag + '_emb_A'], out[tag + '_emb_B'], out[tag + '_emb_pi'] = A.copy(), B.copy(), pi.copy()
        elog = RecLog()
        raised = ''
        try:
            with np.errstate(all='ignore'):
                embed = LHMM(states, S, elog, transmat=A, probmat=[B], pi=pi, hmm_list=hmm_list, fix_code=0)
                embed.add_data([x])
                embed.add_T([t])
                embed.baulm_welch(show_q=False)
        except Exception as ex:             # noqa: BLE001 -- the fixture records WHAT the reference does, whatever it is
            raised = type(ex).__name__
        out[tag + '_raised'] = np.array(raised)
        qs = [float(msg.split(':')[1]) for (c, msg) in elog.msgs if msg.startswith('HMM 当前似然度')]
        out[tag + '_q_trace'] = np.array(qs)
        if not raised:
            out[tag + '_pi'] = embed.pi.copy()
            out[tag + '_ksai'] = embed._LHMM__ksai.copy()
            out[tag + '_gamma'] = embed._LHMM__gamma.copy()
            out[tag + '_logp'] = np.float64(util.log_sum_exp(embed._LHMM__result_f[0][:, -1]))
        for pos, h in enumerate(hmm_list):              # whatever update_acc left (the initial ln 0 when it never ran)
            out['%s_ksai_acc_%d' % (tag, pos)] = np.array(h.ksai_acc, dtype=np.float64)
            out['%s_gamma_acc_%d' % (tag, pos)] = np.array(h.gamma_acc, dtype=np.float64)
            for k in range(S - 2):
                g = h.profunction[1 + k]
                out['%s_acc_%d_%d' % (tag, pos, k)] = np.array(g.acc, dtype=np.float64)
                out['%s_alpha_acc_%d_%d' % (tag, pos, k)] = np.float64(g.alpha_acc)
                out['%s_mean_acc_%d_%d' % (tag, pos, k)] = np.array(g.mean_acc, dtype=np.float64)
                out['%s_cov_acc_%d_%d' % (tag, pos, k)] = np.array(g._GMM__covariance_acc, dtype=np.float64)
        print(tag, 'T', t, 'label', label, 'raised', repr(raised), 'q', qs, 'ksai_acc[0] nan?', bool(np.isnan(out[tag + '_ksai_acc_0']).any()),
              'finite?', bool(np.isfinite(out[tag + '_ksai_acc_0']).any()), 'acc[0][0]', out[tag + '_acc_0_0'])
    # ---- a frame far from the state's centre whose best mixture sits in the third 32-mixture tile, 89.2 / 90 / 91 nats above the first tile, a
    #      runner-up 1 / 2 / 3 nats below it in the second: the case on which the matrix-pipe log-sum-exp of rounds 1-5 lost the runner-up
    #      (DESIGN 4.10).  GMM.point of the reference itself on it (tests/test_gpu_parity.py:far_frame_problem is the same construction).
    for gi, (gap, runner_up) in enumerate([(89.2, 1.0), (90.0, 2.0), (91.0, 3.0), (130.0, 1.0)]):
        rng = np.random.default_rng(1600 + gi)
        M, D, T, R = 96, 39, 6, 30.0
        e = np.zeros(D)
        e[0] = 1.0
        C, p_best = 3.2, 3.0
        p = np.full(M, p_best - (gap + 40.0) / R)
        p[:32] = p_best - gap / R - rng.uniform(0, 0.3, 32) / R
        p[40] = p_best - runner_up / R
        p[77] = p_best
        u = rng.standard_normal((M, D))
        u[:, 0] = 0.0
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        mean = p[:, None] * e[None, :] + np.sqrt(C * C - p * p)[:, None] * u
        var = np.ones((M, D))
        w = np.full(M, 1.0 / M)
        x = (R * e[None, :] + 0.003 * rng.standard_normal((T, D))).astype(np.float32).astype(np.float64)
        gmm = GMM(RecLog(), dimension=D, mix_level=M, alpha=w.copy(), mean=mean.copy(), covariance=diag_cov(var))
        val = np.array([gmm.point(x[t].copy(), log=True) for t in range(T)])
        out['far%d_mean' % gi], out['far%d_var' % gi], out['far%d_w' % gi], out['far%d_x' % gi], out['far%d_point' % gi] = mean, var, w, x, val
        out['far%d_gap' % gi] = np.array([gap, runner_up])
        print('far frame', gap, runner_up, 'ln b', val[:3])
    # ---- LHMM.viterbi with end_state_back on HMMs of fewer than four states: len(p_list) - 4 + argmax(p_list[-4:]) is a negative index there,
    #      which NumPy wraps (LHMM.py:587-588); N = 2, 3 and, for comparison, 4 and 6
    for n in (2, 3, 4, 6):
        rng = np.random.default_rng(1800 + n)
        t = 9
        A = rng.dirichlet(np.ones(n), size=n)
        pi = rng.dirichlet(np.ones(n))
        prob = rng.standard_normal((n, t)) * 3 - 5
        states = {i: 's%d' % i for i in range(n)}
        point, path = LHMM.viterbi(RecLog(), states, A, prob, pi, end_state_back=True)
        out['esb%d_A' % n], out['esb%d_pi' % n], out['esb%d_prob' % n] = A, pi, prob
        out['esb%d_point' % n], out['esb%d_path' % n] = np.float64(point), np.array(path, dtype=np.float64)
        print('end_state_back N', n, 'point', point, 'path', path)
    # ---- GMM.update_param after an E-step in which one mixture has weight 0 (ln 0 = -inf everywhere: its accumulators stay ln 0): what the
    #      reference's M-step makes of 0 / 0 (Clustering.py:682-693)
    rng = np.random.default_rng(1900)
    m, d, t = 4, 5, 12
    mean = rng.standard_normal((m, d))
    var = rng.uniform(0.5, 2.0, (m, d))
    w = rng.dirichlet(np.ones(m))
    w[2] = 0.0
    w /= w.sum()
    x = rng.standard_normal((t, d))
    lval = np.log(rng.dirichlet(np.ones(t)) * 3.0)                # ln gamma_t(j) of some state
    gmm = GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(), covariance=diag_cov(var))
    with np.errstate(all='ignore'):
        bval = np.array([gmm.point(x[i].copy(), log=True, record=True) for i in range(t)])
        gmm.add_data(x) if hasattr(gmm, 'add_data') else None
        try:
            gmm.update_acc(lval, bval, x)
            raised = ''
        except Exception as ex:             # noqa: BLE001
            raised = type(ex).__name__
        out['zero_raised_acc'] = np.array(raised)
        out['zero_acc'] = np.array(gmm.acc, dtype=np.float64)
        out['zero_alpha_acc'] = np.float64(gmm.alpha_acc)
        out['zero_mean_acc'] = np.array(gmm.mean_acc, dtype=np.float64)
        out['zero_cov_acc'] = np.array(gmm._GMM__covariance_acc, dtype=np.float64)
        try:
            gmm.update_param(c_covariance=1e-3)
            raised = ''
        except Exception as ex:             # noqa: BLE001
            raised = type(ex).__name__
    out['zero_raised_mstep'] = np.array(raised)
    out['zero_mean'], out['zero_var'], out['zero_w'], out['zero_x'], out['zero_lval'], out['zero_bval'] = mean, var, w, x, lval, bval
    out['zero_new_w'] = np.array(gmm.alpha, dtype=np.float64)
    out['zero_new_mean'] = np.array(gmm.mean, dtype=np.float64)
    out['zero_new_var'] = np.array([np.diagonal(c) for c in gmm.covariance], dtype=np.float64)
    print('zero-weight mixture: update_acc raised', repr(str(out['zero_raised_acc'])), 'update_param raised', repr(raised), 'acc', out['zero_acc'], 'new w', out['zero_new_w'], 'new mean[2]', out['zero_new_mean'][2], 'new var[2]', out['zero_new_var'][2])
    np.savez_compressed(os.path.join(HERE, 'G15_edges.npz'), **out)


if __name__ == '__main__':
    main()

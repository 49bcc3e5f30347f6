#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script is the only place that imports /root/reference.  It runs in the
build container only (the reference does not exist on the GPU box) and writes
small .npz fixtures holding inputs + the reference's outputs.  Nothing from
the reference's source text is stored, only data.

    python tests/golden/make_golden.py        # rewrites tests/golden/G*.npz

Groups:
  G1 util.gaussian_function / log_sum_exp / matrix_log_sum_exp
  G2 Clustering.GMM.point(log=True, record=True)
  G3 LHMM.cal_observation_pro on a unit HMM (config C1 shape)
  G4 AcousticModel.embedded (states, A, B, pi)
  G5 LHMM.viterbi (+ AcousticModel.viterbi convert=True, discriminate)
  G6 LHMM.baulm_welch for fix_code 0..3 (Q trace, alpha/beta, pi, ksai/gamma acc)
  G7 Clustering.GMM.update_acc accumulators after G6
  G8 LHMM.update_param / GMM.update_param (incl. variance floor hit)
  G9 on-disk layout written by save_parameter / save_acc
"""
import os
import sys
import tempfile
import types
import warnings

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    """Make the reference importable: pyaudio stub, env vars, no bytecode."""
    sys.dont_write_bytecode = True
    warnings.simplefilter('ignore')
    scratch = tempfile.mkdtemp(prefix='pcl_golden_')
    os.makedirs(os.path.join(scratch, 'PARAMS'), exist_ok=True)
    os.makedirs(os.path.join(scratch, 'LOG'), exist_ok=True)
    os.environ['unit_file_path'] = os.path.join(REF, 'AcousticModel', 'Unit')
    os.environ['parameters_file_path'] = os.path.join(scratch, 'PARAMS')
    os.environ['log_file_path'] = os.path.join(scratch, 'LOG')
    stub = types.ModuleType('pyaudio')
    stub.PyAudio = type('PyAudio', (), {})
    stub.paInt16 = 8
    sys.modules['pyaudio'] = stub
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use('Agg')
    from StatisticalModel import util
    from StatisticalModel.LHMM import LHMM
    from StatisticalModel.Clustering import Clustering
    from AcousticModel.AcousticModel import AcousticModel
    return scratch, util, LHMM, Clustering, AcousticModel


class RecLog(object):
    """Duck-typed logger (reference: LogPrint.Log.note) that records messages."""
    unit_type = 'XIF_tone'
    console = False

    def __init__(self):
        self.msgs = []

    def note(self, content, cls='i', show_console=True):
        self.msgs.append((cls, content))

    def close(self):
        pass


def diag_cov(var):
    """(M,D) variances -> (M,D,D) full matrices (reference layout, quirk Q2)."""
    m, d = var.shape
    out = np.zeros((m, d, d))
    for i in range(m):
        out[i] = np.diag(var[i])
    return out


def rand_gmm(rng, m, d, dirichlet=True):
    mean = rng.standard_normal((m, d))
    var = rng.uniform(0.5, 2.0, (m, d))
    if dirichlet:
        w = rng.dirichlet(np.ones(m))
    else:
        w = np.ones(m) / m
    return mean, var, w


def main():
    scratch, util, LHMM, Clustering, AcousticModel = import_reference()
    GMM = Clustering.GMM
    log = RecLog()
    am = AcousticModel(log, 'XIF_tone', processes=1, console=False, state_num=5)
    S = 5

    def unit_hmm(rng, name, m, d, fix_code=0, trans=None):
        """What AcousticModel.init_unit + init_parameter build, without files."""
        states = {i: name for i in range(S)}
        if trans is None:
            trans = np.zeros((S, S))
            trans[0][1] = 1.
            for j in range(1, S - 1):
                trans[j][j] = 0.5
                trans[j][j + 1] = 0.5
        gmms = []
        params = []
        for k in range(S - 2):
            mean, var, w = rand_gmm(rng, m, d)
            params.append((mean, var, w))
            gmms.append(GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(),
                            covariance=diag_cov(var), gmm_id=k))
        prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
        hmm = LHMM(states, S, RecLog(), transmat=trans.copy(), profunc=prof, fix_code=fix_code)
        return hmm, params

    # ------------------------------------------------------------------ G1
    rng = np.random.default_rng(101)
    g1 = {}
    for d in (13, 39):
        y = rng.standard_normal((6, d))
        mean = rng.standard_normal((6, d))
        var = rng.uniform(0.5, 2.0, (6, d))
        out = np.array([util.gaussian_function(y[i].copy(), mean[i], np.diag(var[i]), d, log=True)
                        for i in range(6)])
        g1['gauss_y_%d' % d] = y
        g1['gauss_mean_%d' % d] = mean
        g1['gauss_var_%d' % d] = var
        g1['gauss_out_%d' % d] = out
    lse_in = [rng.standard_normal(7) * 30,
              np.array([-np.inf, -3.0, 2.5, -700.0]),
              np.array([-np.inf, -np.inf, -np.inf]),
              np.array([1.0, np.inf, -2.0]),
              np.array([-1e4, -1e4 + 1e-3, -1e4 - 5])]
    for i, v in enumerate(lse_in):
        g1['lse_in_%d' % i] = v
        g1['lse_out_%d' % i] = np.float64(util.log_sum_exp(v))
    m2 = rng.standard_normal((5, 9)) * 10
    m2[1, :] = -np.inf
    m2[3, 2] = -np.inf
    g1['lse_vec_in'] = m2
    g1['lse_vec_out'] = util.log_sum_exp(m2, vector=True)
    mats = [rng.standard_normal((4, 6)) * 20 for _ in range(3)]
    mats[0][1, 2] = -np.inf
    mats[1][1, 2] = -np.inf
    mats[2][1, 2] = -np.inf
    mats[1][0, 0] = -np.inf
    g1['mlse_in'] = np.array(mats)
    g1['mlse_out_full'] = util.matrix_log_sum_exp(mats, axis_x=4)
    g1['mlse_out_3'] = util.matrix_log_sum_exp(mats, axis_x=3)
    np.savez_compressed(os.path.join(HERE, 'G1_util.npz'), **g1)

    # ------------------------------------------------------------------ G2
    rng = np.random.default_rng(102)
    g2 = {}
    for (m, d) in ((4, 13), (8, 39), (256, 39)):
        mean, var, w = rand_gmm(rng, m, d, dirichlet=(m != 4))
        x = rng.standard_normal((32, d))
        gmm = GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(), covariance=diag_cov(var))
        out = np.array([gmm.point(x[t].copy(), log=True, record=True) for t in range(32)])
        rec = np.array(gmm._GMM__record)
        key = '%d_%d' % (m, d)
        g2['mean_' + key] = mean
        g2['var_' + key] = var
        g2['w_' + key] = w
        g2['x_' + key] = x
        g2['out_' + key] = out
        g2['record_' + key] = rec
    np.savez_compressed(os.path.join(HERE, 'G2_gmm_point.npz'), **g2)

    # ------------------------------------------------------------------ G3
    rng = np.random.default_rng(103)
    hmm, params = unit_hmm(rng, 'a1', 4, 13)
    x = rng.standard_normal((300, 13))
    hmm.cal_observation_pro([x], [300])
    g3 = {'x': x, 'B': hmm.B_p[0]}
    for k, (mean, var, w) in enumerate(params):
        g3['mean_%d' % k] = mean
        g3['var_%d' % k] = var
        g3['w_%d' % k] = w
    np.savez_compressed(os.path.join(HERE, 'G3_unit_B.npz'), **g3)

    # ------------------------------------------------------------ G4 .. G8
    def pipeline(seed, label, m, d, t, fix_code, c_cov, tag, store_ab, trained_trans=False):
        """The reference's multi_embedded_training_1 + _2 for one utterance."""
        rng = np.random.default_rng(seed)
        out = {}
        x = rng.standard_normal((t, d))
        out['x'] = x
        out['label'] = np.array(label)
        hmm_list = []
        unit_params = {}
        for pos, u in enumerate(label):
            if u not in unit_params:
                tr = None
                if trained_trans:
                    tr = np.zeros((S, S))
                    tr[0][1] = 1.
                    for j in range(1, S - 1):
                        p = rng.uniform(0.2, 0.8)
                        tr[j][j] = p
                        tr[j][j + 1] = 1 - p
                r2 = np.random.default_rng(seed * 1000 + len(unit_params))
                _, params = unit_hmm(r2, u, m, d)
                unit_params[u] = (params, tr)
            params, tr = unit_params[u]
            # a fresh instance per label position, same parameters for a repeated unit
            states = {i: u for i in range(S)}
            trans = tr
            if trans is None:
                trans = np.zeros((S, S))
                trans[0][1] = 1.
                for j in range(1, S - 1):
                    trans[j][j] = 0.5
                    trans[j][j + 1] = 0.5
            gmms = [GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(),
                        covariance=diag_cov(var), gmm_id=k) for k, (mean, var, w) in enumerate(params)]
            prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
            h = LHMM(states, S, RecLog(), transmat=trans.copy(), profunc=prof, fix_code=0)
            h.cal_observation_pro([x], [t])
            h.clear_data()
            hmm_list.append(h)
        names = sorted(unit_params)
        out['unit_names'] = np.array(names)
        for ui, u in enumerate(names):
            params, tr = unit_params[u]
            for k, (mean, var, w) in enumerate(params):
                out['mean_%d_%d' % (ui, k)] = mean
                out['var_%d_%d' % (ui, k)] = var
                out['w_%d_%d' % (ui, k)] = w
            out['trans_%d' % ui] = hmm_list[list(label).index(u)].transmat.copy()
        states, A, B, pi = am.embedded(list(label), hmm_list, 0, 15)
        out['emb_states'] = np.array([states[i] for i in range(len(states))])
        out['emb_A'] = A.copy()
        out['emb_B'] = B.copy()
        out['emb_pi'] = pi.copy()
        # Viterbi on the same sentence HMM (AcousticModel.viterbi -> convert=True)
        vlog = RecLog()
        point, seq = LHMM.viterbi(vlog, states, A, B, pi, convert=False)
        out['vit_point'] = np.float64(point)
        out['vit_path'] = seq.copy()
        point2, seq2 = am.viterbi(states, A, B, pi)
        out['vit_point_conv'] = np.float64(point2)
        out['vit_path_conv'] = np.array(seq2)
        for u in names:
            runs = AcousticModel.discriminate(u, seq2)
            out['disc_%s_n' % u] = np.int64(len(runs))
            order = np.argsort([r[0] for r in runs]) if len(runs) else []
            for ri, oi in enumerate(order):
                out['disc_%s_%d' % (u, ri)] = runs[oi]
        # Baum-Welch
        elog = RecLog()
        embed = LHMM(states, S, elog, transmat=A, probmat=[B], pi=pi, hmm_list=hmm_list, fix_code=fix_code)
        embed.add_data([x])
        embed.add_T([t])
        embed.baulm_welch(show_q=False)
        qs = [float(msg.split(':')[1]) for (c, msg) in elog.msgs if msg.startswith('HMM 当前似然度')]
        out['bw_q_trace'] = np.array(qs)
        out['bw_n_pass'] = np.int64(len(qs))
        out['bw_pi'] = embed.pi.copy()
        out['bw_ksai'] = embed._LHMM__ksai.copy()
        out['bw_gamma'] = embed._LHMM__gamma.copy()
        if store_ab:
            out['bw_alpha'] = embed._LHMM__result_f[0].copy()
            out['bw_beta'] = embed._LHMM__result_b[0].copy()
        out['bw_logp'] = np.float64(util.log_sum_exp(embed._LHMM__result_f[0][:, -1]))
        for pos, h in enumerate(hmm_list):
            out['ksai_acc_%d' % pos] = h.ksai_acc.copy()
            out['gamma_acc_%d' % pos] = h.gamma_acc.copy()
            for k in range(S - 2):
                g = h.profunction[1 + k]
                out['acc_%d_%d' % (pos, k)] = np.array(g.acc)
                out['alpha_acc_%d_%d' % (pos, k)] = np.float64(g.alpha_acc)
                out['mean_acc_%d_%d' % (pos, k)] = np.array(g.mean_acc)
                out['cov_acc_%d_%d' % (pos, k)] = np.array(g._GMM__covariance_acc)
        # M-step per label position (what multi_embedded_training_2 would do with a
        # single accumulator file): LHMM.update_param -> GMM.update_param
        for pos, h in enumerate(hmm_list):
            h.fix_code = fix_code
            h.update_param(c_covariance=c_cov)
            out['new_trans_%d' % pos] = h.transmat.copy()
            for k in range(S - 2):
                g = h.profunction[1 + k]
                out['new_w_%d_%d' % (pos, k)] = np.array(g.alpha)
                out['new_mean_%d_%d' % (pos, k)] = np.array(g.mean)
                out['new_var_%d_%d' % (pos, k)] = np.array([np.diagonal(c) for c in g.covariance])
        out['fix_code'] = np.int64(fix_code)
        out['c_covariance'] = np.float64(c_cov)
        np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)
        return hmm_list

    # G4/G5/G6/G7/G8 small: 4-unit label with a repeated unit, M=4, D=13
    hl = None
    for fc in (0, 1, 2, 3, 4, 6):
        hl = pipeline(seed=200 + fc, label=['b', 'a1', 'b', 'ing2'], m=4, d=13, t=60, fix_code=fc,
                      c_cov=1e-3, tag='G6_small_fix%d' % fc, store_ab=(fc in (0, 3)),
                      trained_trans=(fc in (1, 6)))
    # variance-floor hit: huge floor
    pipeline(seed=231, label=['zh', 'ong1'], m=3, d=5, t=40, fix_code=0, c_cov=0.9,
             tag='G8_floor', store_ab=False)
    # N=62 (20 units), canonical label length
    units20 = ['b', 'a1', 'n', 'i3', 'h', 'ao3', 'zh', 'ong1', 'g', 'uo2',
               'b', 'ei3', 'j', 'ing1', 'sh', 'i4', 'd', 'a4', 'x', 'ue2']
    for fc in (0, 3):
        pipeline(seed=260 + fc, label=units20, m=2, d=5, t=150, fix_code=fc, c_cov=1e-3,
                 tag='G6_n62_fix%d' % fc, store_ab=(fc == 0))

    # ------------------------------------------------------------------ G5
    rng = np.random.default_rng(105)
    g5 = {}
    # (i) dense random HMM, N=7
    n, t = 7, 40
    A = rng.dirichlet(np.ones(n), size=n)
    pi = rng.dirichlet(np.ones(n))
    prob = rng.standard_normal((n, t)) * 5 - 20
    states = {i: 's%d' % i for i in range(n)}
    p, path = LHMM.viterbi(RecLog(), states, A, prob, pi)
    g5['dense_A'], g5['dense_pi'], g5['dense_prob'] = A, pi, prob
    g5['dense_point'], g5['dense_path'] = np.float64(p), path
    # (ii) constructed ties: integer log-probs, uniform A -> first-index tie-break
    n, t = 6, 25
    A = np.ones((n, n)) / n
    pi = np.ones(n) / n
    prob = rng.integers(-3, 0, size=(n, t)).astype(np.float64)
    states = {i: 's%d' % i for i in range(n)}
    p, path = LHMM.viterbi(RecLog(), states, A, prob, pi)
    g5['tie_A'], g5['tie_pi'], g5['tie_prob'] = A, pi, prob
    g5['tie_point'], g5['tie_path'] = np.float64(p), path
    # (iii) T = 1
    prob1 = prob[:, :1].copy()
    p, path = LHMM.viterbi(RecLog(), states, A, prob1, pi)
    g5['t1_prob'] = prob1
    g5['t1_point'], g5['t1_path'] = np.float64(p), path
    # (iv) left-right sentence HMM N=62, T=300 random B incl. zero / -inf rows and zeros in A
    n, t = 62, 300
    A = np.zeros((n, n))
    A[0, 1] = 1.
    for j in range(1, n - 1):
        A[j, j] = 0.5
        A[j, j + 1] = 0.5
    pi = np.ones(n) / n
    prob = rng.standard_normal((n, t)) * 4 - 60
    prob[0, :] = 0.
    prob[-1, :] = -np.inf
    states = {i: 'u%d' % ((i - 1) // 3) for i in range(n)}
    p, path = LHMM.viterbi(RecLog(), states, A, prob, pi)
    g5['lr_A'], g5['lr_pi'], g5['lr_prob'] = A, pi, prob
    g5['lr_point'], g5['lr_path'] = np.float64(p), path
    # (v) end_state_back=True (quirk Q9: stale max_index)
    p, path = LHMM.viterbi(RecLog(), states, A, prob, pi, end_state_back=True)
    g5['lr_esb_point'], g5['lr_esb_path'] = np.float64(p), path
    np.savez_compressed(os.path.join(HERE, 'G5_viterbi.npz'), **g5)

    # ------------------------------------------------------------------ G9
    unit_dir = os.path.join(scratch, 'unit_x')
    os.makedirs(unit_dir)
    h = hl[0]
    h.save_parameter(unit_dir)
    h.save_acc(unit_dir)
    for k in range(S - 2):
        h.profunction[1 + k].save_parameter(unit_dir)
        h.profunction[1 + k].save_acc(unit_dir)
    listing = []
    for root, dirs, files in os.walk(unit_dir):
        dirs.sort()
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(root, f), unit_dir)
            if f.endswith('.npy'):
                arr = np.load(os.path.join(root, f), allow_pickle=True)
                # timestamps in accumulator names are replaced by <ts>
                parts = rel.rsplit('_', 1)
                if parts[-1][:-4].isdigit():
                    rel = parts[0] + '_<ts>.npy'
                listing.append('%s|%s|%s' % (rel, arr.dtype, 'x'.join(map(str, arr.shape))))
            else:
                with open(os.path.join(root, f)) as fh:
                    listing.append('%s|text|%s' % (rel, fh.read().replace('\n', '\\n')))
    np.savez_compressed(os.path.join(HERE, 'G9_layout.npz'), listing=np.array(listing))
    for line in listing:
        print(line)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()

"""Pin the CPU oracle (oracle/poccala_oracle.py) to the reference's own outputs.

The golden vectors were produced by tests/golden/make_golden.py, which runs the
reference (util.py, Clustering.GMM, LHMM, AcousticModel helpers) in the build
container.  Every oracle function on the hot path is compared here; tolerance
1e-10 relative (both sides are float64, only the summation order differs).
"""
import numpy as np
import pytest

from oracle import poccala_oracle as po

RTOL = 1e-10
S = 5


def close(a, b, rtol=RTOL, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- G1
def test_g1_gaussian_q1_constant(golden):
    g = golden('G1_util')
    for d in (13, 39):
        y, mean, var = g['gauss_y_%d' % d], g['gauss_mean_%d' % d], g['gauss_var_%d' % d]
        out = np.array([po.gaussian_logpdf(y[i], mean[i], var[i]) for i in range(len(y))])
        close(out, g['gauss_out_%d' % d])
        # quirk Q1: NOT the textbook log-determinant
        textbook = -d / 2 * po.LOG_2PI - 0.5 * np.log(var).sum(1) - 0.5 * ((y - mean) ** 2 / var).sum(1)
        assert np.abs(textbook - g['gauss_out_%d' % d]).max() > 1e-3


def test_g1_log_sum_exp(golden):
    g = golden('G1_util')
    for i in range(5):
        ref = g['lse_out_%d' % i]
        got = po.lse(g['lse_in_%d' % i])
        if np.isinf(ref):
            assert got == ref            # quirk Q4: returns the max itself
        else:
            close(got, ref)
    close(po.lse_rows(g['lse_vec_in']), g['lse_vec_out'])
    close(po.matrix_lse(list(g['mlse_in']), 4), g['mlse_out_full'])
    close(po.matrix_lse(list(g['mlse_in']), 3), g['mlse_out_3'])


# ---------------------------------------------------------------- G2
@pytest.mark.parametrize('key', ['4_13', '8_39', '256_39'])
def test_g2_gmm_point_and_record(golden, key):
    g = golden('G2_gmm_point')
    out, rec = po.gmm_point(g['x_' + key], g['mean_' + key], g['var_' + key], g['w_' + key], record=True)
    close(out, g['out_' + key])
    close(rec, g['record_' + key])
    one = po.faithful_gmm_point(g['x_' + key][3], g['mean_' + key], g['var_' + key], g['w_' + key])
    close(one, g['out_' + key][3])


def test_g2_dimension_error(golden):
    g = golden('G2_gmm_point')
    with pytest.raises(ValueError):
        po.gmm_point(g['x_4_13'][:, :12], g['mean_4_13'], g['var_4_13'], g['w_4_13'])


# ---------------------------------------------------------------- G3
def test_g3_unit_observation(golden):
    g = golden('G3_unit_B')
    gmms = [(g['mean_%d' % k], g['var_%d' % k], g['w_%d' % k]) for k in range(3)]
    b = po.unit_observation(g['x'], gmms)
    assert b.shape == (5, 300)
    assert np.all(b[0] == 0.0) and np.all(np.isneginf(b[4]))
    close(b[1:4], g['B'][1:4])
    assert np.array_equal(np.isneginf(b), np.isneginf(g['B']))


# ---------------------------------------------------------------- G4..G8
CASES = ['G6_small_fix0', 'G6_small_fix1', 'G6_small_fix2', 'G6_small_fix3', 'G6_small_fix4',
         'G6_small_fix6', 'G8_floor', 'G6_n62_fix0', 'G6_n62_fix3']


def load_case(g):
    names = [str(u) for u in g['unit_names']]
    label = [str(u) for u in g['label']]
    model = {}
    for ui, u in enumerate(names):
        model[u] = dict(trans=g['trans_%d' % ui],
                        gmms=[(g['mean_%d_%d' % (ui, k)], g['var_%d_%d' % (ui, k)], g['w_%d_%d' % (ui, k)])
                              for k in range(S - 2)])
    return label, model


@pytest.mark.parametrize('case', CASES)
def test_g4_embedded(golden, case):
    g = golden(case)
    label, model = load_case(g)
    states, a, b, pi = po.score_label(g['x'], label, model)
    assert [states[i] for i in range(len(states))] == [str(s) for s in g['emb_states']]
    close(a, g['emb_A'])
    close(pi, g['emb_pi'])
    assert np.array_equal(np.isneginf(b), np.isneginf(g['emb_B']))
    fin = np.isfinite(b)
    close(b[fin], g['emb_B'][fin])


@pytest.mark.parametrize('case', CASES)
def test_g5_viterbi_on_sentence_hmm(golden, case):
    g = golden(case)
    label, model = load_case(g)
    point, path = po.viterbi(g['emb_A'], g['emb_pi'], g['emb_B'])
    assert np.array_equal(path, g['vit_path'])               # bit-exact given identical prob
    assert point == float(g['vit_point'])
    names = np.array([str(g['emb_states'][int(k)]) for k in path])
    assert np.array_equal(names, g['vit_path_conv'].astype(str))
    for u in set(label):
        runs = po.discriminate(u, names)
        assert len(runs) == int(g['disc_%s_n' % u])
        for ri, r in enumerate(runs):
            assert np.array_equal(r, g['disc_%s_%d' % (u, ri)])


def test_g5_viterbi_cases(golden):
    g = golden('G5_viterbi')
    for tag in ('dense', 'tie', 'lr'):
        point, path = po.viterbi(g[tag + '_A'], g[tag + '_pi'], g[tag + '_prob'])
        assert np.array_equal(path, g[tag + '_path']), tag
        assert point == float(g[tag + '_point']), tag
    point, path = po.viterbi(g['tie_A'], g['tie_pi'], g['t1_prob'])
    assert np.array_equal(path, g['t1_path']) and point == float(g['t1_point'])
    point, path = po.viterbi(g['lr_A'], g['lr_pi'], g['lr_prob'], end_state_back=True)   # quirk Q9
    assert np.array_equal(path, g['lr_esb_path']) and point == float(g['lr_esb_point'])


@pytest.mark.parametrize('case', CASES)
def test_g6_baum_welch(golden, case):
    g = golden(case)
    fix = int(g['fix_code'])
    bw = po.baum_welch(g['emb_A'], g['emb_pi'], [g['emb_B']], fix_code=fix)
    assert bw['n_pass'] == int(g['bw_n_pass'])
    assert bw['n_pass'] == (2 if fix & 1 else 3)             # quirk Q6
    np.testing.assert_allclose(bw['q_trace'][1:], g['bw_q_trace'][1:], atol=2e-6)   # logged with %f
    assert np.isneginf(bw['q_trace'][0])
    close(bw['logp'][0], g['bw_logp'])
    close(bw['pi'], g['bw_pi'], atol=1e-300)
    for name, got in (('bw_ksai', bw['ksai']), ('bw_gamma', bw['gamma'])):
        ref = g[name]
        assert np.array_equal(np.isneginf(got), np.isneginf(ref))
        fin = np.isfinite(ref)
        close(got[fin], ref[fin])
    if 'bw_alpha' in g.files:
        for name, got in (('bw_alpha', bw['alpha'][0]), ('bw_beta', bw['beta'][0])):
            ref = g[name]
            assert np.array_equal(np.isneginf(got), np.isneginf(ref))
            fin = np.isfinite(ref)
            close(got[fin], ref[fin])


@pytest.mark.parametrize('case', CASES)
def test_g7_g8_accumulators_and_mstep(golden, case):
    g = golden(case)
    fix = int(g['fix_code'])
    label, model = load_case(g)
    bw, accs, _ = po.estep_utterance(g['x'], label, model, fix_code=fix)
    for pos in range(len(label)):
        ua = accs[pos]
        for name, got in (('ksai_acc_%d' % pos, ua.ksai_acc), ('gamma_acc_%d' % pos, ua.gamma_acc)):
            ref = g[name]
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)), name
            fin = np.isfinite(ref)
            close(got[fin], ref[fin])
        for k in range(S - 2):
            acc = ua.gmm[k]
            for nm, key in (('acc', 'acc_%d_%d'), ('alpha_acc', 'alpha_acc_%d_%d'),
                            ('mean_acc', 'mean_acc_%d_%d'), ('cov_acc', 'cov_acc_%d_%d')):
                ref = g[key % (pos, k)]
                got = np.asarray(acc[nm])
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), key % (pos, k)
                fin = np.isfinite(ref)
                close(got[fin], ref[fin], rtol=1e-9)
        # M-step (A15)
        if not fix & 4:
            close(po.hmm_update_param(model[label[pos]]['trans'], ua.ksai_acc, ua.gamma_acc),
                  g['new_trans_%d' % pos], atol=1e-300)
        if not fix & 2:
            for k in range(S - 2):
                w, mean, var = po.gmm_update_param(ua.gmm[k], c_covariance=float(g['c_covariance']))
                close(w, g['new_w_%d_%d' % (pos, k)], rtol=1e-9)
                # mean = exp(.) - 100: absolute error of the exp() is amplified by the bias
                close(mean, g['new_mean_%d_%d' % (pos, k)], rtol=1e-7, atol=1e-9)
                close(var, g['new_var_%d_%d' % (pos, k)], rtol=1e-8)


def test_g8_floor_was_hit(golden):
    g = golden('G8_floor')
    assert (g['new_var_0_0'] == 0.9).any()


def test_faithful_forward_backward_matches_vectorised(golden):
    g = golden('G6_small_fix3')
    al, be = po.faithful_forward_backward(g['emb_A'], g['emb_pi'], g['emb_B'])
    fin = np.isfinite(g['bw_alpha'])
    close(al[fin], g['bw_alpha'][fin])
    fin = np.isfinite(g['bw_beta'])
    close(be[fin], g['bw_beta'][fin])


def test_g9_layout(golden):
    listing = [str(s) for s in golden('G9_layout')['listing']]
    assert 'HMM/transmat.npy|float64|5x5' in listing
    assert 'HMM/pi.npy|float64|5' in listing
    assert 'GMM_0/GMM_covariance.npy|float64|4x13x13' in listing     # full matrices, quirk Q2
    assert 'GMM_2/covariance-acc/GMM_covariance_acc_<ts>.npy|float64|4x13' in listing
    assert 'HMM/ksai-acc/ksai_acc_<ts>.npy|float64|3x5' in listing


@pytest.mark.parametrize('tag', ['free', 'pifixed'])
def test_g11_multi_utterance_lhmm(golden, tag):
    """An LHMM holding three utterances: merged xi/gamma/pi and Q over all utterances (LHMM.py:412-422,454-466)."""
    g = golden('G11_multi_utterance')
    bs = [g['B%d_%s' % (k, tag)] for k in range(3)]
    fix = 1 if tag == 'pifixed' else 0
    bw = po.baum_welch(g['A_' + tag], g['pi0_' + tag], bs, fix_code=fix)
    assert bw['n_pass'] == int(g['n_pass_' + tag])
    np.testing.assert_allclose(bw['q_trace'][1:], g['q_trace_' + tag][1:], atol=2e-6)
    close(np.ravel(bw['pi']), np.ravel(g['pi_' + tag]), atol=1e-300)   # the reference's pi becomes (1,N) here and sums to 3
    for name in ('ksai', 'gamma'):
        ref = g['%s_%s' % (name, tag)]
        assert np.array_equal(np.isneginf(bw[name]), np.isneginf(ref))
        close(bw[name][np.isfinite(ref)], ref[np.isfinite(ref)])
    for k in range(3):
        ref = g['alpha%d_%s' % (k, tag)]
        close(bw['alpha'][k][np.isfinite(ref)], ref[np.isfinite(ref)])
    # update_acc adds the MERGED statistics once per utterance (LHMM.py:484-496)
    ua = po.UnitAcc(8, [])
    po.update_acc(bw, bs, [None] * 3, [ua], [[]], fix_code=fix | 2, s=8)
    ref = g['ksai_acc_' + tag]
    close(ua.ksai_acc[np.isfinite(ref)], ref[np.isfinite(ref)])
    close(ua.gamma_acc, g['gamma_acc_' + tag])


# ------------------------------------------------------------------ G15: the edges (one / two / three frames, an utterance shorter than its label)
EDGE_TAGS = ['t1_l1', 't2_l1', 't3_l1', 't1_l2', 't2_l4']


def load_edge(g, tag):
    names = [str(u) for u in g[tag + '_unit_names']]
    label = [str(u) for u in g[tag + '_label']]
    flat = np.zeros((S, S))
    flat[0][1] = 1.
    for j in range(1, S - 1):
        flat[j][j] = flat[j][j + 1] = 0.5
    model = {u: dict(trans=flat, gmms=[(g['%s_mean_%d_%d' % (tag, ui, k)], g['%s_var_%d_%d' % (tag, ui, k)], g['%s_w_%d_%d' % (tag, ui, k)])
                                       for k in range(S - 2)]) for ui, u in enumerate(names)}
    return label, model


@pytest.mark.parametrize('tag', EDGE_TAGS)
def test_g15_edges(golden, tag):
    """What the reference does on very short utterances (tests/golden/make_golden_edges.py ran its worker sequence): with ONE frame
    LHMM.baulm_welch raises ValueError (the sum over t < T - 1 is empty) and every accumulator keeps its ln 0 -- the restatement's
    vectorised empty sums come out NaN there, which the GPU tests read as 'undefined in the reference'; with two or three frames, also
    for a label far longer than the utterance, the reference runs and the restatement reproduces it."""
    g = golden('G15_edges')
    label, model = load_edge(g, tag)
    x = g[tag + '_x']
    raised = str(g[tag + '_raised'])
    states, a, b, pi = po.score_label(x, label, model)
    close(a, g[tag + '_emb_A'])
    fin = np.isfinite(g[tag + '_emb_B'])
    assert np.array_equal(np.isfinite(b), fin)
    close(b[fin], g[tag + '_emb_B'][fin])
    if x.shape[0] == 1:
        assert raised == 'ValueError'
        for pos in range(len(label)):
            assert np.isneginf(g['%s_ksai_acc_%d' % (tag, pos)]).all() and np.isneginf(g['%s_gamma_acc_%d' % (tag, pos)]).all()
            for k in range(S - 2):
                assert np.isneginf(g['%s_acc_%d_%d' % (tag, pos, k)]).all() and np.isneginf(g['%s_alpha_acc_%d_%d' % (tag, pos, k)])
        with np.errstate(all='ignore'):
            bw = po.baum_welch(a, pi, [b])
        assert np.isnan(bw['ksai']).all()
        return
    assert raised == ''
    with np.errstate(all='ignore'):
        bw, accs, _ = po.estep_utterance(x, label, model)
    assert bw['n_pass'] == len(g[tag + '_q_trace'])
    np.testing.assert_allclose(bw['q_trace'][1:], g[tag + '_q_trace'][1:], atol=2e-6)
    close(bw['logp'][0], g[tag + '_logp'])
    close(bw['pi'], g[tag + '_pi'], atol=1e-300)
    for name, got in ((tag + '_ksai', bw['ksai']), (tag + '_gamma', bw['gamma'])):
        ref = g[name]
        assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.array_equal(np.isnan(got), np.isnan(ref)), name
        f = np.isfinite(ref)
        close(got[f], ref[f])
    for pos in range(len(label)):
        ua = accs[pos]
        for name, got in (('%s_ksai_acc_%d' % (tag, pos), ua.ksai_acc), ('%s_gamma_acc_%d' % (tag, pos), ua.gamma_acc)):
            ref = g[name]
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)), name
            f = np.isfinite(ref)
            close(got[f], ref[f])
        for k in range(S - 2):
            for nm in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
                ref = g['%s_%s_%d_%d' % (tag, nm, pos, k)]
                got = np.asarray(ua.gmm[k][nm])
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (nm, pos, k)
                f = np.isfinite(ref)
                close(got[f], ref[f], rtol=1e-9)


@pytest.mark.parametrize('gi', range(4))
def test_g15_far_frame_gmm_point(golden, gi):
    """GMM.point of the reference on a frame 30 sigma from the state's centre whose best mixture is ~90 nats above the first 32 (the case
    the matrix-pipe log-sum-exp of rounds 1-5 got wrong): the restatement reproduces the reference."""
    g = golden('G15_edges')
    got = po.gmm_point(g['far%d_x' % gi], g['far%d_mean' % gi], g['far%d_var' % gi], g['far%d_w' % gi])
    close(got, g['far%d_point' % gi], rtol=1e-12)


# ------------------------------------------------------------------ G16: the ill-conditioned model kinds the randomised GPU tests draw
@pytest.mark.parametrize('kind', ['tight', 'wide', 'skewed'])
def test_g16_ill_conditioned_kinds(golden, kind):
    """The reference's own E-step and M-step on mixtures at the 1e-6 variance floor, variances over four decades inside a state, weights
    down to 1e-12 (tests/golden/make_golden_kinds.py): the restatement reproduces every accumulator and the re-estimated model, so a GPU
    result 'held to the oracle' on such a draw (tests/test_gpu_fuzz_estep.py) is held to the reference."""
    g = golden('G16_kinds')
    label, model = load_edge(g, kind)
    x = g[kind + '_x']
    with np.errstate(all='ignore'):
        bw, accs, (_, a, b, pi) = po.estep_utterance(x, label, model)
    fin = np.isfinite(g[kind + '_emb_B'])
    assert np.array_equal(np.isfinite(b), fin)
    close(b[fin], g[kind + '_emb_B'][fin], rtol=1e-10)
    assert bw['n_pass'] == len(g[kind + '_q_trace'])
    close(bw['logp'][0], g[kind + '_logp'])
    for pos in range(len(label)):
        ua = accs[pos]
        for name, got in (('%s_ksai_acc_%d' % (kind, pos), ua.ksai_acc), ('%s_gamma_acc_%d' % (kind, pos), ua.gamma_acc)):
            ref = g[name]
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)), name
            f = np.isfinite(ref)
            close(got[f], ref[f])
        for k in range(S - 2):
            for nm in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
                ref = g['%s_%s_%d_%d' % (kind, nm, pos, k)]
                got = np.asarray(ua.gmm[k][nm])
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (nm, pos, k)
                f = np.isfinite(ref)
                close(got[f], ref[f], rtol=1e-9, atol=1e-9)
            with np.errstate(all='ignore'):
                w, mean, var = po.gmm_update_param(ua.gmm[k], c_covariance=1e-6)
            rw, rm, rv = g['%s_new_w_%d_%d' % (kind, pos, k)], g['%s_new_mean_%d_%d' % (kind, pos, k)], g['%s_new_var_%d_%d' % (kind, pos, k)]
            ok = np.isfinite(rm).all(axis=1) & np.isfinite(mean).all(axis=1)
            assert np.array_equal(np.isfinite(rm).all(axis=1), np.isfinite(mean).all(axis=1))
            close(w, rw, rtol=1e-9, atol=1e-300)
            close(mean[ok], rm[ok], rtol=1e-6, atol=1e-8)
            close(var[ok], rv[ok], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize('n', [2, 3, 4, 6])
def test_g15_end_state_back_on_small_hmms(golden, n):
    """LHMM.viterbi(end_state_back=True) of the reference on 2, 3, 4 and 6 states: with fewer than four, len(p_list) - 4 + argmax(p_list[-4:])
    is a negative index that NumPy wraps (LHMM.py:587-588); the restatement reproduces score and path."""
    g = golden('G15_edges')
    point, path = po.viterbi(g['esb%d_A' % n], g['esb%d_pi' % n], g['esb%d_prob' % n], end_state_back=True)
    assert point == float(g['esb%d_point' % n])
    assert np.array_equal(path, g['esb%d_path' % n])


def test_g15_zero_occupancy_mixture(golden):
    """A mixture of weight 0 collects nothing (its accumulators stay ln 0) and GMM.update_param of the reference makes 0 / 0 of it: weight 0,
    mean and variance NaN (Clustering.py:682-693).  The restatement reproduces both; the library keeps the mixture's mean and variance
    (tests/test_gpu_units.py::test_zero_occupancy_mixture_as_the_reference_has_it)."""
    g = golden('G15_edges')
    assert str(g['zero_raised_acc']) == '' and str(g['zero_raised_mstep']) == ''
    acc = dict(acc=np.full(4, -np.inf), alpha_acc=-np.inf, mean_acc=np.full((4, 5), -np.inf), cov_acc=np.full((4, 5), -np.inf))
    with np.errstate(all='ignore'):
        po.gmm_update_acc(acc, g['zero_lval'], g['zero_bval'], g['zero_x'], g['zero_mean'], g['zero_var'], g['zero_w'])
        w, mean, var = po.gmm_update_param(acc, c_covariance=1e-3)
    for nm, ref in (('acc', g['zero_acc']), ('mean_acc', g['zero_mean_acc']), ('cov_acc', g['zero_cov_acc'])):
        got = np.asarray(acc[nm])
        assert np.array_equal(np.isneginf(got), np.isneginf(ref)), nm
        f = np.isfinite(ref)
        close(got[f], ref[f], rtol=1e-10)
    assert np.isneginf(g['zero_acc'][2]) and g['zero_new_w'][2] == 0.0 and np.isnan(g['zero_new_mean'][2]).all() and np.isnan(g['zero_new_var'][2]).all()
    assert np.array_equal(np.isnan(mean), np.isnan(g['zero_new_mean'])) and np.array_equal(np.isnan(var), np.isnan(g['zero_new_var']))
    ok = ~np.isnan(g['zero_new_mean']).any(axis=1)
    close(w, g['zero_new_w'], rtol=1e-10, atol=1e-300)
    close(mean[ok], g['zero_new_mean'][ok], rtol=1e-8, atol=1e-9)
    close(var[ok], g['zero_new_var'][ok], rtol=1e-9)


def test_g15_impossible_utterance(golden):
    """A frame no state can emit (P(O) = 0): the reference does not raise -- it leaves the HMM accumulators at ln 0 and turns the GMM
    accumulators of every state of the label into NaN (inf - inf in the posteriors).  The restatement has the same patterns; the library
    adds nothing to either (tests/test_gpu_units.py::test_impossible_utterance_as_the_reference_has_it)."""
    g = golden('G15_edges')
    tag = 'p0'
    label, model = load_edge(g, tag)
    assert str(g[tag + '_raised']) == '' and len(g[tag + '_q_trace']) == 1
    x = g[tag + '_x']
    with np.errstate(all='ignore'):
        _, a, b, pi = po.score_label(x, label, model)
        b = g[tag + '_emb_B']                                  # (the fixture's emissions: one column set to ln 0)
        bw = po.baum_welch(a, pi, [b])
        accs = [po.UnitAcc(S, model[u]['gmms']) for u in label]
        po.update_acc(bw, [b], [x], accs, [model[u]['gmms'] for u in label], s=S)
    assert bw['n_pass'] == 1 and np.isneginf(bw['logp'][0])
    for pos in range(len(label)):
        for name, got in (('%s_ksai_acc_%d' % (tag, pos), accs[pos].ksai_acc), ('%s_gamma_acc_%d' % (tag, pos), accs[pos].gamma_acc)):
            ref = g[name]
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)), name
        for k in range(S - 2):
            for nm in ('acc', 'mean_acc', 'cov_acc'):
                ref = g['%s_%s_%d_%d' % (tag, nm, pos, k)]
                assert np.isnan(ref).all() and np.isnan(np.asarray(accs[pos].gmm[k][nm])).all(), (nm, pos, k)


def test_g17_the_reference_read_the_builds_tree_and_the_oracle_agrees_with_what_it_made_of_it(golden):
    """G17 (tests/golden/check_tree_with_reference.py): the build's host classes wrote a parameter tree and two batches' accumulator
    files, the REFERENCE read them (init_parameter / init_acc / update_param, LHMM.py:243-290,509-524, Clustering.py:297-367,682-693).
    The accumulators it merged are the log of the summed batch statistics, and the oracle's M-step on those sums gives the
    reference's new A, w, mu, sigma^2."""
    g = golden('G17_tree')
    units = [str(u) for u in g['units']]
    e = 3
    for ui, _ in enumerate(units):
        with np.errstate(divide='ignore'):
            ks = po.logaddexp_q4(g['batch0_ksai_%d' % ui], g['batch1_ksai_%d' % ui])
            ga = po.logaddexp_q4(g['batch0_gamma_%d' % ui], g['batch1_gamma_%d' % ui])
        np.testing.assert_allclose(g['ref_ksai_acc_%d' % ui], ks, rtol=1e-12)
        np.testing.assert_allclose(g['ref_gamma_acc_%d' % ui], ga, rtol=1e-12)
        np.testing.assert_allclose(g['new_trans_%d' % ui], po.hmm_update_param(g['trans_%d' % ui], ks, ga), rtol=1e-12, atol=1e-300)
        for k in range(e):
            j = ui * e + k
            with np.errstate(divide='ignore'):
                acc = {key: np.log(g['batch0_' + key][j] + g['batch1_' + key][j]) for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc')}
            np.testing.assert_allclose(g['ref_acc_%d_%d' % (ui, k)], acc['acc'], rtol=1e-12)
            np.testing.assert_allclose(g['ref_alpha_acc_%d_%d' % (ui, k)], acc['alpha_acc'], rtol=1e-12)
            w, mean, var = po.gmm_update_param(acc, c_covariance=float(g['c_covariance']))
            np.testing.assert_allclose(g['new_w_%d_%d' % (ui, k)], w, rtol=1e-10)
            np.testing.assert_allclose(g['new_mean_%d_%d' % (ui, k)], mean, rtol=1e-9, atol=1e-10)      # (exp(ln sum) - 100: the bias costs digits, Clustering.py:686)
            np.testing.assert_allclose(g['new_var_%d_%d' % (ui, k)], var, rtol=1e-9)
            assert (g['new_var_%d_%d' % (ui, k)] >= float(g['c_covariance'])).all()

"""The scaled linear-domain forward-backward (csrc/hmm_fb_linear.hip: value = m 2^e per lane, exp(B) split off the chain) against
the log-domain kernels it replaces on the default route (csrc/hmm_dp.hip, PCL_FB_LINEAR=0) and against the oracle
(LHMM.__forward_algorithm / __backward_algorithm / __maximization / baulm_welch, StatisticalModel/LHMM.py:335-471,526-544).

What only this route has: the packed emission word (16-bit power of two + 48 mantissa bits), int32 exponents with zero living
in the exponent, beta walked once, alpha stored only in the pass that can be the last (with a replay when another pass ends
the loop), ln alpha / ln beta made on demand, and the hand-over of out-of-range utterances to the log-domain kernels."""
import os

import numpy as np
import pytest

from _parity import hold
from oracle import poccala_oracle as po

pytestmark = pytest.mark.gpu
S = 5


@pytest.fixture(scope='module')
def eng():
    from poccala_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def run_fb(eng, labels, trans, Bs, fix_pi, linear, threshold=0.64, logpi=None):
    """forward-backward on given emissions for label-built sentence HMMs; returns every result of the pass loop."""
    from poccala_amd.engine import make_sentence_batch
    lens = np.array([b.shape[1] for b in Bs], dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    old = os.environ.get('PCL_FB_LINEAR')
    os.environ['PCL_FB_LINEAR'] = '1' if linear else '0'
    try:
        eng.load_frames(np.zeros((int(lens.sum()), eng.D), dtype=np.float32))
        b, n = make_sentence_batch(eng, labels, lens, begin, trans)
        if logpi is not None:
            from poccala_amd.engine import embedded_structure
            with np.errstate(divide='ignore'):
                b.set_transitions([np.log(embedded_structure(len(l), [trans[i] for i in l])[0]) for l in labels], logpi)
        b.set_emissions(Bs)
        b.forward_backward(fix_pi=fix_pi, threshold=threshold)
        out = {k: b.get(k) for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi', 'logp', 'npass', 'qtrace')}
        b.close()
    finally:
        if old is None:
            os.environ.pop('PCL_FB_LINEAR', None)
        else:
            os.environ['PCL_FB_LINEAR'] = old
    return out


def same(a, b, rtol, atol, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(np.isnan(a), np.isnan(b)), what + ': NaN pattern'
    ok = ~np.isnan(a)
    assert np.array_equal(np.isneginf(a[ok]), np.isneginf(b[ok])), what + ': -inf pattern'
    fin = np.isfinite(a) & ok
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol, err_msg=what)


def compare(x, y, rtol=1e-12, atol=1e-10, pi_rtol=1e-9):
    """two result sets of run_fb: log-domain quantities to rtol (of values ~1e2..1e4) + atol (an ulp of ln alpha ~ 2e4 is 3.6e-12,
    and the log-domain chain collects one per step)."""
    assert np.array_equal(x['npass'], y['npass'])
    same(x['qtrace'], y['qtrace'], rtol, atol, 'Q trace')
    same(x['logp'], y['logp'], rtol, atol, 'ln P(O)')
    for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma'):
        for u in range(len(x[k])):
            same(x[k][u], y[k][u], rtol, atol, '%s[%d]' % (k, u))
    for u in range(len(x['pi'])):
        np.testing.assert_allclose(x['pi'][u], y['pi'][u], rtol=pi_rtol, atol=1e-300)


def problem(seed, U, L, units=7, Ts=None, scale=4.0, offset=-85.0):
    from poccala_amd import synth
    rng = np.random.default_rng(seed)
    mean, var, w, trans = synth.make_model(units, 2, 13, seed=seed)
    labels = [list(rng.integers(0, units, L)) for _ in range(U)]
    Ts = Ts or [300] * U
    Bs = []
    for u in range(U):
        n = 3 * L + 2
        b = offset + scale * rng.standard_normal((n, Ts[u]))
        b[0] = 0.0                                   # VirtualState(1.): ln 1
        b[-1] = -np.inf                              # VirtualState(0.): ln 0
        Bs.append(b)
    return (mean, var, w), trans, labels, Bs


@pytest.mark.parametrize('fix_pi', [False, True])
@pytest.mark.parametrize('L', [20, 21, 40, 63, 84])
def test_linear_equals_log_domain(eng, fix_pi, L):
    """ragged lengths incl. T = 1 .. 9 (every tail of the 4-frame blocks) and the canonical 62 x 300; L = 21 .. 84 units: 65 .. 254
    states, the multi-wave chain against the log-domain multi-wave kernel."""
    model, trans, labels, Bs = problem(5, 12, L, Ts=[300, 1, 2, 3, 4, 5, 6, 7, 8, 9, 299, 150])
    eng.load_model(*model)
    lin = run_fb(eng, labels, trans, Bs, fix_pi, True)
    log = run_fb(eng, labels, trans, Bs, fix_pi, False)
    compare(lin, log)


@pytest.mark.parametrize('fix_pi', [False, True])
def test_linear_matches_oracle_wide_dynamic_range(eng, fix_pi):
    """emissions spread over thousands of nats inside one frame (aligned speech: the wrong states are hopeless): per-lane
    exponents keep the exact value of states far below the best one, as the reference's log domain does."""
    from poccala_amd.engine import embedded_structure
    model, trans, labels, Bs = problem(9, 6, 6, scale=900.0, offset=-3000.0, Ts=[120, 77, 300, 33, 64, 10])
    Bs[2][5, 40:44] = -np.inf                        # a state with impossible frames
    eng.load_model(*model)
    lin = run_fb(eng, labels, trans, Bs, fix_pi, True)
    for u, lab in enumerate(labels):
        a, pi = embedded_structure(len(lab), [trans[i] for i in lab])
        bw = po.baum_welch(a, pi, [Bs[u]], fix_code=1 if fix_pi else 0)
        tag = 'scaled forward-backward vs oracle (wide dynamic range)'
        assert lin['npass'][u] == bw['n_pass']
        hold(tag, 'ln alpha', lin['alpha'][u], bw['alpha'][0], 1e-10, 1e-9)
        hold(tag, 'ln beta', lin['beta'][u], bw['beta'][0], 1e-10, 1e-9)
        hold(tag, 'ln P(O)', lin['logp'][u], bw['logp'][0], 1e-10)
        hold(tag, 'ln xi (sum over t)', lin['ksai'][u], bw['ksai'], 1e-10, 1e-9)
        hold(tag, 'ln gamma (sum over t)', lin['gamma'][u], bw['gamma'], 1e-10, 1e-9)
        l = bw['alpha'][0] + bw['beta'][0]
        with np.errstate(invalid='ignore'):
            hold(tag, 'ln gamma_t(j)', lin['lgamma'][u], l - po.lse(l, axis=0)[None, :], 1e-10, 1e-9)
        np.testing.assert_allclose(lin['pi'][u], bw['pi'], rtol=1e-9, atol=1e-300)


def test_out_of_range_utterances_take_the_log_kernels(eng):
    """an emission outside the packed range (|ln b| >= 22000) or exponents that would leave int32 send THAT utterance to the
    log-domain kernels: its results equal the PCL_FB_LINEAR=0 run bit for bit, the other utterances stay on the scaled route."""
    model, trans, labels, Bs = problem(13, 5, 20, Ts=[300, 300, 300, 2600, 300])
    Bs[1][7, 100] = -3.0e7                           # a frame 1e3 sigma away
    Bs[3][:] = np.where(np.isfinite(Bs[3]), Bs[3] * 180.0, Bs[3])       # |ln b| ~ 15000: fits the word, but 2600 frames of it leave 2^26
    Bs[3][0] = 0.0
    eng.load_model(*model)
    lin = run_fb(eng, labels, trans, Bs, False, True)
    log = run_fb(eng, labels, trans, Bs, False, False)
    for u in (1, 3):
        for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi'):
            assert np.array_equal(lin[k][u], log[k][u], equal_nan=True), (k, u)
        assert lin['logp'][u] == log['logp'][u] and lin['npass'][u] == log['npass'][u]
    compare(lin, log)


def test_caller_logpi_and_pass_counts(eng):
    """a caller-supplied ln pi with zeros and tiny entries; thresholds that end the loop on pass 1, 2 (the replay with stores)
    and at the pass cap."""
    model, trans, labels, Bs = problem(21, 4, 5, Ts=[50, 60, 70, 80])
    eng.load_model(*model)
    rng = np.random.default_rng(3)
    lp = []
    for lab in labels:
        n = 3 * len(lab) + 2
        p = rng.dirichlet(np.ones(n))
        p[2] = 0.0
        p[3] = 1e-300
        with np.errstate(divide='ignore'):
            lp.append(np.log(p / p.sum()))
    for thr in (0.64, 1e9, -1.0, 1e-13):
        for fix in (False, True):
            lin = run_fb(eng, labels, trans, Bs, fix, True, threshold=thr, logpi=lp)
            log = run_fb(eng, labels, trans, Bs, fix, False, threshold=thr, logpi=lp)
            compare(lin, log)


def test_impossible_utterance(eng):
    """P(O) = 0 (a frame no state can emit): ln P(O) = -inf, posteriors NaN -- as the reference's arithmetic gives."""
    model, trans, labels, Bs = problem(31, 3, 4, Ts=[30, 30, 30])
    Bs[1][:, 11] = -np.inf
    eng.load_model(*model)
    lin = run_fb(eng, labels, trans, Bs, False, True)
    log = run_fb(eng, labels, trans, Bs, False, False)
    assert np.isneginf(lin['logp'][1]) and np.isneginf(log['logp'][1])
    compare(lin, log)


@pytest.mark.parametrize('fix_pi', [False, True])
@pytest.mark.parametrize('wide', [False, True])
def test_left_to_right_hmms_of_any_size_match_the_oracle(eng, fix_pi, wide):
    """left-to-right HMMs that are NOT built from labels: N = 2 .. 64 states (the wave's last lane included), random self-loop /
    advance probabilities, a state without a self-loop, a dead end, random pi with zeros, -inf emissions sprinkled in -- against
    the oracle's baum_welch (LHMM.py:335-471, 526-544) at 1e-10, and equal to the log-domain kernels.  wide: up to 256 states in one
    batch with short ones (the chain spread over 2..4 wavefronts, hmm_fb_linear_mw.inc: values crossing a wave boundary through LDS,
    whole waves without a state)."""
    rng = np.random.default_rng(77)
    sizes = [(2, 9), (3, 1), (5, 40), (14, 33), (63, 120), (64, 300), (64, 2), (31, 7)]
    if wide:
        sizes = [(65, 50), (122, 300), (128, 33), (129, 20), (200, 64), (256, 40), (7, 12), (64, 31), (192, 1), (193, 2)]
    As, pis, Bs = [], [], []
    for n, t in sizes:
        a = np.zeros((n, n))
        for i in range(n - 1):
            x = rng.uniform(0.1, 0.9)
            a[i, i], a[i, i + 1] = x, 1.0 - x
        a[n - 1, n - 1] = 1.0
        if n > 4:
            a[2, 2], a[2, 3] = 0.0, 1.0                     # no self-loop
            a[n - 1, n - 1] = 0.0                           # the last state leads nowhere (as the exit state of a sentence HMM)
        p = rng.dirichlet(np.ones(n))
        if n > 3:
            p[1] = 0.0
            p /= p.sum()
        b = -60.0 + 25.0 * rng.standard_normal((n, t))
        b[rng.random((n, t)) < 0.03] = -np.inf
        b[0, 0] = -50.0                                     # (at least one way in)
        As.append(a); pis.append(p); Bs.append(b)
    res = {}
    for linear in (True, False):
        old = os.environ.get('PCL_FB_LINEAR')
        os.environ['PCL_FB_LINEAR'] = '1' if linear else '0'
        try:
            b = eng.batch([s[0] for s in sizes], [s[1] for s in sizes])
            with np.errstate(divide='ignore'):
                b.set_transitions([np.log(a) for a in As], [np.log(p) for p in pis])
            b.set_emissions(Bs)
            b.forward_backward(fix_pi=fix_pi)
            res[linear] = {k: b.get(k) for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi', 'logp', 'npass', 'qtrace')}
            b.close()
        finally:
            if old is None:
                os.environ.pop('PCL_FB_LINEAR', None)
            else:
                os.environ['PCL_FB_LINEAR'] = old
    compare(res[True], res[False])
    lin = res[True]
    tag = 'scaled forward-backward vs oracle (left-to-right HMMs of 2..64 states)'
    for u in range(len(sizes)):
        with np.errstate(all='ignore'):
            bw = po.baum_welch(As[u], pis[u], [Bs[u]], fix_code=1 if fix_pi else 0)
        assert lin['npass'][u] == bw['n_pass'], u
        if not np.isfinite(bw['logp'][0]):
            assert not np.isfinite(lin['logp'][u])
            continue
        hold(tag, 'ln alpha', lin['alpha'][u], bw['alpha'][0], 1e-10, 1e-9)
        hold(tag, 'ln beta', lin['beta'][u], bw['beta'][0], 1e-10, 1e-9)
        hold(tag, 'ln P(O)', lin['logp'][u], bw['logp'][0], 1e-10)
        if sizes[u][1] > 1:
            fk = np.isfinite(bw['ksai'])
            assert np.array_equal(np.isfinite(lin['ksai'][u]), fk), u
            hold(tag, 'ln xi (sum over t)', lin['ksai'][u][fk], bw['ksai'][fk], 1e-10, 1e-9)
            hold(tag, 'ln gamma (sum over t)', lin['gamma'][u], bw['gamma'], 1e-10, 1e-9)
        np.testing.assert_allclose(lin['pi'][u], bw['pi'], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize('n', [2, 3, 4, 6])
def test_viterbi_end_state_back_on_small_hmms_as_the_reference_has_it(eng, golden, n):
    """Golden G15: LHMM.viterbi(end_state_back=True) of the reference itself on 2, 3, 4 and 6 states (a negative index wrapped by NumPy when
    there are fewer than four, LHMM.py:587-588): score and path through the C-ABI, bit for bit."""
    g = golden('G15_edges')
    A, pi, prob = g['esb%d_A' % n], g['esb%d_pi' % n], g['esb%d_prob' % n]
    eng.load_frames(np.zeros((prob.shape[1], max(eng.D, 1)), dtype=np.float32))
    b = eng.batch([n], [prob.shape[1]], [0])
    with np.errstate(divide='ignore'):
        b.set_transitions([np.log(A)], [np.log(pi)])
    b.set_emissions([prob])
    b.viterbi(end_state_back=True)
    assert b.get('point')[0] == float(g['esb%d_point' % n])
    assert np.array_equal(b.get('path')[0].astype(np.float64), g['esb%d_path' % n])
    b.close()

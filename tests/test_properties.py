"""Property tests of the oracle (hypothesis, CPU): size-independent identities the domain offers
(SURVEY section 4).  The same identities are asserted on the GPU at BASELINE sizes in test_gpu_parity.py."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import poccala_oracle as po


def random_hmm(rng, n, t, sparse):
    a = rng.dirichlet(np.ones(n), size=n)
    if sparse:
        a = np.triu(a) * (np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= 1)
        a[-1, -1] = 1.0
        a /= a.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    b = rng.standard_normal((n, t)) * 3 - 20
    return a, pi, b


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10 ** 6), st.integers(2, 12), st.integers(2, 30), st.booleans())
def test_forward_backward_identities(seed, n, t, sparse):
    rng = np.random.default_rng(seed)
    a, pi, b = random_hmm(rng, n, t, sparse)
    al, be = po.forward(a, pi, b), po.backward(a, b)
    logp = po.lse(al[:, -1])
    # P(O) from any time slice
    for k in (0, t // 2, t - 1):
        assert abs(po.lse(al[:, k] + be[:, k]) - logp) <= 1e-9 * abs(logp)
    with np.errstate(divide='ignore'):
        assert abs(po.lse(np.log(pi) + b[:, 0] + be[:, 0]) - logp) <= 1e-9 * abs(logp)
    ksai, gamma, log_pi = po.xi_gamma_pi(a, b, al, be)
    # sum_j xi_t(i,j) = gamma_t(i) summed over t < T-1 (un-normalised, quirk Q5)
    row = po.lse(ksai, axis=1)
    fin = np.isfinite(gamma)
    np.testing.assert_allclose(row[fin], gamma[fin], rtol=1e-9)
    assert abs(po.lse(log_pi)) < 1e-9
    # Viterbi score <= forward score, and its path has that score
    point, path = po.viterbi(a, pi, b)
    assert point <= logp + 1e-9
    with np.errstate(divide='ignore'):
        la = np.log(a)
        s = np.log(pi[int(path[0])]) + b[int(path[0]), 0]
        for k in range(1, t):
            s = (s + la[int(path[k - 1]), int(path[k])]) + b[int(path[k]), k]
    assert abs(s - point) <= 1e-9 * max(1.0, abs(point))


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10 ** 6), st.integers(1, 6), st.integers(1, 9), st.integers(1, 20))
def test_gmm_statistics_identities(seed, m, d, t):
    """sum_m gamma_t(j,m) = gamma_t(j); the accumulators are permutation invariant in the frames."""
    rng = np.random.default_rng(seed)
    mean, var, w = rng.standard_normal((m, d)), rng.uniform(0.5, 2, (m, d)), rng.dirichlet(np.ones(m))
    x = rng.standard_normal((t, d))
    lg = np.log(rng.uniform(0.01, 1.0, t))
    lb = po.gmm_point(x, mean, var, w)
    acc = po.UnitAcc(5, [(mean, var, w)] * 3).gmm[0]
    po.gmm_update_acc(acc, lg, lb, x, mean, var, w)
    np.testing.assert_allclose(po.lse(acc['acc']), acc['alpha_acc'], rtol=1e-9, atol=1e-9)
    perm = rng.permutation(t)
    acc2 = po.UnitAcc(5, [(mean, var, w)] * 3).gmm[0]
    po.gmm_update_acc(acc2, lg[perm], lb[perm], x[perm], mean, var, w)
    for k in ('acc', 'mean_acc', 'cov_acc'):
        np.testing.assert_allclose(acc2[k], acc[k], rtol=1e-9, atol=1e-9)


@settings(max_examples=30, deadline=None)
@given(st.lists(st.floats(-700, 700), min_size=1, max_size=20))
def test_lse_bounds(v):
    v = np.array(v)
    out = po.lse(v)
    assert v.max() <= out + 1e-12 and out <= v.max() + np.log(len(v)) + 1e-9


def test_lse_inf_semantics():
    assert po.lse(np.array([-np.inf, -np.inf])) == -np.inf              # quirk Q4
    assert po.lse(np.array([1.0, np.inf])) == np.inf
    assert po.lse(np.array([-np.inf, 0.0])) == 0.0

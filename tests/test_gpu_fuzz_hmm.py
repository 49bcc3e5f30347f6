"""Randomised HMMs on GIVEN emissions against the oracle (round 5): the drop-in LHMM takes ANY transition matrix (LHMM.py:26-120), not
only the sentence HMMs AcousticModel.embedded builds.  Per case a batch of 1-12 HMMs, each with its own state count (3 .. 150: one and
several wavefronts), structure (ergodic, banded, left-to-right with skips, sparse random; entry / exit rows as the caller likes), initial
distribution (with exact zeros), utterance length (1 .. 150) and emissions (a few nats to thousands of nats of spread, -inf entries,
now and then a frame nobody can emit).  Held to oracle/poccala_oracle.py: the pass loop (LHMM.py:526-544) with its pass count and Q trace,
ln alpha, ln beta, ln P(O), the summed ln xi / ln gamma, ln gamma_t(j), the re-estimated pi; the Viterbi path and score bit for bit
(LHMM.py:546-609) on the same operands.

   python tests/test_gpu_fuzz_hmm.py [cases] [first seed]"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _parity import hold  # noqa: E402
from oracle import poccala_oracle as po  # noqa: E402

pytestmark = pytest.mark.gpu


def draw_hmm(rng, many_states=False):
    n = int(rng.choice([3, 4, 5, 8, 14, 33, 62, 64, 65, 100, 150])) if rng.random() < 0.7 else int(rng.integers(3, 151))
    if many_states:                                  # up to the 1024-state limit: four and more wavefronts per HMM, the log-domain kernels beyond 256 states
        n = int(rng.choice([192, 193, 255, 256, 257, 320, 511, 512, 700, 1024]))
    kind = rng.choice(['ergodic', 'banded', 'left-right', 'sparse'])
    a = np.zeros((n, n))
    if kind == 'ergodic':
        a = rng.dirichlet(np.ones(n), size=n)
    elif kind == 'banded':
        wdt = int(rng.integers(1, 4))
        for i in range(n):
            lo, hi = max(0, i - wdt), min(n, i + wdt + 1)
            a[i, lo:hi] = rng.dirichlet(np.ones(hi - lo))
    elif kind == 'left-right':
        skip = int(rng.integers(1, 4))
        for i in range(n):
            hi = min(n, i + skip + 1)
            a[i, i:hi] = rng.dirichlet(np.ones(hi - i))
    else:
        for i in range(n):
            k = int(rng.integers(1, min(n, 6) + 1))
            a[i, rng.choice(n, size=k, replace=False)] = rng.dirichlet(np.ones(k))
    if rng.random() < 0.3:                           # a state nobody leaves towards (an exit state as the reference's sentence HMMs have)
        a[-1] = 0.0
    pi = rng.dirichlet(np.ones(n))
    if rng.random() < 0.6:                           # exact zeros in pi (the reference's sentence HMMs: all mass on the entry state)
        z = rng.random(n) < 0.7
        z[int(rng.integers(0, n))] = False
        pi[z] = 0.0
        pi /= pi.sum()
    return a, pi, str(kind)


def draw(seed):
    rng = np.random.default_rng(seed)
    long_ = 5000 <= seed < 9000                      # seeds from 5000: a few long utterances (the scaled route's exponents, the hand-over to the log kernels)
    many = seed >= 9000                              # seeds from 9000: HMMs of 192 .. 1024 states
    U = int(rng.integers(1, 13)) if not (long_ or many) else int(rng.integers(1, 4))
    hmms, Bs = [], []
    for _ in range(U):
        a, pi, kind = draw_hmm(rng, many)
        n = a.shape[0]
        T = int(rng.integers(1, 151)) if rng.random() < 0.9 else int(rng.integers(1, 5))
        if long_:
            T = int(rng.integers(800, 5000))
        if many:
            T = int(rng.integers(1, 120))
        spread = float(rng.choice([1.0, 4.0, 40.0, 900.0]))
        b = float(rng.choice([0.0, -85.0, -3000.0])) + spread * rng.standard_normal((n, T))
        if rng.random() < 0.4:
            b[rng.random((n, T)) < 0.1] = -np.inf    # states that cannot emit some frames
        if rng.random() < 0.2:
            b[0] = 0.0                               # VirtualState(1.) / VirtualState(0.) rows as AcousticModel.embedded makes them
            b[-1] = -np.inf
        if rng.random() < 0.05:
            b[:, int(rng.integers(0, T))] = -np.inf  # a frame nobody can emit: P(O) = 0
        hmms.append((a, pi, kind))
        Bs.append(b)
    return dict(hmms=hmms, Bs=Bs, fix_pi=bool(rng.random() < 0.4), threshold=float(rng.choice([0.64, 0.64, 0.05, 1e9])), end_state_back=bool(rng.random() < 0.3))


def same(tag, what, got, want, rtol, atol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (tag, what, got.shape, want.shape)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan), '%s / %s: NaN pattern' % (tag, what)
    hold(tag, what, np.where(nan, 0.0, got), np.where(nan, 0.0, want), rtol, atol)


def run_case(eng, seed):
    c = draw(seed)
    U = len(c['hmms'])
    N = np.array([h[0].shape[0] for h in c['hmms']], dtype=np.int32)
    T = np.array([b.shape[1] for b in c['Bs']], dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(T[:-1])]).astype(np.int64)
    eng.load_frames(np.zeros((int(T.sum()), max(eng.D, 1)), dtype=np.float32))
    b = eng.batch(N, T, begin)
    with np.errstate(divide='ignore'):
        b.set_transitions([np.log(h[0]) for h in c['hmms']], [np.log(h[1]) for h in c['hmms']])
    b.set_emissions(c['Bs'])
    b.forward_backward(fix_pi=c['fix_pi'], threshold=c['threshold'])
    b.viterbi(end_state_back=c['end_state_back'])
    out = {k: b.get(k) for k in ('alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi', 'logp', 'npass', 'qtrace', 'path', 'point')}
    b.close()
    tag = 'hmm fuzz'
    for u in range(U):
        a, pi, kind = c['hmms'][u]
        B = c['Bs'][u]
        with np.errstate(all='ignore'):
            bw = po.baum_welch(a, pi, [B], fix_code=1 if c['fix_pi'] else 0, threshold=c['threshold'])
            rp, rpath = po.viterbi(a, pi, B, end_state_back=c['end_state_back'])
        ctx = (seed, u, kind, int(N[u]), int(T[u]))
        # sums of up to 150 emissions of this size in float64 on both sides, in different orders
        big = float(np.abs(B[np.isfinite(B)]).max()) * T[u] if np.isfinite(B).any() else 1.0
        at = max(1e-9, 32 * 2.2e-16 * big * np.sqrt(T[u]))     # (the log-domain oracle rounds once per frame at the size of ln alpha ~ |b| T)
        if bw['n_pass'] > 16:                        # PCL_MAX_PASS (include/poccala_hip.h): the reference has no cap, and at its threshold (0.64 nats
            assert int(out['npass'][u]) == 16, ctx   # of gain from re-estimating pi alone) never gets near it; a caller's tiny threshold can
            continue
        assert int(out['npass'][u]) == int(bw['n_pass']), ctx
        if np.isneginf(bw['logp'][0]) or np.isnan(bw['logp'][0]):
            assert np.isneginf(out['logp'][u]) or np.isnan(out['logp'][u]), ctx      # P(O) = 0: the posteriors are the reference's NaNs (test_impossible_utterance)
            continue
        same(tag, 'ln P(O)', out['logp'][u], bw['logp'][0], 1e-10, at)
        same(tag, 'ln alpha', out['alpha'][u], bw['alpha'][0], 1e-10, at)
        same(tag, 'ln beta', out['beta'][u], bw['beta'][0], 1e-10, at)
        if T[u] > 1:                                 # one frame: the reference raises (golden G15; the restatement's empty sums come out NaN); the library says ln 0
            same(tag, 'ln xi (sum over t)', out['ksai'][u], bw['ksai'], 1e-10, at)
            same(tag, 'ln gamma (sum over t)', out['gamma'][u], bw['gamma'], 1e-10, at)
        else:
            assert np.isneginf(out['ksai'][u]).all(), ctx
        l = bw['alpha'][0] + bw['beta'][0]
        if T[u] > 1:
            with np.errstate(all='ignore'):
                same(tag, 'ln gamma_t(j)', out['lgamma'][u], l - po.lse(l, axis=0)[None, :], 1e-10, at)
        else:
            assert np.isneginf(out['lgamma'][u]).all(), ctx       # (the reference raises on a one-frame utterance, golden G15: no occupancies)
        np.testing.assert_allclose(out['pi'][u], bw['pi'], rtol=max(1e-9, 100 * at), atol=1e-300, err_msg=str(ctx))
        assert np.array_equal(out['path'][u].astype(np.float64), rpath) and (rp == out['point'][u] or (np.isnan(rp) and np.isnan(out['point'][u]))), ctx
    return c


@pytest.fixture(scope='module')
def eng():
    from poccala_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize('seed', list(range(30)) + [5000, 5001, 5002, 9001, 9002])
def test_random_hmms_against_the_oracle(eng, seed):
    run_case(eng, seed)


if __name__ == '__main__':
    from poccala_amd import Engine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    e = Engine(0)
    failed = 0
    for s in range(first, first + n):
        try:
            c = run_case(e, s)
            print('seed %d ok  (%s)' % (s, ', '.join('%s %dx%d' % (h[2], h[0].shape[0], b.shape[1]) for h, b in zip(c['hmms'], c['Bs']))[:150]), flush=True)
        except Exception as ex:          # noqa: BLE001 -- a sweep: report and go on
            failed += 1
            print('seed %d FAILED: %s' % (s, str(ex).splitlines()[0][:300] if str(ex) else repr(ex)), flush=True)
            e.close()
            e = Engine(0)
    print('%d cases, %d failed' % (n, failed))
    sys.exit(1 if failed else 0)

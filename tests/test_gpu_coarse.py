"""The coarse pass over the off-pipe mixtures (csrc/gmm_score_coarse.hip, round 6): a bound evaluated on the matrix pipe rules out almost every
(frame, mixture) pair of a tight mixture, the pairs it cannot rule out are evaluated exactly, the result is log-added to the pipe's.

What must hold: (1) no pair that matters is ever ruled out -- ln b against the oracle on models whose mixtures span six decades of variance,
with frames sitting exactly on collapsed mixtures, near them, and far from everything; (2) the bound does rule out nearly everything (the
count of exactly evaluated pairs); (3) frames whose scaled features leave the f16 range fall back to the direct-form kernel; (4) states without
a single on-pipe mixture start from a threshold of -inf and settle; (5) zero-weight and padding mixtures never surface."""
import os

import numpy as np
import pytest

from _parity import hold, note
from oracle import poccala_oracle as po

pytestmark = pytest.mark.gpu

F32_LOGLIK_ATOL = 5e-5


def _engine(**env):
    from poccala_amd import Engine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _bound(mean, var, w, x):
    from _oracle_pool import f32_evaluation_bound_rows
    return f32_evaluation_bound_rows([(mean[j], var[j], w[j]) for j in range(mean.shape[0])], x)


def tight_model(seed, J, M, D, shares):
    """state j: shares[j] of its mixtures get variances log-uniform in [1e-6, 0.2] per mixture (all features alike within a factor 4), the
    rest are the bench model's (variance 0.5 .. 2); a few weights are exactly zero."""
    from poccala_amd import synth
    rng = np.random.default_rng(seed)
    mean, var, w, _ = synth.make_model((J + 2) // 3, M, D, seed=seed)
    mean, var, w = mean[:J].copy(), var[:J].copy(), w[:J].copy()
    tight = []
    for j in range(J):
        n = int(round(shares[j] * M))
        idx = np.sort(rng.choice(M, n, replace=False))
        level = 10.0 ** rng.uniform(-6, np.log10(0.2), n)
        var[j, idx] = level[:, None] * rng.uniform(0.5, 2.0, (n, D))
        tight.append(idx)
        zero = rng.choice(M, 3, replace=False)
        w[j, zero] = 0.0
        w[j] /= w[j].sum()
    return mean, var, w, tight


def frames_for(rng, mean, var, tight, per_state):
    """for every state: frames sampled from its tight mixtures (a third exactly ON a mean, rounded to f32), from its broad mixtures, and
    N(0,1) noise; returns (x (T,D) f32, owner state (T,))"""
    J, M, D = mean.shape
    xs, own = [], []
    for j in range(J):
        n = per_state
        comp = rng.integers(0, M, n)
        if len(tight[j]):
            comp[: 2 * n // 3] = rng.choice(tight[j], 2 * n // 3)
        x = mean[j, comp] + np.sqrt(var[j, comp]) * rng.standard_normal((n, D))
        x[: n // 3] = mean[j, comp[: n // 3]]                        # exactly on the mean
        k = max(1, n // 6)
        x[n - k:] = rng.standard_normal((k, D))
        xs.append(x)
        own += [j] * n
    return np.concatenate(xs).astype(np.float32), np.array(own)


def score_all(eng, mean, var, w, x):
    from poccala_amd import PCL_F32
    J = mean.shape[0]
    eng.load_model(mean, var, w)
    eng.load_frames(x)
    b = eng.batch([J + 2], [len(x)], [0])
    b.set_states([np.concatenate([[-1], np.arange(J), [-2]]).astype(np.int32)])
    b.score(PCL_F32)
    got = b.get('B')[0][1:-1]
    b.close()
    return got


def check(tag, got, mean, var, w, x):
    with np.errstate(divide='ignore'):
        ref = np.stack([po.gmm_point(x.astype(np.float64), mean[j], var[j], w[j]) for j in range(mean.shape[0])])
    bound = _bound(mean, var, w, x)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    return hold(tag, 'ln b', got[fin], ref[fin], 5e-6, (F32_LOGLIK_ATOL + bound)[fin])


@pytest.mark.parametrize('seed,D,passes', [(11, 39, 1), (12, 39, 1), (13, 13, 1), (14, 26, 1), (11, 39, 3), (13, 13, 3)])
def test_coarse_pass_against_the_oracle_over_six_decades_of_variance(seed, D, passes):
    rng = np.random.default_rng(seed)
    J, M = 8, 96
    shares = [0.0, 0.05, 0.3, 0.5, 0.7, 0.84, 0.3, 0.6]
    mean, var, w, tight = tight_model(seed, J, M, D, shares)
    x, own = frames_for(rng, mean, var, tight, 300)
    eng = _engine(PCL_COARSE_STATS=1, PCL_COARSE_PASSES=passes)     # (1: one f16 product per term, the default; 3: the two-piece operands' three)
    try:
        eng.enable_timing(True)
        got = score_all(eng, mean, var, w, x)
        n_off, limit = eng.model_split_info()
        assert limit == int(np.float32(0.99) * np.float32(M)) and (n_off <= limit).all() and n_off[0] == 0 and (n_off[1:] > 0).all()
        assert eng.kernel_time('score_coarse')[1] == 1 and eng.kernel_time('score_subset')[1] == 0 and eng.kernel_time('score_direct')[1] == 0
        exact = eng.coarse_pairs()
        all_pairs = int(n_off.sum()) * len(x)
        note('coarse pass, seed %d D=%d, %d product(s)' % (seed, D, passes), 'pairs evaluated exactly / pairs of off-pipe mixtures', [exact, all_pairs])
        # frames ON a collapsed mixture and within a few sigma of a tight one do pass -- that is a third of the frames times one mixture or so;
        # the rest of the pairs must have been ruled out by the bound
        assert 0 < exact < 0.05 * all_pairs, (exact, all_pairs)
        m = check('coarse pass D=%d' % D, got, mean, var, w, x)
        # the frames sitting exactly on a collapsed mean are dominated by ONE off-pipe mixture: held there on their own
        j_on = [(j, t) for t, j in enumerate(own) if (t % 300) < 100 and len(tight[j])]
        ref_on = np.array([po.gmm_point(x[t:t + 1].astype(np.float64), mean[j], var[j], w[j])[0] for j, t in j_on])
        got_on = np.array([got[j, t] for j, t in j_on])
        b_on = _bound(mean, var, w, x)
        hold('coarse pass D=%d' % D, 'ln b of frames ON an off-pipe mean (own state)', got_on, ref_on, 5e-6, np.array([F32_LOGLIK_ATOL + b_on[j, t] for j, t in j_on]))
    finally:
        eng.close()
    # the round 4-5 route (every off-pipe mixture in direct form) on the same input: both sum exact terms
    e0 = _engine(PCL_COARSE=0, PCL_SPLIT_MAX=0.85)
    try:
        got0 = score_all(e0, mean, var, w, x)
        fin = np.isfinite(got0)
        # (the direct-form route evaluates its mixtures from f32 roundings of the parameters -- the analytical bound of that is part of the
        #  allowance; the coarse route's exact part is float64 from the master copy)
        hold('coarse pass D=%d' % D, 'ln b against the direct-form subset route', got[fin], got0[fin], 5e-6, (2 * F32_LOGLIK_ATOL + _bound(mean, var, w, x))[fin])
    finally:
        e0.close()


def test_states_without_a_single_on_pipe_mixture_and_frames_out_of_the_f16_range():
    """PCL_COARSE_SPLIT_MAX=1: every state stays split, also the ones ALL of whose mixtures are off the pipe (the pipe writes -inf, the
    threshold starts at -inf and is raised by the first exact values); a block of frames carries an offset of 40 -- their scaled features
    leave the f16 range, the tile is flagged and rescored by the direct-form subset kernel in the same call."""
    rng = np.random.default_rng(21)
    J, M, D = 6, 64, 39
    mean, var, w, tight = tight_model(21, J, M, D, [1.0, 1.0, 0.9, 0.5, 1.0, 0.2])
    x, own = frames_for(rng, mean, var, tight, 384)
    x[500:564] += 40.0                                                  # (one 64-frame stretch, inside one tile of 256)
    eng = _engine(PCL_COARSE_SPLIT_MAX=1.0, PCL_COARSE_STATS=1)
    try:
        got = score_all(eng, mean, var, w, x)
        n_off, limit = eng.model_split_info()
        assert limit == M and n_off[0] == M and n_off[1] == M and n_off[4] == M
        exact = eng.coarse_pairs()
        note('coarse pass, all-tight states', 'pairs evaluated exactly / pairs of off-pipe mixtures', [exact, int(n_off.sum()) * len(x)])
        check('coarse pass, states with no on-pipe mixture, frames out of range', got, mean, var, w, x)
    finally:
        eng.close()


def test_the_pass_gives_up_where_its_reference_says_nothing():
    """A state ALL of whose 256 mixtures have collapsed (variance 1e-6), kept on the coarse route by PCL_COARSE_SPLIT_MAX=1, and frames far
    from every one of them: ln b ~ -4e7, the threshold sits at the end of the f16 range and the bound under the floored variance lets every
    pair through.  A wave that has evaluated more than max(4096, 2 x 256) pairs gives the tile up; the direct-form subset kernel rescored
    it in the same call.  Same results against the oracle; the tiles of the other states are not given up."""
    rng = np.random.default_rng(41)
    J, M, D = 3, 256, 39
    mean, var, w, tight = tight_model(41, J, M, D, [1.0, 0.9, 0.3])
    var[0] = 1e-6 * rng.uniform(0.5, 2.0, (M, D))
    x, own = frames_for(rng, mean, var, tight, 512)
    eng = _engine(PCL_COARSE_SPLIT_MAX=1.0, PCL_COARSE_STATS=1)
    try:
        got = score_all(eng, mean, var, w, x)
        n_off, limit = eng.model_split_info()
        assert limit == M and n_off[0] == M and 0 < n_off[1] < M and 0 < n_off[2] < M
        exact, given_up = eng.coarse_counters()
        tiles = 3 * ((len(x) + 255) // 256)
        note('coarse pass, a state it gives up on', '[pairs evaluated exactly, tiles given up, tiles]', [exact, given_up, tiles])
        assert 0 < given_up <= tiles // 3, (given_up, tiles)                  # (state 0's tiles, and only those)
        check('coarse pass, tiles given up', got, mean, var, w, x)
    finally:
        eng.close()
    # without the counters (the default): the same bits
    eng = _engine(PCL_COARSE_SPLIT_MAX=1.0)
    try:
        assert np.array_equal(score_all(eng, mean, var, w, x), got)
    finally:
        eng.close()


def test_coarse_pass_inside_the_e_step_matches_the_direct_form_route():
    """score -> forward-backward -> accumulate on a label batch whose states have 30-80 % off-pipe mixtures: ln P(O) and the four statistics
    from the coarse route against the round 4-5 route (PCL_COARSE=0), 1e-6 relative (the emissions differ by f32-class rounding only)."""
    from poccala_amd import PCL_F32, synth
    from poccala_amd.engine import make_sentence_batch
    units, M, D, U, T, L = 4, 96, 39, 6, 60, 3
    J = units * 3
    mean, var, w, tight = tight_model(31, J, M, D, [0.3, 0.5, 0.8, 0.4, 0.6, 0.7, 0.3, 0.5, 0.8, 0.4, 0.6, 0.7])
    _, _, _, trans = synth.make_model(units, M, D, seed=31)
    rng = np.random.default_rng(31)
    labels = synth.make_labels(U, L, units, seed=32)
    lens = np.full(U, T, dtype=np.int32)
    begin = np.arange(U, dtype=np.int64) * T
    st = np.concatenate([np.repeat([u * 3 + k for u in lab for k in range(3)], T // 9 + 1)[:T] for lab in labels])
    comp = np.array([rng.choice(tight[j]) if t % 2 else rng.integers(0, M) for t, j in enumerate(st)])
    x = (mean[st, comp] + np.sqrt(var[st, comp]) * rng.standard_normal((len(st), D))).astype(np.float32)
    res = {}
    for name, env in (('coarse', {}), ('direct', dict(PCL_COARSE=0, PCL_SPLIT_MAX=0.85))):
        e = _engine(**env)
        try:
            e.load_model(mean, var, w)
            e.load_units(np.stack(trans))
            e.load_frames(x)
            b, _ = make_sentence_batch(e, labels, lens, begin, trans)
            b.score(PCL_F32); b.forward_backward(fix_pi=False)
            e.stats_zero(); b.accumulate(PCL_F32)
            res[name] = (b.get('logp').copy(), e.stats_download(), np.concatenate([m_[1:-1].ravel() for m_ in b.get('B')]))
            b.close()
        finally:
            e.close()
    # what the two routes' emissions differ by (the direct-form route evaluates tight mixtures from f32 roundings of their parameters: up to
    # ~1e-4 nats on frames a few sigma from a mixture of variance 1e-6); a posterior exp(ln w N_m - ln b) inherits it, as in the E-step fuzz
    fin = np.isfinite(res['direct'][2])
    assert np.array_equal(np.isfinite(res['coarse'][2]), fin)
    e_max = float(np.abs(res['coarse'][2][fin] - res['direct'][2][fin]).max())
    note('coarse pass in the E-step', 'max |d ln b| between the two routes', e_max)
    assert e_max < 1e-3
    hold('coarse pass in the E-step', 'ln P(O) against the direct-form route', res['coarse'][0], res['direct'][0], 1e-6)
    for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
        a, d = res['coarse'][1][key], res['direct'][1][key]
        hold('coarse pass in the E-step', key + ' against the direct-form route', a, d, 1e-4 + 4.0 * e_max, 1e-6 * np.abs(d).max())


def test_compacted_main_layout_agrees_with_the_layout_in_mixture_order():
    """A split state's on-pipe mixtures are compacted to the front of its matrix-pipe tiles (compact_main_kernel): the scoring kernel walks
    ceil(on-pipe / 32) tiles, the accumulate consumer maps a layout row back to its mixture at the flush.  Against PCL_COMPACT_MAIN=0 (every
    state in mixture order, off-pipe mixtures as zero-weight rows, all tiles walked) on the same E-step: the same ln b up to the order of an
    f32 sum, the same statistics per MIXTURE -- which is what a wrong row -> mixture map would break -- and the states without off-pipe
    mixtures bit for bit."""
    from poccala_amd import PCL_F32, synth
    from poccala_amd.engine import make_sentence_batch
    units, M, D, U, T, L = 4, 160, 39, 6, 60, 3
    J = units * 3
    shares = [0.0, 0.1, 0.45, 0.0, 0.3, 0.2, 0.45, 0.05, 0.0, 0.4, 0.25, 0.35]          # (<= 0.5: the accumulate pass keeps these states on the consumer)
    mean, var, w, tight = tight_model(41, J, M, D, shares)
    _, _, _, trans = synth.make_model(units, M, D, seed=41)
    rng = np.random.default_rng(41)
    labels = synth.make_labels(U, L, units, seed=42)
    lens = np.full(U, T, dtype=np.int32)
    begin = np.arange(U, dtype=np.int64) * T
    st = np.concatenate([np.repeat([u * 3 + k for u in lab for k in range(3)], T // 9 + 1)[:T] for lab in labels])
    comp = np.array([rng.choice(tight[j]) if (t % 3 == 0 and len(tight[j])) else rng.integers(0, M) for t, j in enumerate(st)])
    x = (mean[st, comp] + np.sqrt(var[st, comp]) * rng.standard_normal((len(st), D))).astype(np.float32)
    res = {}
    for name, env in (('compact', {}), ('mixture order', dict(PCL_COMPACT_MAIN=0))):
        e = _engine(**env)
        try:
            e.enable_timing(True)
            e.load_model(mean, var, w)
            e.load_units(np.stack(trans))
            e.load_frames(x)
            b, _ = make_sentence_batch(e, labels, lens, begin, trans)
            b.score(PCL_F32); b.forward_backward(fix_pi=False)
            e.stats_zero(); b.accumulate(PCL_F32)
            assert e.kernel_time('acc_consume')[1] >= 1                       # the matrix-pipe consumer ran (these states are split, not whole)
            res[name] = (b.get('B'), b.get('logp').copy(), e.stats_download())
            b.close()
        finally:
            e.close()
    Bc, Bm = res['compact'][0], res['mixture order'][0]
    for u, lab in enumerate(labels):
        for pos, unit in enumerate(lab):
            for k in range(3):
                j, row = unit * 3 + k, 1 + pos * 3 + k
                if shares[j] == 0.0:
                    assert np.array_equal(Bc[u][row], Bm[u][row])         # not a split state: nothing about it changed
                else:
                    hold('compacted main layout', 'ln b against the layout in mixture order', Bc[u][row], Bm[u][row], 0.0, F32_LOGLIK_ATOL)      # (each is within it of the oracle: another sum order, another f16-rounded reference)
    hold('compacted main layout', 'ln P(O)', res['compact'][1], res['mixture order'][1], 1e-7)
    for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
        a, d = res['compact'][2][key], res['mixture order'][2][key]
        assert np.array_equal(a == 0, d == 0), key                          # the same mixtures seen
        hold('compacted main layout', key + ' per mixture', a, d, 1e-4, 1e-6 * np.abs(d).max())

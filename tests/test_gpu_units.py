"""GPU tests of the round-2 C-ABI additions: the unit inventory and label-built batches (A7 in the library), the
per-unit transition accumulators and their M-step (A12 / A15 on the device), the guarded GMM M-step, batch
re-validation, opt-in timers, and the E-step exchange (reduce-scatter -> owner M-step -> all-gather) rehearsed with
two GPU processes on one device."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from oracle import poccala_oracle as po

pytestmark = pytest.mark.gpu
S = 5
E = S - 2


@pytest.fixture(scope='module')
def eng():
    from poccala_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def fin_close(got, ref, rtol, atol=0.0):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=rtol, atol=atol)


def problem(seed, units=4, M=8, D=13, U=7, T=50, L=3, ragged=True, dense_trans=False):
    from poccala_amd import synth
    mean, var, w, trans = synth.make_model(units, M, D, seed=seed)
    if dense_trans:   # unit matrices with every allowed entry of rows 1..S-2 non-zero (skips, back loops)
        rng = np.random.default_rng(seed + 9)
        trans = []
        for _ in range(units):
            a = np.zeros((S, S))
            a[0, 1:3] = [0.8, 0.2]
            a[1:-1, :] = rng.dirichlet(np.ones(S), size=E)
            a[1:-1, 0] = 0.0                      # nothing returns to the entry column of the own unit ...
            a[1:-1] /= a[1:-1].sum(axis=1, keepdims=True)
            trans.append(a)
    frames, lens, begin = synth.make_frames(U, T, D, seed=seed + 1, ragged=ragged)
    labels = synth.make_labels(U, L, units, seed=seed + 2)
    return mean, var, w, trans, frames, lens, begin, labels


def oracle_model(mean, var, w, trans):
    return {u: dict(trans=trans[u], gmms=[(mean[u * E + k], var[u * E + k], w[u * E + k]) for k in range(E)])
            for u in range(len(trans))}


# ------------------------------------------------------------------ A7 in the library
@pytest.mark.parametrize('dense', [False, True])
def test_label_batch_equals_host_built_batch(eng, dense):
    """pcl_batch_create_labels builds what AcousticModel.embedded builds (AcousticModel.py:957-1014): the same
    emissions, forward-backward results and Viterbi paths, bit for bit, as a batch fed host-built (N,N) matrices."""
    from poccala_amd import PCL_F64
    from poccala_amd.engine import make_sentence_batch
    mean, var, w, trans, frames, lens, begin, labels = problem(41, dense_trans=dense)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    ref, n = make_sentence_batch(eng, labels, lens, begin, trans)
    lab = eng.label_batch(labels, lens, begin)
    assert np.array_equal(lab.N, n)
    for b in (ref, lab):
        b.score(PCL_F64)
        b.forward_backward()
        b.viterbi()
    for what in ('B', 'alpha', 'beta', 'lgamma', 'ksai', 'gamma', 'pi', 'path'):
        for x, y in zip(ref.get(what), lab.get(what)):
            assert np.array_equal(x, y), what
    for what in ('logp', 'point', 'npass'):
        assert np.array_equal(ref.get(what), lab.get(what)), what
    # and against the oracle's own construction of the sentence HMM
    model = oracle_model(mean, var, w, trans)
    x = frames[begin[2]:begin[2] + lens[2]].astype(np.float64)
    _, a, bref, pi = po.score_label(x, list(labels[2]), model)
    bw = po.baum_welch(a, pi, [bref])
    np.testing.assert_allclose(lab.get('logp')[2], bw['logp'][0], rtol=1e-10)
    ref.close()
    lab.close()


# ------------------------------------------------------------------ A12 on the device
@pytest.mark.parametrize('dense', [False, True])
def test_hmm_accumulators_match_oracle(eng, dense):
    """pcl_batch_accumulate_hmm = LHMM.update_acc + add_acc over every utterance x label position (LHMM.py:473-500,
    149-161), merged per unit, against the oracle's per-position accumulators log-added per unit; two batches
    accumulate into the same context-resident accumulators."""
    from poccala_amd import PCL_F64
    mean, var, w, trans, frames, lens, begin, labels = problem(77, units=5, U=11, L=4, dense_trans=dense)
    model = oracle_model(mean, var, w, trans)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    eng.stats_zero()
    halves = [np.arange(0, 6), np.arange(6, 11)]
    for idx in halves:
        b = eng.label_batch([labels[u] for u in idx], lens[idx], begin[idx])
        b.score(PCL_F64)
        b.forward_backward()
        b.accumulate_hmm()
        b.close()
    ks, ga = eng.hmm_acc_download()
    rk = np.full((5, E, S), -np.inf)
    rg = np.full((5, E), -np.inf)
    for u, lab in enumerate(labels):
        x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
        _, accs, _ = po.estep_utterance(x, list(lab), model)
        for pos, unit in enumerate(lab):
            rk[unit] = np.logaddexp(rk[unit], accs[pos].ksai_acc)
            rg[unit] = np.logaddexp(rg[unit], accs[pos].gamma_acc)
    fin_close(ks, rk, rtol=1e-10)
    fin_close(ga, rg, rtol=1e-10)
    # transition M-step (LHMM.py:519-520) for the units that occurred; the others keep their matrix
    eng.mstep_transitions()
    got = eng.units_download()
    for unit in range(5):
        if np.isfinite(rg[unit]).all():
            np.testing.assert_allclose(got[unit], po.hmm_update_param(trans[unit], rk[unit], rg[unit]), rtol=1e-9, atol=1e-300)
        else:
            assert np.array_equal(got[unit], trans[unit])
    eng.stats_zero()
    ks, ga = eng.hmm_acc_download()
    assert np.isneginf(ks).all() and np.isneginf(ga).all()


@pytest.mark.parametrize('case', ['G6_small_fix0', 'G6_small_fix1', 'G8_floor', 'G6_n62_fix0'])
def test_hmm_accumulators_golden(eng, golden, case):
    """The reference's own per-position ksai_acc / gamma_acc (golden G6 / G8, written by the reference's
    update_acc), merged per unit, through the C-ABI at 1e-10; and the reference's new transition matrices."""
    from poccala_amd import PCL_F64
    g = golden(case)
    names = [str(u) for u in g['unit_names']]
    label = [names.index(str(u)) for u in g['label']]
    x = g['x']
    mean = np.stack([g['mean_%d_%d' % (u, k)] for u in range(len(names)) for k in range(E)])
    var = np.stack([g['var_%d_%d' % (u, k)] for u in range(len(names)) for k in range(E)])
    w = np.stack([g['w_%d_%d' % (u, k)] for u in range(len(names)) for k in range(E)])
    trans = np.stack([g['trans_%d' % u] for u in range(len(names))])
    eng.load_model(mean, var, w)
    eng.load_frames(x)
    eng.load_units(trans)
    b = eng.label_batch([label], [len(x)], [0])
    b.score(PCL_F64)
    fix = int(g['fix_code'])
    b.forward_backward(fix_pi=bool(fix & 1))
    fin_close(b.get('B')[0], g['emb_B'], rtol=1e-12)
    np.testing.assert_allclose(b.get('logp')[0], float(g['bw_logp']), rtol=1e-10)
    eng.stats_zero()
    b.accumulate_hmm()
    ks, ga = eng.hmm_acc_download()
    b.close()
    for unit in set(label):
        pos = [p for p, u in enumerate(label) if u == unit]
        rk = np.full((E, S), -np.inf)
        rg = np.full(E, -np.inf)
        for p in pos:
            rk = np.logaddexp(rk, g['ksai_acc_%d' % p])
            rg = np.logaddexp(rg, g['gamma_acc_%d' % p])
        fin_close(ks[unit], rk, rtol=1e-10)
        fin_close(ga[unit], rg, rtol=1e-10)
    if not fix & 4 and len(set(label)) == len(label):       # every unit once: the reference's update_param output applies as is
        eng.mstep_transitions()
        got = eng.units_download()
        for p, unit in enumerate(label):
            np.testing.assert_allclose(got[unit], g['new_trans_%d' % p], rtol=1e-9, atol=1e-300)


def test_em_loop_stays_on_device(eng):
    """E-step -> em_exchange (one rank: GMM + transition M-step) -> refresh_transitions -> second E-step, against the
    oracle doing the same with merged accumulators and update_param."""
    from poccala_amd import PCL_F64
    mean, var, w, trans, frames, lens, begin, labels = problem(123, units=3, M=4, U=14, T=70, L=3)
    model = oracle_model(mean, var, w, trans)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F64)
    b.forward_backward()
    eng.stats_zero()
    b.accumulate(PCL_F64)
    b.accumulate_hmm()
    eng.em_exchange(c_covariance=1e-3, update_transitions=True)
    nm, nv, nw = eng.model_download()
    nt = eng.units_download()
    J, M, D = mean.shape
    merged = [dict(acc=np.full(M, -np.inf), alpha_acc=-np.inf, mean_acc=np.full((M, D), -np.inf), cov_acc=np.full((M, D), -np.inf)) for _ in range(J)]
    rk = np.full((3, E, S), -np.inf)
    rg = np.full((3, E), -np.inf)
    for u, lab in enumerate(labels):
        x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
        _, accs, _ = po.estep_utterance(x, list(lab), model)
        for pos, unit in enumerate(lab):
            rk[unit] = np.logaddexp(rk[unit], accs[pos].ksai_acc)
            rg[unit] = np.logaddexp(rg[unit], accs[pos].gamma_acc)
            for k in range(E):
                mj, a = merged[unit * E + k], accs[pos].gmm[k]
                for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
                    mj[key] = np.logaddexp(mj[key], a[key])
    new_model = {}
    for unit in range(3):
        gm = []
        for k in range(E):
            rw, rm, rv = po.gmm_update_param(merged[unit * E + k], c_covariance=1e-3)
            np.testing.assert_allclose(nw[unit * E + k], rw, rtol=1e-8)
            np.testing.assert_allclose(nm[unit * E + k], rm, rtol=1e-8, atol=1e-8)
            np.testing.assert_allclose(nv[unit * E + k], rv, rtol=1e-7)
            gm.append((rm, rv, rw))
        rt = po.hmm_update_param(trans[unit], rk[unit], rg[unit])
        np.testing.assert_allclose(nt[unit], rt, rtol=1e-9, atol=1e-300)
        new_model[unit] = dict(trans=rt, gmms=gm)
    # second E-step with the re-estimated model, transitions refreshed in the live batch
    b.refresh_transitions()
    b.score(PCL_F64)
    b.forward_backward()
    lp = b.get('logp')
    for u in (0, 5, 13):
        x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
        _, a, bref, pi = po.score_label(x, list(labels[u]), new_model)
        np.testing.assert_allclose(lp[u], po.baum_welch(a, pi, [bref])['logp'][0], rtol=1e-7)
    b.close()


# ------------------------------------------------------------------ ADVICE r1
def test_mstep_unseen_state_and_zero_occupancy_mixture(eng):
    """A state no utterance contains keeps its parameters; a mixture of a seen state whose occupancy is exactly zero in
    the f32 statistics gets weight 0 and keeps its mean / variance; nothing becomes NaN, the layouts are re-derived
    and the next E-step is finite.  (The reference would produce NaN for the unseen state and a vanishing weight for
    the far mixture, Clustering.py:685-692.)"""
    from poccala_amd import PCL_F32
    mean, var, w, trans, frames, lens, begin, labels = problem(9, units=3, M=6, D=13, U=6, T=60, L=2)
    labels = [np.array([0, 1]) for _ in labels]                 # unit 2 (states 6..8) is never seen
    mean = mean.copy()
    var = var.copy()
    mean[0, 5] += 400.0                                         # a mixture of a seen state 400 sigma away: gamma flushes to 0
    var[0, 5] = 0.5
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F32)
    b.forward_backward()
    eng.stats_zero()
    b.accumulate(PCL_F32)
    b.accumulate_hmm()
    st = eng.stats_download()
    assert st['acc'][0, 5] == 0.0 and (st['alpha_acc'][6:] == 0).all()
    eng.em_exchange(c_covariance=1e-3, update_transitions=True)
    nm, nv, nw = eng.model_download()
    nt = eng.units_download()
    assert np.isfinite(nm).all() and np.isfinite(nv).all() and np.isfinite(nw).all() and np.isfinite(nt).all()
    assert np.array_equal(nm[6:], mean[6:]) and np.array_equal(nv[6:], var[6:]) and np.array_equal(nw[6:], w[6:])
    assert np.array_equal(nt[2], trans[2])
    assert nw[0, 5] == 0.0 and np.array_equal(nm[0, 5], mean[0, 5]) and np.array_equal(nv[0, 5], var[0, 5])
    np.testing.assert_allclose(nw[:6].sum(axis=1), 1.0, rtol=1e-5)
    # the seen mixtures follow the oracle (f32 tolerance)
    model = oracle_model(mean, var, w, trans)
    J, M, D = mean.shape
    mj = dict(acc=np.full(M, -np.inf), alpha_acc=-np.inf, mean_acc=np.full((M, D), -np.inf), cov_acc=np.full((M, D), -np.inf))
    for u, lab in enumerate(labels):
        x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
        _, accs, _ = po.estep_utterance(x, list(lab), model)
        a = accs[0].gmm[1]                                      # unit 0, state 1
        for key in mj:
            mj[key] = np.logaddexp(mj[key], a[key])
    rw, rm, rv = po.gmm_update_param(mj, c_covariance=1e-3)
    from _parity import hold
    hold('guarded M-step small f32', 're-estimated weights', nw[1], rw, 1e-4)
    hold('guarded M-step small f32', 're-estimated means', nm[1], rm, 1e-4, 1e-4)
    hold('guarded M-step small f32', 're-estimated variances', nv[1], rv, 1e-4)
    b.refresh_transitions()
    b.score(PCL_F32)
    b.forward_backward()
    assert np.isfinite(b.get('logp')).all()
    b.close()


def test_batch_is_revalidated_after_reuploads(eng):
    """A live batch whose frame rows or state ids no longer exist in the re-uploaded buffers fails with PCL_ERR_STATE
    instead of indexing out of bounds."""
    from poccala_amd import PCL_F32, PoccalaHipError
    mean, var, w, trans, frames, lens, begin, labels = problem(5)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F32)
    b.forward_backward()
    eng.load_frames(frames[:int(lens[0])])                      # shorter frame matrix
    with pytest.raises(PoccalaHipError) as ei:
        b.score(PCL_F32)
    assert ei.value.code == -3
    with pytest.raises(PoccalaHipError):
        b.accumulate(PCL_F32)
    eng.load_frames(frames)
    b.score(PCL_F32)                                            # valid again
    eng.load_model(mean[:6], var[:6], w[:6])                    # fewer states than the batch refers to
    with pytest.raises(PoccalaHipError) as ei:
        b.score(PCL_F32)
    assert ei.value.code == -3
    b.close()


def test_timers_are_opt_in(eng):
    from poccala_amd import PCL_F32
    mean, var, w, trans, frames, lens, begin, labels = problem(6)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    eng.enable_timing(False)
    b.score(PCL_F32)
    assert eng.kernel_time('score') == (0.0, 0)
    eng.enable_timing(True)
    b.score(PCL_F32)
    b.score(PCL_F32)
    ms, n = eng.kernel_time('score')
    assert n == 2 and ms > 0
    eng.enable_timing(False)
    b.close()


def test_save_batch_acc_twice_merges_to_the_sum(tmp_path):
    """two batches saved one after the other must merge (init_acc) to batch 1 + batch 2, for the HMM
    accumulators as for the GMM ones."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd.Exceptions import NullLog
    from poccala_amd.StatisticalModel.Clustering import Clustering
    from poccala_amd.StatisticalModel.LHMM import LHMM
    from poccala_amd import PCL_F64
    mean, var, w, trans, frames, lens, begin, labels = problem(15, units=2, M=3, D=5, U=6, T=40, L=2)
    names = ['a', 'b']

    def units():
        out = {}
        for ui, u in enumerate(names):
            gm = [Clustering.GMM(NullLog(), dimension=5, mix_level=3, alpha=w[ui * E + k].copy(), mean=mean[ui * E + k].copy(),
                                 covariance=np.array([np.diag(v) for v in var[ui * E + k]]), gmm_id=k) for k in range(E)]
            prof = [AcousticModel.VirtualState(1.)] + gm + [AcousticModel.VirtualState(0.)]
            out[u] = LHMM({i: u for i in range(S)}, S, NullLog(), transmat=trans[ui].copy(), profunc=prof)
        return out
    am = AcousticModel(NullLog(), 'T', state_num=S, dct_num=5, delta_1=False, delta_2=False, parameters_path=str(tmp_path))
    hm = units()
    data = [frames[begin[u]:begin[u] + lens[u]].astype(np.float64) for u in range(6)]
    lab = [[names[i] for i in l] for l in labels]
    parts = []
    for idx in (range(0, 3), range(3, 6)):
        st, ha, _ = am.estep_batch([lab[u] for u in idx], [data[u] for u in idx], hm, precision=PCL_F64)
        parts.append((st, ha))
        am.save_batch_acc(st, ha, hm)
    st_all, ha_all, _ = am.estep_batch(lab, data, hm, precision=PCL_F64)
    fresh = units()
    for u in names:
        fresh[u].init_acc(am.unit_path(u))
        if u in ha_all:
            fin_close(fresh[u].ksai_acc, ha_all[u][0], rtol=1e-10)
            fin_close(fresh[u].gamma_acc, ha_all[u][1], rtol=1e-10)
        for k in range(E):
            g = fresh[u].profunction[1 + k]
            g.init_acc(am.unit_path(u))
            j = names.index(u) * E + k
            with np.errstate(divide='ignore'):
                fin_close(g.acc, np.log(st_all['acc'][j]), rtol=1e-9, atol=1e-9)
                fin_close(np.float64(g.alpha_acc), np.log(st_all['alpha_acc'][j]), rtol=1e-9, atol=1e-9)


def test_repeated_units_are_scored_once_and_accumulated_per_row(eng):
    """A label that names a unit several times (AcousticModel.py:897-902 scores it once per label position): the library scores
    the first row of a state in an utterance and copies its emission row to the others -- every row must still equal the
    oracle's, the copies bit for bit the scored row, and the statistics (each row has its own posteriors, Clustering.py:653-680)
    the oracle's E-step."""
    from poccala_amd import PCL_F32, PCL_F64
    mean, var, w, trans, frames, lens, begin, _ = problem(131, units=3, M=32, U=5, T=60)
    labels = [[0, 1, 0, 0], [2, 2, 2], [1, 0, 1, 2, 1], [0], [2, 1, 2, 1]]
    model = oracle_model(mean, var, w, trans)
    for prec, tol in ((PCL_F32, 5e-5), (PCL_F64, 1e-10)):
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        eng.load_frames(frames)
        b = eng.label_batch(labels, lens, begin)
        b.score(prec)
        B = b.get('B')
        b.forward_backward()
        eng.stats_zero()
        b.accumulate(prec)
        st = eng.stats_download()
        lp = b.get('logp')
        b.close()
        occ = 0.0
        for u, lab in enumerate(labels):
            x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
            _, a, bref, pi = po.score_label(x, list(lab), model)
            fin = np.isfinite(bref)
            assert np.allclose(B[u][fin], bref[fin], rtol=0, atol=tol)
            for p1 in range(len(lab)):                               # rows of a repeated unit: the same bits
                for p2 in range(p1 + 1, len(lab)):
                    if lab[p1] == lab[p2]:
                        assert np.array_equal(B[u][1 + p1 * E:1 + (p1 + 1) * E], B[u][1 + p2 * E:1 + (p2 + 1) * E])
            bw = po.baum_welch(a, pi, [bref])
            assert abs(lp[u] - bw['logp'][0]) <= 1e-6 * abs(bw['logp'][0])
            lg = (bw['alpha'][0] + bw['beta'][0])[1:-1]              # the emitting rows' share of the frames, copies included
            occ += float(np.exp(lg - bw['logp'][0]).sum())
        np.testing.assert_allclose(st['alpha_acc'].sum(), occ, rtol=1e-5)
        np.testing.assert_allclose(st['acc'].sum(axis=1), st['alpha_acc'], rtol=1e-4 if prec == PCL_F32 else 1e-10)


# ------------------------------------------------------------------ the E-step exchange, two GPU processes on one device
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange_worker(rank, world, port, payload, uneven, q, chunk_kb=None):
    try:
        if chunk_kb:
            os.environ['PCL_HOST_CHUNK_KB'] = str(chunk_kb)      # the transport's chunk edges fall inside states
        from poccala_amd import Engine, PCL_F32, PCL_F64, synth
        from poccala_amd.distributed import Control, shard_range
        units = 5 if uneven else 4                      # J = 15 states over 2 ranks: 7 + 8 (the rooted-reduce path)
        mean, var, w, trans = synth.make_model(units, 6, 13, seed=71)
        frames, lens, begin = synth.make_frames(10, 60, 13, seed=72, ragged=True)
        labels = synth.make_labels(10, 3, units, seed=73)
        eng = Engine(0)
        ctl = Control(rank, world, addr='127.0.0.1', port=port, token=b't')
        if world > 1:
            eng.comm_init_host(rank, world, ctl.allgather_bytes)
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        lo, hi = shard_range(10, rank, world)
        f0, f1 = int(begin[lo]), int(begin[hi - 1] + lens[hi - 1])
        eng.load_frames(frames[f0:f1])
        b = eng.label_batch(labels[lo:hi], lens[lo:hi], begin[lo:hi] - f0)
        b.score(PCL_F64)
        b.forward_backward()
        eng.stats_zero()
        b.accumulate(PCL_F64)
        b.accumulate_hmm()
        eng.em_exchange(1e-3, PCL_F32 if payload == 'f32' else PCL_F64, True)
        info = eng.comm_info()
        m, v, ww = eng.model_download()
        t = eng.units_download()
        b.refresh_transitions()
        b.score(PCL_F64)
        b.forward_backward()
        lp = b.get('logp')
        b.close()
        ctl.barrier()
        ctl.close()
        eng.close()
        q.put((rank, info, m, v, ww, t, lp))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, 'error', traceback.format_exc()))


@pytest.mark.parametrize('payload,uneven,chunk_kb', [('f64', False, None), ('f64', True, None), ('f32', False, None), ('f64', True, 1), ('f32', False, 1)])
def test_em_exchange_world2_on_one_device_equals_single_rank(payload, uneven, chunk_kb):
    """two GPU processes (both on device 0, host rehearsal transport in place of RCCL, which
    refuses duplicate devices) shard the utterances, reduce-scatter the statistics by state range, re-estimate the
    states they own and all-gather the model: every rank must end with the model a single rank computes from all
    utterances (f64 payload: 1e-12; f32 payload: the f32 rounding of it), and with the same next-iteration
    log-likelihoods for its utterances.  chunk_kb = 1: the transport moves the arrays in 1-KiB pieces (in production 64 MiB:
    the C4-shape statistics, 3.9 GB, do not fit one callback), so chunk edges cut through states and owner ranges."""
    ctx = mp.get_context('spawn')
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, payload, uneven, q, chunk_kb)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for g in got:
            assert g[1] != 'error', g[2]
        res[world] = sorted(got, key=lambda g: g[0])
    single = res[1][0]
    assert single[1]['transport'] == 'none'
    tol = dict(rtol=1e-12, atol=1e-12) if payload == "f64" else dict(rtol=1e-6, atol=1e-6)   # two partial sums added vs one running sum
    for g in res[2]:
        assert g[1]['transport'] == 'host-rehearsal' and g[1]['nranks'] == 2
        for k in (2, 3, 4):
            np.testing.assert_allclose(g[k], single[k], **tol)
        np.testing.assert_allclose(g[5], single[5], rtol=1e-12, atol=1e-300)
    for k in (2, 3, 4, 5):                                       # one model on every GPU
        assert np.array_equal(res[2][0][k], res[2][1][k])
    lp2 = np.concatenate([g[6] for g in res[2]])
    np.testing.assert_allclose(lp2, single[6], rtol=1e-10 if payload == 'f64' else 1e-6)


def _pipe_worker(rank, world, port, payload, pipelined, q, mode=1):
    try:
        os.environ['PCL_ACC_IMAGE_MB'] = '1'               # many state groups: chunks leave while later groups accumulate
        os.environ['PCL_PIPE_MODE'] = str(mode)            # 1: the reduce-scatter leaves early; 0: the chunk's whole chain
        from poccala_amd import Engine, PCL_F32, PCL_F64, synth
        from poccala_amd.distributed import Control, shard_range
        units = 4                                            # J = 12 states, 5 chunks of 2-3 states, slices of 1-2 states per rank
        mean, var, w, trans = synth.make_model(units, 64, 13, seed=91)
        frames, lens, begin = synth.make_frames(40, 100, 13, seed=92, ragged=True)
        labels = synth.make_labels(40, 3, units, seed=93)
        eng = Engine(0)
        eng.enable_timing(True)
        ctl = Control(rank, world, addr='127.0.0.1', port=port, token=b't')
        if world > 1:
            eng.comm_init_host(rank, world, ctl.allgather_bytes)
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        lo, hi = shard_range(40, rank, world)
        f0, f1 = int(begin[lo]), int(begin[hi - 1] + lens[hi - 1])
        eng.load_frames(frames[f0:f1])
        b = eng.label_batch(labels[lo:hi], lens[lo:hi], begin[lo:hi] - f0)
        pay = PCL_F32 if payload == 'f32' else PCL_F64
        out = []
        for it in range(2):                                  # two EM iterations: the second scores with the layouts the pipe derived
            b.refresh_transitions()
            b.score(PCL_F32)
            b.forward_backward()
            eng.stats_zero()
            b.accumulate_hmm()
            if pipelined:
                b.accumulate_exchange(PCL_F32, 1e-3, pay, True, n_chunks=5)
            else:
                b.accumulate(PCL_F32)
                eng.em_exchange(1e-3, pay, True)
            out.append(eng.model_download() + (eng.units_download(), b.get('logp').copy()))
        groups = eng.kernel_time('acc_consume')[1]          # state groups of the two accumulate passes
        if pipelined:
            chunks, early = eng.pipe_info()
            assert chunks == 5 and early >= 1, (chunks, early)   # the pass itself released chunks (no silent fall-back to 'no overlap')
        b.close()
        ctl.barrier()
        ctl.close()
        eng.close()
        q.put((rank, out, groups))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, 'error', traceback.format_exc()))


@pytest.mark.parametrize('payload,world,mode', [('f64', 1, 1), ('f64', 2, 1), ('f32', 2, 1), ('f64', 1, 0), ('f64', 2, 0), ('f32', 2, 0)])
def test_pipelined_exchange_equals_unpipelined(payload, world, mode):
    """pcl_batch_accumulate_exchange releases state chunks to reduce-scatter -> M-step -> all-gather ->
    derive while the accumulate pass works on later state groups (rank r owns the r-th slice of every chunk; mode 1, the
    default: only the reduce-scatter leaves early, the rest follows at the end; mode 0: the chunk's whole chain).  Same sums,
    same M-step arithmetic (LHMM.py:256-290, Clustering.py:314-367,682-693): the model, the transitions and the next
    iteration's log-likelihoods equal pcl_batch_accumulate + pcl_em_exchange bit for bit, on one rank and on two (host
    rehearsal transport: RCCL refuses two ranks on one device), through two EM iterations."""
    ctx = mp.get_context('spawn')
    res = {}
    for pipelined in (False, True):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, payload, pipelined, q, mode)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for g in got:
            assert g[1] != 'error', g[2]
        res[pipelined] = sorted(got, key=lambda g: g[0])
    for r in range(world):
        assert res[True][r][2] >= 6                          # >= 3 state groups per pass: chunks left while later groups accumulated
        for it in range(2):
            for a, b_ in zip(res[False][r][1][it], res[True][r][1][it]):
                assert np.array_equal(a, b_), (r, it)
    if world == 2:                                           # one model on both ranks
        for k in range(4):
            assert np.array_equal(res[True][0][1][1][k], res[True][1][1][1][k])
    # the model did move (the exchange is not a no-op)
    assert not np.array_equal(res[True][0][1][0][0], res[True][0][1][1][0])


def test_rccl_world1_em_exchange_and_allreduce(eng):
    """RCCL itself (ncclCommInitRank, reduce-scatter, all-gather, all-reduce) at the one world size a one-GPU box
    allows: the exchange must leave what pcl_mstep leaves."""
    from poccala_amd import PCL_F64
    mean, var, w, trans, frames, lens, begin, labels = problem(88)
    res = []
    for use_rccl in (False, True):
        eng.load_model(mean, var, w)
        eng.load_frames(frames)
        eng.load_units(np.stack(trans))
        if use_rccl:
            eng.comm_init(0, 1, eng.comm_unique_id())
            info = eng.comm_info()
            assert info['transport'] == 'rccl' and info['rccl_nranks'] == 1
        b = eng.label_batch(labels, lens, begin)
        b.score(PCL_F64)
        b.forward_backward()
        eng.stats_zero()
        b.accumulate(PCL_F64)
        b.accumulate_hmm()
        if use_rccl:
            eng.stats_allreduce()
        eng.em_exchange(1e-3, PCL_F64, True)
        res.append(eng.model_download() + (eng.units_download(),))
        b.close()
        if use_rccl:
            eng._lib.pcl_comm_destroy(eng._ctx)
    for a, b_ in zip(*res):
        assert np.array_equal(a, b_)


def test_accumulate_outlier_frames_take_the_direct_form_fixup(eng):
    """Frames whose scaled features leave the f16 range (here: thousands of sigma from a tight mixture) are taken out of the
    producer / consumer accumulate path and added by the direct-form kernel: the statistics must match the oracle's
    GMM.update_acc (Clustering.py:653-680) for the whole block, outliers included, and equal the all-VALU path."""
    import os
    from poccala_amd import PCL_F32
    rng = np.random.default_rng(404)
    M, D, T = 40, 39, 100
    mean = rng.standard_normal((2, M, D))
    var = rng.uniform(0.5, 2.0, (2, M, D))
    var[0, :4] = 1e-2                                            # tight mixtures: the feature scales of state 0 are large
    w = rng.dirichlet(np.ones(M), size=2)
    x = rng.standard_normal((T, D))
    # one feature 40 units out: 400 sigma of the tight mixtures, whose power-of-two scale (64) takes x'^2 = 1600 past the
    # f16 range (6e4) -- the frame leaves the matrix-pipe path -- while the exponent itself (~1e3) is still something f32
    # resolves to ~1e-4: further out NO f32 evaluation of ln gamma_t(j,m) = ln N - ln b keeps 1e-4 (both terms ~x^2)
    x[7, 0] += 40.0
    x[55, 3] += 45.0
    x = x.astype(np.float32)
    eng.load_model(mean, var, w)
    eng.load_frames(x)
    b = eng.batch([4], [T], [0])
    b.set_states([np.array([-1, 0, 1, -2], dtype=np.int32)])
    b.score(PCL_F32)
    lb = b.get('B')[0]
    lg = np.log(rng.dirichlet(np.ones(2), size=T).T)             # posteriors of the two emitting rows
    ninf = np.full(T, -np.inf)
    b.set_posteriors([np.stack([ninf, lg[0], lg[1], ninf])])
    eng.stats_zero()
    b.accumulate(PCL_F32)
    st = eng.stats_download()
    b.close()
    xx = x.astype(np.float64)
    for j in range(2):
        acc = dict(acc=np.full(M, -np.inf), alpha_acc=-np.inf, mean_acc=np.full((M, D), -np.inf), cov_acc=np.full((M, D), -np.inf))
        bj = po.gmm_point(xx, mean[j], var[j], w[j])
        po.gmm_update_acc(acc, lg[j], bj, xx, mean[j], var[j], w[j])
        for key, got in (('acc', st['acc'][j]), ('mean_acc', st['mean_acc'][j]), ('cov_acc', st['cov_acc'][j])):
            ref = np.exp(acc[key])
            np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-6 * ref.max(), err_msg='%s state %d' % (key, j))
        np.testing.assert_allclose(st['alpha_acc'][j], np.exp(acc['alpha_acc']), rtol=1e-6)
    np.testing.assert_allclose(lb[1], po.gmm_point(xx, mean[0], var[0], w[0]), rtol=1e-5, atol=2e-4)


def test_accumulate_first_pass_stores_later_passes_add(eng):
    """The first accumulate after stats_zero stores the statistics (every (state, mixture) belongs to one wave, the buffer is
    zero), later ones read-modify-write: the same batch accumulated twice gives exactly twice the statistics, states the
    batch does not contain stay zero, and a second stats_zero starts over."""
    from poccala_amd import PCL_F32
    mean, var, w, trans, frames, lens, begin, labels = problem(17, units=4, M=40, D=39, U=6, T=70, L=2)
    labels = [np.array([0, 2]) for _ in labels]                 # units 1 and 3 are never seen
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F32)
    b.forward_backward()
    eng.stats_zero()
    b.accumulate(PCL_F32)
    one = eng.stats_download()
    b.accumulate(PCL_F32)
    two = eng.stats_download()
    eng.stats_zero()
    b.accumulate(PCL_F32)
    again = eng.stats_download()
    b.close()
    for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
        assert np.array_equal(two[key], 2.0 * one[key]), key
        assert np.array_equal(again[key], one[key]), key
    assert one['alpha_acc'][0] > 0 and (one['alpha_acc'][3:6] == 0).all() and (one['acc'][9:] == 0).all()


def test_accumulate_prune_is_off_by_default_and_bounded(eng):
    """pcl_accumulate_prune: the default (exact zeros only) is what every parity test runs under; with a threshold of 2^-40 the
    state occupancies (float64 sums) move by less than the left-out mass (pairs x 2^-40), the f32 mixture sums by that plus their
    own regrouping noise, and everything comes back bit for bit when the mode is switched off."""
    from poccala_amd import PCL_F32, PoccalaHipError
    mean, var, w, trans, frames, lens, begin, labels = problem(23, units=5, M=40, D=39, U=8, T=80, L=3)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F32)
    b.forward_backward()
    eng.stats_zero(); b.accumulate(PCL_F32)
    exact = eng.stats_download()
    eng.accumulate_prune(-40.0)
    eng.stats_zero(); b.accumulate(PCL_F32)
    cut = eng.stats_download()
    eng.accumulate_prune(-1e300)
    eng.stats_zero(); b.accumulate(PCL_F32)
    back = eng.stats_download()
    with pytest.raises(PoccalaHipError):
        eng.accumulate_prune(1.0)
    lg = b.get('lgamma')
    b.close()
    n_cut = sum(int(((l[1:-1] < -40 * np.log(2)) & (l[1:-1] >= -150 * np.log(2))).sum()) for l in lg)
    assert n_cut > 0                                            # the threshold does leave pairs out on this problem
    for key in exact:
        assert np.array_equal(back[key], exact[key]), key
    assert np.abs(cut['alpha_acc'] - exact['alpha_acc']).max() <= n_cut * 2.0 ** -40
    # (the f32 sums also regroup: leaving frames out moves the others to different 32-frame tiles -> f32 rounding noise)
    np.testing.assert_allclose(cut['acc'], exact['acc'], rtol=2e-6, atol=n_cut * 2.0 ** -40 + 1e-9)


# ------------------------------------------------------------------ the exchange at the world sizes the target has (4 and 8)
class _ThreadGather(object):
    """all-gather of host bytes between n rank THREADS of this process (the rehearsal transport's callback): a GPU box allows
    at most 6 processes on its card, so 8 ranks are 8 contexts on device 0 driven by 8 threads -- the library's calls release
    the GIL and the callback re-enters Python on the calling thread."""

    def __init__(self, n):
        import threading
        self.n, self.slots, self.bar = n, [None] * n, threading.Barrier(n)

    def fn(self, rank):
        def gather(data):
            self.slots[rank] = data
            self.bar.wait(timeout=120)
            out = list(self.slots)
            self.bar.wait(timeout=120)
            return out
        return gather


def _run_ranks(world, body):
    """run body(rank, gather_fn or None) on `world` threads; returns the results in rank order (exceptions re-raised)."""
    import threading
    tg = _ThreadGather(world)
    out, err = [None] * world, []

    def run(r):
        try:
            out[r] = body(r, tg.fn(r) if world > 1 else None)
        except Exception:          # noqa
            import traceback
            err.append(traceback.format_exc())
            tg.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not err, err[0]
    return out


@pytest.mark.parametrize('world', [4, 8])
@pytest.mark.parametrize('payload', ['f64', 'f32'])
def test_exchange_world_4_and_8_equals_single_rank(world, payload):
    """the N-rank statistics / model equal the 1-rank ones (f64 payload 1e-12, f32 payload rtol 1e-5), at the world
    sizes of the target node.  21 utterances (not divisible by 4 or 8) and J = 15 states (not divisible either: rooted reduces /
    broadcasts per owner); per rank: E-step on its utterances, then (a) pcl_stats_allreduce -> every rank holds the global
    statistics, (b) pcl_em_exchange -> every rank holds the model a single rank re-estimates from all utterances, (c) the
    pipelined pcl_batch_accumulate_exchange = (b) bit for bit.  Replaces LHMM.py:256-290 / Clustering.py:314-367."""
    from poccala_amd import Engine, PCL_F32, PCL_F64, synth
    from poccala_amd.distributed import shard_range
    units, U = 5, 21
    mean, var, w, trans = synth.make_model(units, 8, 13, seed=131)
    frames, lens, begin = synth.make_frames(U, 60, 13, seed=132, ragged=True)
    labels = synth.make_labels(U, 3, units, seed=133)
    pay = PCL_F32 if payload == 'f32' else PCL_F64

    def body_for(n):
        def body(rank, gather):
            eng = Engine(0)
            try:
                if gather is not None:
                    eng.comm_init_host(rank, n, gather)
                eng.load_model(mean, var, w)
                eng.load_units(np.stack(trans))
                lo, hi = shard_range(U, rank, n)
                f0, f1 = int(begin[lo]), int(begin[hi - 1] + lens[hi - 1])
                eng.load_frames(frames[f0:f1])
                b = eng.label_batch(labels[lo:hi], lens[lo:hi], begin[lo:hi] - f0)
                res = {}
                # (a) the plain all-reduce: global statistics on every rank
                b.score(PCL_F64); b.forward_backward(); eng.stats_zero(); b.accumulate(PCL_F64); b.accumulate_hmm()
                if gather is not None:
                    eng.stats_allreduce()
                res['stats'] = eng.stats_download()
                res['hmm'] = eng.hmm_acc_download()
                # (b) reduce-scatter -> owned M-step -> all-gather
                eng.stats_zero(); b.accumulate(PCL_F64); b.accumulate_hmm()
                eng.em_exchange(1e-3, pay, True)
                res['model'] = eng.model_download() + (eng.units_download(),)
                res['split'] = (eng.model_conditioning()[0], eng.model_split_info()[0])     # (of ALL states, derived from the gathered model)
                b.refresh_transitions(); b.score(PCL_F64); b.forward_backward()
                res['lp'] = b.get('logp')
                # (c) the same from the same start, pipelined
                eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); b.refresh_transitions()
                b.score(PCL_F64); b.forward_backward(); eng.stats_zero(); b.accumulate_hmm()
                b.accumulate_exchange(PCL_F64, 1e-3, pay, True, n_chunks=4)
                res['model_pipe'] = eng.model_download() + (eng.units_download(),)
                res['split_pipe'] = (eng.model_conditioning()[0], eng.model_split_info()[0])   # (derived range by range inside the pipe)
                res['info'] = eng.comm_info()
                b.close()
                return res
            finally:
                eng.close()
        return body
    single = _run_ranks(1, body_for(1))[0]
    ranks = _run_ranks(world, body_for(world))
    tol = dict(rtol=1e-12, atol=1e-12) if payload == 'f64' else dict(rtol=1e-5, atol=1e-6)
    for r, g in enumerate(ranks):
        assert g['info']['transport'] == 'host-rehearsal' and g['info']['nranks'] == world and g['info']['rank'] == r
        for k in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):                 # (a) is float64 on the wire whatever the payload
            np.testing.assert_allclose(g['stats'][k], single['stats'][k], rtol=1e-12, atol=1e-12, err_msg='all-reduced %s on rank %d' % (k, r))
        for a, b_ in zip(g['hmm'], single['hmm']):
            fin = np.isfinite(b_)
            assert np.array_equal(np.isfinite(a), fin)
            np.testing.assert_allclose(a[fin], b_[fin], rtol=1e-12)
        for k in range(3):
            np.testing.assert_allclose(g['model'][k], single['model'][k], err_msg='model part %d on rank %d' % (k, r), **tol)
        np.testing.assert_allclose(g['model'][3], single['model'][3], rtol=1e-12, atol=1e-300)
        for k in range(4):                                                   # every rank holds ONE model; the pipelined call gives the same bits
            assert np.array_equal(g['model'][k], ranks[0]['model'][k])
            assert np.array_equal(g['model_pipe'][k], g['model'][k]), (r, k)
        for k in range(2):                                                   # ... and one classification of its mixtures (matrix pipe / direct form)
            assert np.array_equal(g['split'][k], ranks[0]['split'][k]) and np.array_equal(g['split_pipe'][k], g['split'][k]), (r, k)
    lp = np.concatenate([g['lp'] for g in ranks])
    np.testing.assert_allclose(lp, single['lp'], rtol=1e-10 if payload == 'f64' else 1e-5)


def test_batch_limits_are_said_at_create(eng):
    """65536 utterances in one batch: several kernels index the utterance with the grid's second dimension; the library says so when the batch
    is made (rounds 1-5a: a failed launch inside the first forward-backward)."""
    from poccala_amd import PoccalaHipError
    mean, var, w, trans, frames, lens, begin, labels = problem(5)
    eng.load_model(mean, var, w)
    eng.load_frames(frames)
    eng.load_units(np.stack(trans))
    U = 65536
    with pytest.raises(PoccalaHipError, match='65535'):
        eng.batch(np.full(U, 5, dtype=np.int32), np.full(U, 2, dtype=np.int32), np.zeros(U, dtype=np.int64))
    with pytest.raises(PoccalaHipError, match='65535'):
        eng.label_batch(np.zeros((U, 1), dtype=np.int32), np.full(U, 2, dtype=np.int32), np.zeros(U, dtype=np.int64))
    # ... and the largest one that is allowed runs (one-unit sentence HMMs of two frames)
    U = 65535
    b = eng.label_batch(np.zeros((U, 1), dtype=np.int32), np.full(U, 2, dtype=np.int32), np.zeros(U, dtype=np.int64))
    b.score()
    b.forward_backward()
    b.viterbi()
    lp = b.get('logp')
    assert lp.shape == (U,) and np.all(lp == lp[0]) and np.isfinite(lp[0])
    b.close()


@pytest.mark.parametrize('payload', ['f64', 'f32'])
def test_exchange_with_ranks_that_hold_no_utterance(payload):
    """Two utterances on a world of four: ranks 2 and 3 have nothing to score (a corpus shard smaller than the node).  They zero their
    statistics and take part in the exchange; every rank ends with the model a single rank re-estimates from the two utterances."""
    from poccala_amd import Engine, PCL_F32, PCL_F64, synth
    from poccala_amd.distributed import shard_range
    units, U, world = 3, 2, 4
    mean, var, w, trans = synth.make_model(units, 5, 13, seed=431)
    frames, lens, begin = synth.make_frames(U, 50, 13, seed=432, ragged=True)
    labels = synth.make_labels(U, 3, units, seed=433)
    pay = PCL_F32 if payload == 'f32' else PCL_F64

    def body_for(n):
        def body(rank, gather):
            eng = Engine(0)
            try:
                if gather is not None:
                    eng.comm_init_host(rank, n, gather)
                eng.load_model(mean, var, w)
                eng.load_units(np.stack(trans))
                lo, hi = shard_range(U, rank, n)
                eng.stats_zero()
                if hi > lo:
                    f0, f1 = int(begin[lo]), int(begin[hi - 1] + lens[hi - 1])
                    eng.load_frames(frames[f0:f1])
                    b = eng.label_batch(labels[lo:hi], lens[lo:hi], begin[lo:hi] - f0)
                    b.score(PCL_F64); b.forward_backward(); b.accumulate(PCL_F64); b.accumulate_hmm()
                    b.close()
                eng.em_exchange(1e-3, pay, True)
                return eng.model_download() + (eng.units_download(),), (lo, hi)
            finally:
                eng.close()
        return body
    single = _run_ranks(1, body_for(1))[0][0]
    ranks = _run_ranks(world, body_for(world))
    assert sorted(hi - lo for _, (lo, hi) in ranks) == [0, 0, 1, 1]
    tol = dict(rtol=1e-12, atol=1e-12) if payload == 'f64' else dict(rtol=1e-5, atol=1e-6)
    for r, (model, _) in enumerate(ranks):
        for got, want, nm in zip(model, single, ('mean', 'var', 'weight', 'transitions')):
            np.testing.assert_allclose(got, want, err_msg='%s on rank %d' % (nm, r), **tol)
        for got, first in zip(model, ranks[0][0]):
            assert np.array_equal(got, first), 'rank %d holds another model than rank 0' % r


@pytest.mark.parametrize('tag', ['t1_l1', 't2_l1', 't3_l1', 't1_l2', 't2_l4'])
def test_very_short_utterances_as_the_reference_has_them(eng, golden, tag):
    """Golden G15 (the reference's own worker sequence on utterances of one, two and three frames, and on a label of four units over two
    frames): with one frame LHMM.baulm_welch raises and every accumulator keeps its ln 0 -- the library adds nothing for that utterance;
    with two or three frames the reference's accumulators through the C-ABI at 1e-9."""
    from poccala_amd import PCL_F64
    g = golden('G15_edges')
    names = [str(u) for u in g[tag + '_unit_names']]
    label = [names.index(str(u)) for u in g[tag + '_label']]
    x = g[tag + '_x']
    flat = np.zeros((S, S))
    flat[0][1] = 1.
    for j in range(1, S - 1):
        flat[j][j] = flat[j][j + 1] = 0.5
    mean = np.stack([g['%s_mean_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    var = np.stack([g['%s_var_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    w = np.stack([g['%s_w_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    eng.load_model(mean, var, w)
    eng.load_units(np.stack([flat] * len(names)))
    eng.load_frames(x)
    eng.stats_zero()
    b = eng.label_batch([np.array(label)], np.array([x.shape[0]], dtype=np.int32), np.zeros(1, dtype=np.int64))
    b.score(PCL_F64)
    b.forward_backward()
    b.accumulate(PCL_F64)
    b.accumulate_hmm()
    st = eng.stats_download()
    ks, ga = eng.hmm_acc_download()
    if x.shape[0] == 1:
        assert str(g[tag + '_raised']) == 'ValueError'
        assert not st['acc'].any() and not st['alpha_acc'].any() and not st['mean_acc'].any() and not st['cov_acc'].any()
        assert np.isneginf(ks).all() and np.isneginf(ga).all() and np.isneginf(b.get('lgamma')[0]).all()
        b.close()
        return
    np.testing.assert_allclose(b.get('logp')[0], float(g[tag + '_logp']), rtol=1e-10)
    assert int(b.get('npass')[0]) == len(g[tag + '_q_trace'])
    rk, rg = np.full((len(names), E, S), -np.inf), np.full((len(names), E), -np.inf)
    ref = dict(acc=np.zeros_like(st['acc']), alpha_acc=np.zeros_like(st['alpha_acc']), mean_acc=np.zeros_like(st['mean_acc']), cov_acc=np.zeros_like(st['cov_acc']))
    for pos, unit in enumerate(label):
        rk[unit] = np.logaddexp(rk[unit], g['%s_ksai_acc_%d' % (tag, pos)])
        rg[unit] = np.logaddexp(rg[unit], g['%s_gamma_acc_%d' % (tag, pos)])
        for k in range(E):
            for nm in ref:
                ref[nm][unit * E + k] += np.exp(g['%s_%s_%d_%d' % (tag, nm, pos, k)])
    fin_close(ks, rk, rtol=1e-9)
    fin_close(ga, rg, rtol=1e-9)
    for nm in ref:
        np.testing.assert_allclose(st[nm], ref[nm], rtol=1e-9, atol=1e-13 * np.abs(ref[nm]).max(), err_msg=nm)
    b.close()


@pytest.mark.parametrize('kind', ['tight', 'wide', 'skewed'])
def test_ill_conditioned_models_against_the_reference_itself(eng, golden, kind):
    """Golden G16: the reference's own accumulators and re-estimated model on mixtures at the 1e-6 variance floor / variances over four
    decades in one state / weights down to 1e-12, through the C-ABI in the float64 mode (the default mode is held to the oracle on such
    draws by tests/test_gpu_fuzz_estep.py, and the oracle to this fixture by tests/test_oracle_golden.py)."""
    from poccala_amd import PCL_F64
    g = golden('G16_kinds')
    names = [str(u) for u in g[kind + '_unit_names']]
    label = [names.index(str(u)) for u in g[kind + '_label']]
    x = g[kind + '_x']
    flat = np.zeros((S, S))
    flat[0][1] = 1.
    for j in range(1, S - 1):
        flat[j][j] = flat[j][j + 1] = 0.5
    mean = np.stack([g['%s_mean_%d_%d' % (kind, ui, k)] for ui in range(len(names)) for k in range(E)])
    var = np.stack([g['%s_var_%d_%d' % (kind, ui, k)] for ui in range(len(names)) for k in range(E)])
    w = np.stack([g['%s_w_%d_%d' % (kind, ui, k)] for ui in range(len(names)) for k in range(E)])
    eng.load_model(mean, var, w)
    eng.load_units(np.stack([flat] * len(names)))
    eng.load_frames(x)
    eng.stats_zero()
    b = eng.label_batch([np.array(label)], np.array([x.shape[0]], dtype=np.int32), np.zeros(1, dtype=np.int64))
    b.score(PCL_F64)
    fin = np.isfinite(g[kind + '_emb_B'])
    Bd = b.get('B')[0]
    assert np.array_equal(np.isfinite(Bd), fin)
    np.testing.assert_allclose(Bd[fin], g[kind + '_emb_B'][fin], rtol=1e-11, atol=1e-10)
    b.forward_backward()
    b.accumulate(PCL_F64)
    b.accumulate_hmm()
    np.testing.assert_allclose(b.get('logp')[0], float(g[kind + '_logp']), rtol=1e-11)
    assert int(b.get('npass')[0]) == len(g[kind + '_q_trace'])
    st = eng.stats_download()
    ks, ga = eng.hmm_acc_download()
    for pos, unit in enumerate(label):                       # (two different units: a position is a unit)
        fin_close(ks[unit], g['%s_ksai_acc_%d' % (kind, pos)], rtol=1e-9)
        fin_close(ga[unit], g['%s_gamma_acc_%d' % (kind, pos)], rtol=1e-9)
        for k in range(E):
            j = unit * E + k
            with np.errstate(all='ignore'):
                for nm in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
                    ref = np.exp(g['%s_%s_%d_%d' % (kind, nm, pos, k)])
                    np.testing.assert_allclose(st[nm][j], ref, rtol=1e-8, atol=1e-12 * max(float(np.max(ref)), 1e-300), err_msg='%s state %d' % (nm, j))
    eng.em_exchange(1e-6)
    nm_, nv_, nw_ = eng.model_download()
    for pos, unit in enumerate(label):
        for k in range(E):
            j = unit * E + k
            rw, rm, rv = g['%s_new_w_%d_%d' % (kind, pos, k)], g['%s_new_mean_%d_%d' % (kind, pos, k)], g['%s_new_var_%d_%d' % (kind, pos, k)]
            seen = st['acc'][j] > 0.0                        # (a mixture with occupancy 0: the reference's 0 / 0; the library keeps its parameters)
            ok = seen & np.isfinite(rm).all(axis=1)
            np.testing.assert_allclose(nw_[j][ok], rw[ok], rtol=1e-8, atol=1e-300)
            np.testing.assert_allclose(nm_[j][ok], rm[ok], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(nv_[j][ok], rv[ok], rtol=1e-7, atol=1e-12)
    b.close()


@pytest.mark.parametrize('seed', range(6))
def test_exchange_random_worlds(seed):
    """Random world sizes (2 .. 8, threads on one device over the host transport), state counts that no world divides, mixtures, feature
    dimensions, utterance counts (fewer utterances than ranks included), payloads: after the exchange every rank holds the model a single
    rank re-estimates from all utterances (float64 payload 1e-12, f32 payload 1e-5), every rank the SAME bits, and the pipelined
    accumulate + exchange equals the plain one bit for bit."""
    from poccala_amd import Engine, PCL_F32, PCL_F64, synth
    from poccala_amd.distributed import shard_range
    rng = np.random.default_rng(7100 + seed)
    world = int(rng.choice([2, 3, 5, 6, 8]))
    units = int(rng.integers(2, 8))
    M = int(rng.choice([1, 3, 8, 33]))
    D = int(rng.choice([13, 26, 39]))
    U = int(rng.integers(1, 3 * world))
    mean, var, w, _ = synth.make_model(units, M, D, seed=7200 + seed)
    trans = [synth.random_left_right_transmat(rng) for _ in range(units)]
    frames, lens, begin = synth.make_frames(U, 40, D, seed=7300 + seed, ragged=True)
    labels = synth.make_labels(U, int(rng.integers(1, 4)), units, seed=7400 + seed)
    pay = PCL_F32 if rng.random() < 0.5 else PCL_F64
    c_cov = float(rng.choice([1e-3, 1e-6]))
    n_chunks = int(rng.integers(1, 6))

    def body_for(n):
        def body(rank, gather):
            eng = Engine(0)
            try:
                if gather is not None:
                    eng.comm_init_host(rank, n, gather)
                res = {}
                for mode in ('plain', 'pipe'):
                    eng.load_model(mean, var, w)
                    eng.load_units(np.stack(trans))
                    eng.stats_zero()
                    lo, hi = shard_range(U, rank, n)
                    b = None
                    if hi > lo:
                        f0, f1 = int(begin[lo]), int(begin[hi - 1] + lens[hi - 1])
                        eng.load_frames(frames[f0:f1])
                        b = eng.label_batch(labels[lo:hi], lens[lo:hi], begin[lo:hi] - f0)
                        b.score(PCL_F64); b.forward_backward(); b.accumulate_hmm()
                    # (all ranks must run the same sequence of collectives: a rank without a batch takes the idle form of the pipelined call)
                    if mode == 'plain':
                        if b is not None:
                            b.accumulate(PCL_F64)
                        eng.em_exchange(c_cov, pay, True)
                    elif b is not None:
                        b.accumulate_exchange(PCL_F64, c_cov, pay, True, n_chunks=n_chunks)
                    else:
                        eng.accumulate_exchange_idle(c_cov, pay, True, n_chunks=n_chunks)
                    res[mode] = eng.model_download() + (eng.units_download(),)
                    if b is not None:
                        b.close()
                return res
            finally:
                eng.close()
        return body
    single = _run_ranks(1, body_for(1))[0]
    ranks = _run_ranks(world, body_for(world))
    tol = dict(rtol=1e-12, atol=1e-12) if pay == PCL_F64 else dict(rtol=1e-5, atol=1e-6)
    ctx = 'seed %d world %d units %d M %d D %d U %d payload %s' % (seed, world, units, M, D, U, 'f64' if pay == PCL_F64 else 'f32')
    for r, g in enumerate(ranks):
        for got, want, nm in zip(g['plain'], single['plain'], ('mean', 'var', 'weight', 'transitions')):
            np.testing.assert_allclose(got, want, err_msg='%s: %s on rank %d' % (ctx, nm, r), **tol)
        for got, first in zip(g['plain'], ranks[0]['plain']):
            assert np.array_equal(got, first), '%s: rank %d holds another model than rank 0' % (ctx, r)
        if True:
            for got, plain in zip(g['pipe'], g['plain']):
                assert np.array_equal(got, plain), '%s: pipelined differs from plain on rank %d' % (ctx, r)


def test_zero_occupancy_mixture_as_the_reference_has_it(eng, golden):
    """Golden G15: the reference's GMM.update_acc + update_param on a state one of whose mixtures has weight 0.  Through the C-ABI: the same
    accumulators, the same re-estimated parameters for the mixtures that were responsible for something; for the one that was not, the reference's
    0 / 0 is NaN mean and variance (the fixture holds them), the library's is weight 0 with mean and variance kept."""
    from poccala_amd import PCL_F64
    g = golden('G15_edges')
    mean, var, w, x = g['zero_mean'][None], g['zero_var'][None], g['zero_w'][None], g['zero_x']
    t = x.shape[0]
    eng.load_model(mean, var, w)
    eng.load_frames(x)
    b = eng.batch([3], [t], [0])
    b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
    ninf = np.full(t, -np.inf)
    b.set_emissions([np.stack([np.zeros(t), g['zero_bval'], ninf])])
    b.set_posteriors([np.stack([ninf, g['zero_lval'], ninf])])
    eng.stats_zero()
    b.accumulate(PCL_F64)
    st = eng.stats_download()
    with np.errstate(all='ignore'):
        np.testing.assert_allclose(st['acc'][0], np.exp(g['zero_acc']), rtol=1e-9)
        np.testing.assert_allclose(st['mean_acc'][0], np.exp(g['zero_mean_acc']), rtol=1e-9)
        np.testing.assert_allclose(st['cov_acc'][0], np.exp(g['zero_cov_acc']), rtol=1e-9, atol=1e-300)
    assert st['acc'][0][2] == 0.0
    eng.mstep(1e-3)
    nm, nv, nw = eng.model_download()
    ok = ~np.isnan(g['zero_new_mean']).any(axis=1)
    np.testing.assert_allclose(nw[0], g['zero_new_w'], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(nm[0][ok], g['zero_new_mean'][ok], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(nv[0][ok], g['zero_new_var'][ok], rtol=1e-8)
    assert nw[0][2] == 0.0 and np.array_equal(nm[0][2], mean[0][2]) and np.array_equal(nv[0][2], var[0][2])
    b.close()


def test_impossible_utterance_as_the_reference_has_it(eng, golden):
    """Golden G15 'p0': a frame no state can emit.  The reference does not raise: it leaves the per-unit accumulators at ln 0 and turns the GMM
    accumulators of the label's states into NaN (the fixture holds them).  The library: ln P(O) = -inf and nothing added to either."""
    from poccala_amd import PCL_F64
    g = golden('G15_edges')
    tag = 'p0'
    names = [str(u) for u in g[tag + '_unit_names']]
    label = [names.index(str(u)) for u in g[tag + '_label']]
    x = g[tag + '_x']
    flat = np.zeros((S, S))
    flat[0][1] = 1.
    for j in range(1, S - 1):
        flat[j][j] = flat[j][j + 1] = 0.5
    mean = np.stack([g['%s_mean_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    var = np.stack([g['%s_var_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    w = np.stack([g['%s_w_%d_%d' % (tag, ui, k)] for ui in range(len(names)) for k in range(E)])
    assert np.isnan(g[tag + '_acc_0_0']).all() and np.isneginf(g[tag + '_ksai_acc_0']).all()      # what the reference left
    eng.load_model(mean, var, w)
    eng.load_units(np.stack([flat] * len(names)))
    eng.load_frames(x)
    eng.stats_zero()
    b = eng.label_batch([np.array(label)], np.array([x.shape[0]], dtype=np.int32), np.zeros(1, dtype=np.int64))
    b.score(PCL_F64)
    b.set_emissions([g[tag + '_emb_B']])
    b.forward_backward()
    b.accumulate(PCL_F64)
    b.accumulate_hmm()
    assert np.isneginf(b.get('logp')[0]) and int(b.get('npass')[0]) == 1
    st = eng.stats_download()
    ks, ga = eng.hmm_acc_download()
    assert not st['acc'].any() and not st['alpha_acc'].any() and not st['mean_acc'].any() and not st['cov_acc'].any()
    assert np.isfinite(st['acc']).all() and np.isneginf(ks).all() and np.isneginf(ga).all()
    b.close()

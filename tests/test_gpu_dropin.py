"""The drop-in classes (poccala_amd.StatisticalModel / poccala_amd.AcousticModel) driven exactly the way
the reference's worker drives its own classes (multi_embedded_training_1, AcousticModel.py:884-916), and
compared with what the reference produced for the same inputs (golden G4..G9)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
S = 5
CASES = ['G6_small_fix0', 'G6_small_fix1', 'G6_small_fix2', 'G6_small_fix3', 'G6_small_fix4', 'G6_small_fix6',
         'G8_floor', 'G6_n62_fix0']


class RecLog(object):
    def __init__(self):
        self.msgs = []

    def note(self, content, cls='i', show_console=True):
        self.msgs.append((cls, content))


def fin_close(got, ref, rtol, atol=0.0):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=rtol, atol=atol)


def build_units(g):
    """Unit HMMs from the fixture, one instance per label position (as init_unit + init_parameter do)."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd.StatisticalModel.Clustering import Clustering
    from poccala_amd.StatisticalModel.LHMM import LHMM
    names = [str(u) for u in g['unit_names']]
    label = [str(u) for u in g['label']]
    hmm_list = []
    for u in label:
        ui = names.index(u)
        gmms = []
        for k in range(S - 2):
            var = g['var_%d_%d' % (ui, k)]
            cov = np.array([np.diag(v) for v in var])                       # (M,D,D) as the reference stores it
            gmms.append(Clustering.GMM(RecLog(), dimension=var.shape[1], mix_level=var.shape[0],
                                       alpha=g['w_%d_%d' % (ui, k)].copy(), mean=g['mean_%d_%d' % (ui, k)].copy(),
                                       covariance=cov, gmm_id=k))
        prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
        hmm_list.append(LHMM({i: u for i in range(S)}, S, RecLog(), transmat=g['trans_%d' % ui].copy(), profunc=prof))
    return label, hmm_list


@pytest.mark.parametrize('case', CASES)
def test_worker_flow_matches_reference(golden, case):
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd.StatisticalModel.LHMM import LHMM
    g = golden(case)
    fix = int(g['fix_code'])
    x = g['x']
    t = len(x)
    label, hmm_list = build_units(g)
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S)
    # --- the reference worker, line for line in meaning (AcousticModel.py:897-910)
    for hmm in hmm_list:
        hmm.cal_observation_pro([x], [t])
        hmm.clear_data()
    states, A, B, pi = am.embedded(label, hmm_list, 0, 15)
    fin_close(B, g['emb_B'], rtol=1e-12)
    np.testing.assert_allclose(A, g['emb_A'], rtol=0)
    assert [states[i] for i in range(len(states))] == [str(s) for s in g['emb_states']]
    # alignment (AcousticModel.py:750)
    point, seq = am.viterbi(states, A, B, pi)
    assert np.array_equal(np.asarray(seq).astype(str), g['vit_path_conv'].astype(str))
    np.testing.assert_allclose(point, float(g['vit_point_conv']), rtol=1e-12)
    for u in set(label):
        runs = AcousticModel.discriminate(u, seq)
        assert len(runs) == int(g['disc_%s_n' % u])
        for ri, r in enumerate(runs):
            assert np.array_equal(r, g['disc_%s_%d' % (u, ri)])
    # E-step (AcousticModel.py:906-910)
    elog = RecLog()
    embed = LHMM(states, S, elog, transmat=A, probmat=[B], pi=pi, hmm_list=hmm_list, fix_code=fix)
    embed.add_data([x])
    embed.add_T([t])
    embed.baulm_welch(show_q=False)
    assert embed.n_pass == int(g['bw_n_pass'])
    np.testing.assert_allclose(embed.q_trace[1:], g['bw_q_trace'][1:], atol=2e-6)
    np.testing.assert_allclose(embed.pi, g['bw_pi'], rtol=1e-9, atol=1e-300)
    for pos, h in enumerate(hmm_list):
        fin_close(h.ksai_acc, g['ksai_acc_%d' % pos], rtol=1e-10)
        fin_close(h.gamma_acc, g['gamma_acc_%d' % pos], rtol=1e-10)
        for k in range(S - 2):
            gm = h.profunction[1 + k]
            fin_close(gm.acc, g['acc_%d_%d' % (pos, k)], rtol=1e-8, atol=1e-10)          # log-domain values may sit near 0
            fin_close(np.float64(gm.alpha_acc), g['alpha_acc_%d_%d' % (pos, k)], rtol=1e-10, atol=1e-12)   # (ln of an occupancy near 1: 2.3e-5 in G6_small_fix0)
            fin_close(gm.mean_acc, g['mean_acc_%d_%d' % (pos, k)], rtol=1e-8, atol=1e-10)
            fin_close(np.array(gm._GMM__covariance_acc), g['cov_acc_%d_%d' % (pos, k)], rtol=1e-8, atol=1e-10)
    # M-step (multi_embedded_training_2 -> LHMM.update_param, AcousticModel.py:918-935)
    for pos, h in enumerate(hmm_list):
        h.fix_code = fix
        h.update_param(c_covariance=float(g['c_covariance']))
        np.testing.assert_allclose(h.transmat, g['new_trans_%d' % pos], rtol=1e-9, atol=1e-300)
        for k in range(S - 2):
            gm = h.profunction[1 + k]
            np.testing.assert_allclose(gm.alpha, g['new_w_%d_%d' % (pos, k)], rtol=1e-8)
            np.testing.assert_allclose(gm.mean, g['new_mean_%d_%d' % (pos, k)], rtol=1e-6, atol=1e-8)
            got_var = np.array([np.diagonal(c) for c in gm.covariance])
            np.testing.assert_allclose(got_var, g['new_var_%d_%d' % (pos, k)], rtol=1e-7)


def test_model_tree_round_trip_matches_reference_layout(golden, tmp_path):
    """T3 / G9: the directory tree written by save_parameter / save_acc has the reference's files, dtypes
    and shapes, and loads back."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    g = golden('G6_small_fix0')
    label, hmm_list = build_units(g)
    x = g['x']
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S, mix_level=4, parameters_path=str(tmp_path))
    am.save_parameter('b', hmm_list[0])
    am.save_acc('b', hmm_list[0])
    listing = []
    root = am.unit_path('b')
    for r, dirs, files in os.walk(root):
        dirs.sort()
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(r, f), root)
            if f.endswith('.npy'):
                arr = np.load(os.path.join(r, f))
                parts = rel.rsplit('_', 1)
                if parts[-1][:-4].isdigit():
                    rel = parts[0] + '_<ts>.npy'
                listing.append('%s|%s|%s' % (rel, arr.dtype, 'x'.join(map(str, arr.shape))))
    ref = [s for s in (str(v) for v in golden('G9_layout')['listing']) if '|text|' not in s]
    assert sorted(listing) == sorted(ref)
    fresh = am.init_unit('b')
    am.init_parameter('b', fresh)
    np.testing.assert_array_equal(fresh.transmat, hmm_list[0].transmat)
    np.testing.assert_array_equal(fresh.profunction[2].mean, hmm_list[0].profunction[2].mean)
    fresh.cal_observation_pro([x], [len(x)])
    hmm_list[0].cal_observation_pro([x], [len(x)])
    fin_close(fresh.B_p[0], hmm_list[0].B_p[0], rtol=0)


def test_gmm_point_single_frame_and_dimension_error(golden):
    from poccala_amd.Exceptions import DataDimensionError
    from poccala_amd.StatisticalModel.Clustering import Clustering
    from poccala_amd.StatisticalModel.util import gaussian_function
    g = golden('G2_gmm_point')
    key = '8_39'
    gm = Clustering.GMM(RecLog(), dimension=39, mix_level=8, alpha=g['w_' + key], mean=g['mean_' + key],
                        covariance=np.array([np.diag(v) for v in g['var_' + key]]))
    for t in (0, 5):
        np.testing.assert_allclose(gm.point(g['x_' + key][t], log=True, record=True), g['out_' + key][t], rtol=1e-12)
    with pytest.raises(DataDimensionError):
        gm.point(g['x_' + key][0][:38], log=True)
    g1 = golden('G1_util')
    out = gaussian_function(g1['gauss_y_39'][2], g1['gauss_mean_39'][2], np.diag(g1['gauss_var_39'][2]), 39, log=True)
    np.testing.assert_allclose(out, g1['gauss_out_39'][2], rtol=1e-12)


def test_batched_estep_equals_per_utterance_dropin(golden):
    """AcousticModel.estep_batch (everything resident on the GPU) gives the statistics that the
    per-utterance drop-in flow accumulates."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd import PCL_F64
    g = golden('G6_small_fix0')
    label, hmm_list = build_units(g)
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S, mix_level=4)
    unit_hmms = {u: hmm_list[label.index(u)] for u in set(label)}
    stats, hmm_acc, logp = am.estep_batch([label], [g['x']], unit_hmms, fix_code=0, precision=PCL_F64)
    np.testing.assert_allclose(logp[0], float(g['bw_logp']), rtol=1e-10)
    units = sorted(unit_hmms)
    for unit in units:
        pos_list = [p for p, u in enumerate(label) if u == unit]
        ref_k = np.logaddexp.reduce([g['ksai_acc_%d' % p] for p in pos_list])
        fin_close(hmm_acc[unit][0], ref_k, rtol=1e-9)
        for k in range(S - 2):
            j = units.index(unit) * (S - 2) + k
            ref = sum(np.exp(g['acc_%d_%d' % (p, k)]) for p in pos_list)
            np.testing.assert_allclose(stats['acc'][j], ref, rtol=1e-8)
            refm = sum(np.exp(g['mean_acc_%d_%d' % (p, k)]) for p in pos_list)
            np.testing.assert_allclose(stats['mean_acc'][j], refm, rtol=1e-8)


def test_segment_batch_and_accumulator_files(golden, tmp_path):
    """The steps after the two workers: alignment -> per-unit segments (AcousticModel.py:758-764) and the
    reference-format accumulator files a reference multi_embedded_training_2 could merge."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd import PCL_F64
    g = golden('G6_small_fix0')
    label, hmm_list = build_units(g)
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S, mix_level=4, dct_num=13, delta_1=False, delta_2=False,
                       parameters_path=str(tmp_path))
    unit_hmms = {u: hmm_list[label.index(u)] for u in set(label)}
    segs, dropped = am.segment_batch([label], [g['x']], unit_hmms, precision=PCL_F64)
    assert dropped == []
    for u in set(label):
        assert len(segs[u]) == int(g['disc_%s_n' % u])
        for ri, blk in enumerate(segs[u]):
            np.testing.assert_array_equal(blk, g['x'][g['disc_%s_%d' % (u, ri)]])
    stats, hmm_acc, _ = am.estep_batch([label], [g['x']], unit_hmms, fix_code=0, precision=PCL_F64)
    am.save_batch_acc(stats, hmm_acc, unit_hmms)
    fresh = am.init_unit('b')
    fresh.init_acc(am.unit_path('b'))
    for k in range(S - 2):
        fresh.profunction[1 + k].init_acc(am.unit_path('b'))
    pos = [p for p, u in enumerate(label) if u == 'b']
    fin_close(fresh.ksai_acc, np.logaddexp.reduce([g['ksai_acc_%d' % p] for p in pos]), rtol=1e-9)
    fin_close(fresh.profunction[1].acc, np.logaddexp.reduce([g['acc_%d_0' % p] for p in pos]), rtol=1e-8, atol=1e-10)
    fin_close(fresh.profunction[2].mean_acc, np.logaddexp.reduce([g['mean_acc_%d_1' % p] for p in pos]), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('tag', ['free', 'pifixed'])
def test_lhmm_with_several_utterances(golden, tag):
    """datasize > 1: the host-driven pass loop of the drop-in LHMM (one device pass per iteration, merge as
    LHMM.py:454-466) against the reference's own run (G11)."""
    from poccala_amd.StatisticalModel.LHMM import LHMM
    g = golden('G11_multi_utterance')
    n = 8
    bs = [g['B%d_%s' % (k, tag)].copy() for k in range(3)]
    fix = 1 if tag == 'pifixed' else 0
    unit = LHMM({i: 'u' for i in range(n)}, n, RecLog(), transmat=g['A_' + tag].copy(), probmat=[np.zeros((n, 1))], fix_code=6)
    h = LHMM({i: 'u' for i in range(n)}, n, RecLog(), transmat=g['A_' + tag].copy(), probmat=bs, pi=g['pi0_' + tag].copy(),
             hmm_list=[unit], fix_code=fix | 2)
    h.add_data([np.zeros((b.shape[1], 1)) for b in bs])
    h.add_T([b.shape[1] for b in bs])
    h.baulm_welch()
    assert h.n_pass == int(g['n_pass_' + tag])
    np.testing.assert_allclose(h.q_trace[1:], g['q_trace_' + tag][1:], atol=2e-6)
    assert np.shape(h.pi) == np.shape(g['pi_' + tag])
    np.testing.assert_allclose(h.pi, g['pi_' + tag], rtol=1e-9, atol=1e-300)
    fin_close(unit.ksai_acc, g['ksai_acc_' + tag], rtol=1e-10)
    fin_close(unit.gamma_acc, g['gamma_acc_' + tag], rtol=1e-10)


def test_regroup_kernel_matches_oracle_and_golden(golden):
    """pcl_batch_regroup (next row f2) against the oracle's per-frame form, which golden G12 pins to the reference's own
    discriminate / __eq_segment / __get_gmmdata: random left-right paths are forced through the Viterbi kernel by a
    one-hot emission matrix, incl. runs shorter than the number of GMM states, a unit repeated back to back, T = 1."""
    from poccala_amd import Engine
    from oracle import poccala_oracle as po
    rng = np.random.default_rng(21)
    eng = Engine(0)
    U, e = 9, S - 2
    n_list, t_list, a_list, pi_list, b_list, row_units, want_rows = [], [], [], [], [], [], []
    for u in range(U):
        L = int(rng.integers(1, 6))
        lab = list(rng.integers(0, 4, L))
        if u == 1:
            lab = [2, 2, 1]                               # the same unit twice in a row: ONE run
        N = e * len(lab) + 2
        T = 1 if u == 0 else int(rng.integers(N, 4 * N))
        # a monotone row sequence over the emitting rows, every row visited at least once when T allows it
        rows = np.sort(np.concatenate([np.arange(1, N - 1), rng.integers(1, N - 1, max(T - (N - 2), 0))]))[:T]
        if T < N - 2:
            rows = np.arange(1, 1 + T)
        A = np.full((N, N), -np.inf)
        for i in range(N - 1):
            A[i, i] = np.log(0.5)
            A[i, i + 1] = np.log(0.5)
        pi = np.full(N, -np.inf)
        pi[rows[0]] = 0.0
        B = np.full((N, T), -50.0)
        B[rows, np.arange(T)] = 0.0                       # the path is forced
        ids = np.repeat(lab, e)
        n_list.append(N); t_list.append(T); a_list.append(A); pi_list.append(pi); b_list.append(B)
        row_units.append(np.concatenate([[ids[0]], ids, [ids[-1]]]).astype(np.int32))
        want_rows.append(rows)
    b = eng.batch(n_list, t_list)                        # emissions are given: no frames behind this batch
    b.set_transitions(a_list, pi_list)
    b.set_emissions(b_list)
    b.viterbi()
    paths = b.get('path')
    fu, fk = b.regroup(row_units, e)
    for u in range(U):
        np.testing.assert_array_equal(paths[u], want_rows[u])
        unit_seq = row_units[u][paths[u]]
        np.testing.assert_array_equal(fu[u], unit_seq)
        np.testing.assert_array_equal(fk[u], po.regroup_frame_states(unit_seq, e))
    b.close()
    eng.close()


def test_regroup_batch_equals_discriminate_and_get_gmmdata(golden, tmp_path):
    """The drop-in's device path (Viterbi + pcl_batch_regroup) returns, per unit, exactly the GMM-state data the
    reference's flow builds on the host: discriminate -> blocks -> __get_gmmdata."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd import PCL_F64
    g = golden('G6_small_fix0')
    label, hmm_list = build_units(g)
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S, mix_level=4, dct_num=13, delta_1=False, delta_2=False,
                       parameters_path=str(tmp_path))
    unit_hmms = {u: hmm_list[label.index(u)] for u in set(label)}
    data = [g['x'], g['x'][::-1].copy()]
    labels = [label, label]
    got, dropped = am.regroup_batch(labels, data, unit_hmms, precision=PCL_F64)
    segs, dropped2 = am.segment_batch(labels, data, unit_hmms, precision=PCL_F64)
    assert dropped == dropped2
    for u in segs:
        want = am.get_gmmdata(segs[u])
        for k in range(S - 2):
            np.testing.assert_array_equal(got[u][k], want[k])
    # the two small helpers against the golden made by the reference's private methods
    G12 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G12_regroup.npz'))
    for n in G12['g_lens']:
        sl = am.eq_segment(G12['g_data_%d' % n], S - 2, mode='g')
        assert [len(x) for x in sl] == list(G12['g_sizes_%d' % n])
    saved = []
    am.eq_segment(G12['e_data'], list(G12['e_label']), mode='e', save=lambda unit, blk: saved.append((unit, blk)))
    assert [u for u, _ in saved] == list(G12['e_units'])
    for i, (_, blk) in enumerate(saved):
        np.testing.assert_array_equal(blk, G12['e_block_%d' % i])


def test_per_gmm_calls_do_not_disturb_a_batched_estep_in_flight(golden):
    """GMM.update_acc / point used to zero and re-upload on the shared engine.  Now the per-object
    calls run in their own context (runtime.scratch_engine): a batched E-step's statistics, model and frames on the default
    engine survive any number of interleaved per-GMM calls, and the per-GMM results are what they are without the batch."""
    from poccala_amd import PCL_F64, synth
    from poccala_amd.runtime import default_engine
    from poccala_amd.StatisticalModel.Clustering import Clustering
    mean, var, w, trans = synth.make_model(3, 4, 13, seed=91)
    frames, lens, begin = synth.make_frames(6, 40, 13, seed=92)
    labels = synth.make_labels(6, 2, 3, seed=93)
    eng = default_engine()
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_frames(frames)
    b = eng.label_batch(labels, lens, begin)
    b.score(PCL_F64)
    b.forward_backward()
    eng.stats_zero()
    b.accumulate(PCL_F64)
    before = eng.stats_download()
    # per-GMM traffic in between (a different model, different frames, its own statistics)
    rng = np.random.default_rng(94)
    g = Clustering.GMM(RecLog(), dimension=13, mix_level=5, alpha=np.full(5, 0.2), mean=rng.standard_normal((5, 13)),
                       covariance=np.array([np.diag(v) for v in rng.uniform(0.5, 2.0, (5, 13))]))
    x = rng.standard_normal((30, 13))
    p1 = np.array([g.point(x[t], log=True) for t in range(5)])
    lb = g.point_frames(x)
    np.testing.assert_array_equal(p1, lb[:5])                    # the cached upload serves every frame
    g.update_acc(np.log(np.full(30, 0.5)), lb, x)
    acc1 = g.acc.copy()
    g.update_acc(np.log(np.full(30, 0.5)), lb, x)                # log-adds onto its own running accumulator
    np.testing.assert_allclose(g.acc, acc1 + np.log(2.0), rtol=1e-12)
    after = eng.stats_download()
    for k in before:
        np.testing.assert_array_equal(after[k], before[k])
    b.accumulate(PCL_F64)                                        # and the batch can go on accumulating
    twice = eng.stats_download()
    np.testing.assert_allclose(twice['acc'], 2 * before['acc'], rtol=1e-12)
    b.close()


@pytest.mark.parametrize('case', ['G6_small_fix0', 'G6_small_fix1', 'G6_small_fix4', 'G6_n62_fix0'])
@pytest.mark.parametrize('prec', ['f64', 'f32'])
def test_worker_flow_through_the_deferred_shim(golden, tmp_path, case, prec):
    """The reference's two workers called with THEIR signatures (multi_embedded_training_1(label, data, init, show_q, load_num,
    file_count, fix_code), AcousticModel.py:884-916; multi_process_data(label, data, init, load_num, file_count, fix_code),
    :723-768) on the drop-in's deferred-batch shim: the calls queue, flush_workers() runs one batched E-step / alignment and
    writes the reference's files.  What multi_embedded_training_2's merge (init_acc: a log-sum-exp over the accumulator files)
    then reads must equal the golden per-position accumulators of the reference merged per unit; the data pickles must be the
    reference's runs (discriminate) of the golden Viterbi path.  The utterance is queued TWICE: every accumulator doubles."""
    import pickle
    from poccala_amd import PCL_F32, PCL_F64
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    g = golden(case)
    fix = int(g['fix_code'])
    x = g['x']
    label, hmm_list = build_units(g)
    am = AcousticModel(RecLog(), 'XIF_tone', state_num=S, mix_level=int(g['w_0_0'].shape[0]), dct_num=x.shape[1], delta_1=False, delta_2=False,
                       parameters_path=str(tmp_path))
    for pos, u in enumerate(label):                           # the parameter tree the workers read their unit models from
        am.save_parameter(u, hmm_list[pos])
    am.worker_precision = PCL_F64 if prec == 'f64' else PCL_F32
    # ---- E-step worker
    for n in range(2):
        am.multi_embedded_training_1(label, x, False, False, n + 1, 2, fix)
    out = am.flush_workers()
    assert out['train'][0] == 2 and out['train'][1] == 2 * len(x)
    rt, at = (1e-8, 1e-10) if prec == 'f64' else (1e-4, 1e-4)
    ln2 = np.log(2.0)
    for u in sorted(set(label)):
        pos_u = [p for p, v in enumerate(label) if v == u]
        hmm = am.init_unit(u)
        am.init_parameter(u, hmm)
        hmm.init_acc(am.unit_path(u))
        if not fix & 4:
            rk = np.logaddexp.reduce([g['ksai_acc_%d' % p] for p in pos_u], axis=0) + ln2
            rg = np.logaddexp.reduce([g['gamma_acc_%d' % p] for p in pos_u], axis=0) + ln2
            # (log-domain, un-normalised: under f32-class scoring ln P(O) itself moves by ~1e-6; 1e-4 absolute = 1e-4 relative on xi / gamma)
            fin_close(hmm.ksai_acc, rk, rtol=1e-10 if prec == 'f64' else 0.0, atol=0.0 if prec == 'f64' else 1e-4)
            fin_close(hmm.gamma_acc, rg, rtol=1e-10 if prec == 'f64' else 0.0, atol=0.0 if prec == 'f64' else 1e-4)
        if not fix & 2:
            for k in range(S - 2):
                gm = hmm.profunction[1 + k]
                gm.init_acc(am.unit_path(u))
                for name, got in (('acc', gm.acc), ('alpha_acc', np.float64(gm.alpha_acc)), ('mean_acc', gm.mean_acc),
                                  ('cov_acc', np.array(gm._GMM__covariance_acc))):
                    want = np.logaddexp.reduce([g['%s_%d_%d' % (name, p, k)] for p in pos_u], axis=0) + ln2
                    fin_close(got, want, rtol=rt, atol=at)
    # ---- alignment worker
    am.multi_process_data(label, x, False, 1, 1, 2)
    out = am.flush_workers()
    if len(set(g['vit_path_conv'].astype(str))) < len(set(label)):          # the path misses a label unit: the reference discards the utterance (:754-757)
        assert out['align'][0] == 1 and out['align'][2] == [0]
        assert not any(os.path.isdir(am.unit_path(u) + '/data') for u in set(label))
        return
    assert out['align'][0] == 1 and out['align'][2] == []
    for u in sorted(set(label)):
        d = am.unit_path(u) + '/data'
        blocks = [pickle.load(open(os.path.join(d, f), 'rb')) for f in sorted(os.listdir(d))]
        want = [x[g['disc_%s_%d' % (u, ri)]] for ri in range(int(g['disc_%s_n' % u]))]
        assert len(blocks) == len(want)
        for b_, w_ in zip(blocks, want):
            assert np.array_equal(b_, w_)

"""Host-side logic of the product package that needs no GPU: sentence-HMM construction (A7), discriminate,
the log-sum-exp helpers that merge per-unit accumulators, the MFCC filter/DCT matrices, sharding."""
import os
import numpy as np
import pytest

from oracle import mfcc_oracle as mo
from oracle import poccala_oracle as po

CASES = ['G6_small_fix0', 'G6_small_fix1', 'G6_n62_fix0', 'G8_floor']


@pytest.mark.parametrize('case', CASES)
def test_embedded_structure_matches_reference(golden, case):
    """engine.embedded_structure / embedded_row_states == AcousticModel.embedded's A, pi and row layout (G4)."""
    from poccala_amd.engine import embedded_row_states, embedded_structure
    g = golden(case)
    names = [str(u) for u in g['unit_names']]
    label = [str(u) for u in g['label']]
    trans = [g['trans_%d' % names.index(u)] for u in label]
    a, pi = embedded_structure(len(label), trans)
    np.testing.assert_array_equal(a, g['emb_A'])
    np.testing.assert_array_equal(pi, g['emb_pi'])
    ids = np.array([[names.index(u) * 3 + k for k in range(3)] for u in label])
    rows = embedded_row_states(ids)
    assert rows[0] == -1 and rows[-1] == -2 and len(rows) == len(g['emb_states'])
    unit_of_row = [str(g['emb_states'][0])] + [names[r // 3] for r in rows[1:-1]] + [str(g['emb_states'][-1])]
    assert unit_of_row == [str(s) for s in g['emb_states']]


def test_dropin_embedded_and_discriminate_on_cpu(golden):
    """AcousticModel.embedded (with stand-in unit HMMs) and discriminate need no GPU."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    g = golden('G6_small_fix0')
    names = [str(u) for u in g['unit_names']]
    label = [str(u) for u in g['label']]

    class Unit(object):
        def __init__(self, trans, b):
            self.transmat, self.B_p = trans, [b]
    e = 3
    units = []
    for pos, u in enumerate(label):
        b = np.vstack([np.zeros((1, g['emb_B'].shape[1])), g['emb_B'][1 + pos * e:1 + (pos + 1) * e], np.full((1, g['emb_B'].shape[1]), -np.inf)])
        units.append(Unit(g['trans_%d' % names.index(u)], b))
    am = AcousticModel(None, 'XIF_tone', state_num=5)
    states, a, b, pi = am.embedded(label, units, 0, 15)
    np.testing.assert_array_equal(a, g['emb_A'])
    np.testing.assert_array_equal(b, g['emb_B'])
    assert [states[i] for i in range(len(states))] == [str(s) for s in g['emb_states']]
    assert am.embedded(label, units, 0, 4)[0].shape == a.shape        # alter bitmask: transmat only
    seq = g['vit_path_conv'].astype(str)
    for u in set(label):
        runs = AcousticModel.discriminate(u, seq)
        assert len(runs) == int(g['disc_%s_n' % u])
        for ri, r in enumerate(runs):
            np.testing.assert_array_equal(r, g['disc_%s_%d' % (u, ri)])
    assert AcousticModel.VirtualState(1.).point(None, log=True) == 0.0
    assert np.isneginf(AcousticModel.VirtualState(0.).point(None, log=True))


def test_host_lse_helpers_match_reference(golden):
    from poccala_amd.StatisticalModel.util import log_sum_exp, matrix_log_sum_exp
    g = golden('G1_util')
    for i in range(5):
        ref, got = g['lse_out_%d' % i], log_sum_exp(g['lse_in_%d' % i])
        assert (got == ref) if np.isinf(ref) else abs(got - ref) <= 1e-12 * abs(ref)
    np.testing.assert_allclose(log_sum_exp(g['lse_vec_in'], vector=True), g['lse_vec_out'], rtol=1e-12)
    np.testing.assert_allclose(matrix_log_sum_exp(list(g['mlse_in']), 4), g['mlse_out_full'], rtol=1e-12)
    np.testing.assert_allclose(matrix_log_sum_exp(list(g['mlse_in']), 3), g['mlse_out_3'], rtol=1e-12)


def test_mfcc_host_matrices_match_oracle():
    from poccala_amd.StatisticalModel.AudioProcessing import dct_basis, frame_count, mel_filter_matrix
    for rate in (8000, 16000, 44100):
        np.testing.assert_allclose(mel_filter_matrix(rate), mo.mel_response(rate), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(dct_basis(26, 13), mo.dct_matrix(26, 13), rtol=1e-14)
    for n in (400, 401, 599, 600, 9000):
        assert frame_count(n, 16000) == mo.frame_geometry(n, 16000)[2]


def test_gmm_dropin_parameter_files_without_gpu(tmp_path):
    """save/init of parameters and accumulators (T3) are pure file logic."""
    from poccala_amd.StatisticalModel.Clustering import Clustering
    rng = np.random.default_rng(0)
    g = Clustering.GMM(None, dimension=5, mix_level=3, alpha=rng.dirichlet(np.ones(3)), mean=rng.standard_normal((3, 5)),
                       covariance=np.array([np.diag(v) for v in rng.uniform(0.5, 2, (3, 5))]), gmm_id=1)
    g.acc = np.log(rng.uniform(0.1, 1, 3))
    g.alpha_acc = 0.3
    g.mean_acc = rng.standard_normal((3, 5))
    g.covariance_acc = list(rng.standard_normal((3, 5)))
    g.save_parameter(str(tmp_path))
    g.save_acc(str(tmp_path))
    g.save_acc(str(tmp_path))                               # same second: must not overwrite (the reference's race)
    h = Clustering.GMM(None, dimension=5, mix_level=3, gmm_id=1)
    h.init_parameter(str(tmp_path))
    np.testing.assert_array_equal(h.mean, g.mean)
    np.testing.assert_array_equal(h.diag_variance(), g.diag_variance())
    h.init_acc(str(tmp_path))
    np.testing.assert_allclose(h.acc, g.acc + np.log(2), rtol=1e-12)           # two identical files merged
    np.testing.assert_allclose(h.alpha_acc, 0.3 + np.log(2), rtol=1e-12)
    assert h.covariance_acc == 100.0                        # the reference's getter returns the bias (T2)


def test_decode_result_unpacking_and_word_chain_without_gpu():
    """Batch.decode_unpack (raw result arrays -> per-utterance dicts) and Decoder._report (history chain -> words, as
    transfer() reports them, Decoder.py:183-186) are host-only; crafted arrays: two utterances, one with a two-word history
    and a final token on a word-end node, one whose best token has no history and sits inside a word."""
    from poccala_amd import Decoder
    from poccala_amd.engine import Batch
    tree = dict(words={3: ['ni', 'ni2'], 5: ['hao'], 7: ['ma']}, node_word=np.array([0, 0, 0, 1, 0, 1, 0, 1]))
    nf = np.array([2, 1], np.int32)
    node = np.array([[7, 2], [4, 0]], np.int32)
    score = np.array([[-10.5, -11.0], [-3.25, 0.0]])
    hist = np.array([[1, 0], [-1, 0]], np.int32)
    hn = np.array([2, 0], np.int32)
    hp = np.array([[-1, 0, 0], [0, 0, 0]], np.int32)              # entry 1 -> entry 0 -> start
    hnode = np.array([[3, 5, 0], [0, 0, 0]], np.int32)
    nt = np.array([[4, 5, 6], [2, 9, 9]], np.int32)
    ov = np.array([0, 1], np.int32)
    T = np.array([3, 1], np.int32)
    res = Batch.decode_unpack((nf, node, score, hist, hn, hp, hnode, nt, ov, T))
    assert res[0]['final'] == [(7, -10.5, 1), (2, -11.0, 0)] and res[0]['history'] == [(-1, 3), (0, 5)]
    assert res[0]['n_tokens'].tolist() == [4, 5, 6] and res[0]['overflow'] is False
    assert res[1]['final'] == [(4, -3.25, -1)] and res[1]['history'] == [] and res[1]['n_tokens'].tolist() == [2] and res[1]['overflow'] is True
    rep = Decoder._report(res, tree)
    assert rep[0][0] == [['ni', 'ni2'], ['hao'], ['ma']] and rep[0][1] == -10.5
    assert rep[1][0] == [] and rep[1][1] == -3.25


def test_save_acc_file_writes_np_save_bytes_and_never_overwrites(tmp_path):
    """The worker shim's accumulator writer (util.save_acc_file): byte-for-byte what np.save writes for the reference's
    accumulator shapes (incl. the 0-d alpha accumulator), and a second file of the same second gets a new name."""
    import io
    from poccala_amd.StatisticalModel.util import save_acc_file
    d = str(tmp_path / 'u' / 'GMM_0' / 'acc')
    names = []
    for v in (np.float64(3.5), 2.0, np.arange(6.).reshape(2, 3), np.float32([1, 2]), np.full(4, -np.inf)):
        f = save_acc_file(d, 'GMM_acc', 1700000000, v)
        names.append(os.path.basename(f))
        ref = io.BytesIO()
        np.save(ref, v)
        assert open(f, 'rb').read() == ref.getvalue()
        assert np.load(f).shape == np.shape(v)
    assert names == ['GMM_acc_1700000000.npy'] + ['GMM_acc_1700000000%03d.npy' % k for k in range(1, 5)]
    open(d + '/GMM_acc_1700000001.npy', 'wb').close()                    # a file some other worker wrote this second
    assert os.path.basename(save_acc_file(d, 'GMM_acc', 1700000001, np.zeros(2))) == 'GMM_acc_1700000001001.npy'
    t = np.arange(6.).reshape(2, 3).T                                     # not C-contiguous: written in C order
    assert np.array_equal(np.load(save_acc_file(d, 'GMM_acc', 1700000002, t)), t)


def test_load_unit_has_the_reference_semantics(tmp_path, monkeypatch):
    """AcousticModel.load_unit(unit_type=None) (AcousticModel.py:134-161): `$unit_file_path/<unit_type>`, the unit type taken over when
    given, UnitFileExistsError for a missing file, the parameter directory of the type created; unit_file= reads a path."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd.Exceptions import UnitFileExistsError
    units = tmp_path / 'Unit'
    units.mkdir()
    (units / 'XIF_tone').write_text('a description line\nb,p,m\na0,a1\n')
    (units / 'IF').write_text('another\nzh,ch\n')
    monkeypatch.setenv('unit_file_path', str(units))
    params = tmp_path / 'PARAMS'
    am = AcousticModel(None, 'XIF_tone', parameters_path=str(params))
    am.load_unit()
    assert am.loaded_units == ['b', 'p', 'm', 'a0', 'a1'] and (params / 'XIF_tone').is_dir()
    am.load_unit('IF')                                       # the unit type is taken over (AcousticModel.py:140-141)
    assert am.loaded_units[-2:] == ['zh', 'ch'] and (params / 'IF').is_dir()
    assert am.unit_path('zh').endswith('/IF/zh')
    with pytest.raises(UnitFileExistsError):
        am.load_unit('nothing_here')
    am2 = AcousticModel(None, 'XIF_tone', parameters_path=str(params))
    am2.load_unit(unit_file=str(units / 'IF'))
    assert am2.loaded_units == ['zh', 'ch']


def test_gmm_data_holder_methods():
    """Clustering.GMM.add_data / clear_data / data setter (Clustering.py:106-120, 166-174)."""
    from poccala_amd.StatisticalModel.Clustering import Clustering
    g = Clustering.GMM(None, dimension=3, mix_level=2)
    assert g.data is None
    g.add_data([[1., 2., 3.], [4., 5., 6.]])
    g.add_data(np.ones((1, 3)))
    assert g.data.shape == (3, 3)
    g.clear_data()
    assert g.data is None
    g.data = [[0., 0., 0.]]
    assert g.data.shape == (1, 3)


def test_worker_shim_copies_what_it_queues_and_warns_when_dropped(tmp_path):
    """The deferred workers (AcousticModel.multi_embedded_training_1 / multi_process_data) copy the caller's data at call time -- the
    reference's Pool pickled them (AcousticModel.py:865-870), a caller may refill its buffer -- keep every call's own (fix_code, init),
    and a model dropped with calls still queued says so instead of losing them silently."""
    import gc
    import warnings
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    am = AcousticModel(None, 'XIF_tone', parameters_path=str(tmp_path))
    am.flush_frames = 1 << 30
    buf = np.zeros((7, 39), dtype=np.float32)
    am.multi_embedded_training_1(['a', 'b'], buf, True, False, 1, 2, 2)
    buf[:] = 9.0                                             # the caller refills its buffer
    am.multi_embedded_training_1(['a'], buf, False, False, 2, 2, 0)
    q = am._AcousticModel__queued['train']
    assert [(x[2], x[3]) for x in q] == [(True, 2), (False, 0)]
    assert q[0][1].max() == 0.0 and q[1][1].min() == 9.0 and q[0][1] is not buf
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        am._AcousticModel__queued = {'train': q, 'align': []}
        del am
        gc.collect()
    assert any('never flushed' in str(x.message) for x in w)


def test_the_builds_classes_read_their_own_tree_as_the_reference_did(golden, tmp_path):
    """G17's inputs through the BUILD's host classes alone (no GPU): AcousticModel.init_unit / save_parameter / save_batch_acc write the
    tree, fresh LHMM + GMM objects read it back (init_parameter, init_acc: the merge of the two accumulator files) and re-estimate
    (update_param) -- and arrive at what the REFERENCE arrived at from the same tree (fixture G17)."""
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    g = golden('G17_tree')
    units = [str(u) for u in g['units']]
    S, e = 5, 3
    M, D = g['mean_0_0'].shape
    am = AcousticModel(None, 'XIF_tone', parameters_path=str(tmp_path), state_num=S, mix_level=M, dct_num=D, delta_1=False, delta_2=False)
    hmms = {}
    for ui, u in enumerate(units):
        h = am.init_unit(u)
        h.transmat[:] = g['trans_%d' % ui]
        for k in range(e):
            gm = h.profunction[1 + k]
            gm.alpha, gm.mean = g['w_%d_%d' % (ui, k)].copy(), g['mean_%d_%d' % (ui, k)].copy()
            gm.covariance = np.array([np.diag(v) for v in g['var_%d_%d' % (ui, k)]])
        am.save_parameter(u, h)
        hmms[u] = h
    for bi in range(2):
        stats = {key: g['batch%d_%s' % (bi, key)] for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc')}
        hacc = {u: (g['batch%d_ksai_%d' % (bi, ui)], g['batch%d_gamma_%d' % (bi, ui)]) for ui, u in enumerate(units)}
        am.save_batch_acc(stats, hacc, hmms)
    for ui, u in enumerate(units):
        h = am.init_unit(u)
        am.init_parameter(u, h)
        np.testing.assert_array_equal(h.transmat, g['trans_%d' % ui])
        h.init_acc(am.unit_path(u))
        for k in range(e):
            h.profunction[1 + k].init_acc(am.unit_path(u))
        np.testing.assert_allclose(h.ksai_acc, g['ref_ksai_acc_%d' % ui], rtol=1e-12)
        np.testing.assert_allclose(h.gamma_acc, g['ref_gamma_acc_%d' % ui], rtol=1e-12)
        h.update_param(c_covariance=float(g['c_covariance']))
        np.testing.assert_allclose(h.transmat, g['new_trans_%d' % ui], rtol=1e-12, atol=1e-300)
        for k in range(e):
            gm = h.profunction[1 + k]
            np.testing.assert_allclose(gm.acc, g['ref_acc_%d_%d' % (ui, k)], rtol=1e-12)
            np.testing.assert_allclose(gm.alpha, g['new_w_%d_%d' % (ui, k)], rtol=1e-10)
            np.testing.assert_allclose(gm.mean, g['new_mean_%d_%d' % (ui, k)], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(gm.diag_variance(), g['new_var_%d_%d' % (ui, k)], rtol=1e-9)

"""The token-passing decoder (BASELINE config 5's decode half) on the GPU against its CPU restatement
(oracle/decoder_oracle.py: recursion, pruning, frame loop and in-word hand-over pinned by golden G14 from the reference's
own Decoder.py; the completion rules D1-D5 unpinned): bit-exact scores, nodes, histories and token counts given identical
emissions, plus the properties the recursion guarantees."""
import json
import os

import numpy as np
import pytest

from oracle import decoder_oracle as do
from oracle import poccala_oracle as po

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
S, E = 5, 3


@pytest.fixture(scope='module')
def eng():
    from poccala_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def lex(tmp_path_factory):
    from poccala_amd.Lexicon import PinYin, PronunciationLexicon
    g = json.load(open(os.path.join(HERE, 'golden', 'G13_lexicon.json')))
    path = str(tmp_path_factory.mktemp('lex') / 'Mandarin.dat')
    with open(path, 'w') as f:
        for k, v in g['table'].items():
            f.write('%s\t%s\n' % (k, v))
    py = PinYin(path)
    lx = PronunciationLexicon()
    lx.generate_lexicon(words=g['words'], pinyin=py)
    units = sorted({u for w in g['words'] for r in py.word2pinyin(w) for x in r for u in x.split(',')})
    return lx, units, lx.compile({u: i for i, u in enumerate(units)})


def model_for(units, M, D, seed, dense=False):
    from poccala_amd import synth
    mean, var, w, trans = synth.make_model(len(units), M, D, seed=seed)
    if dense:
        rng = np.random.default_rng(seed + 1)
        trans = []
        for _ in units:
            a = np.zeros((S, S))
            a[0, 1:3] = [0.7, 0.3]
            a[1:-1, 1:] = rng.dirichlet(np.ones(S - 1), size=E)
            trans.append(a)
    return mean, var, w, np.stack(trans)


@pytest.mark.parametrize('dense,beam,cap', [(False, 0.85, 4096), (True, 0.85, 4096), (False, 1.0, 4096), (False, 0.5, 4096), (False, 0.85, 230), (True, 0.7, 5)])
def test_decode_matches_cpu_restatement_bit_for_bit(eng, lex, dense, beam, cap):
    from poccala_amd import PCL_F64, synth
    lx, units, tree = lex
    mean, var, w, trans = model_for(units, 3, 13, 11, dense)
    frames, lens, begin = synth.make_frames(4, 90, 13, seed=12, ragged=True)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    got = b.decode(beam=beam, candidate=6, max_tokens=cap)
    b.close()
    for u in range(4):
        b_all = B[u][1:-1]                                       # (J, T): the same emissions the device decoded from
        trace, info = [], {}
        fin, hist = do.decode(tree, list(trans), b_all, beam=beam, candidate=6, max_tokens=cap, trace=trace, info=info)
        g = got[u]
        assert np.array_equal(g['n_tokens'], np.array(trace)), (u, g['n_tokens'][:12], trace[:12])
        assert g['history'] == [(int(p), int(n)) for p, n in hist]
        assert [(n, h) for n, _, h in g['final']] == [(n, h) for n, _, h in fin]
        assert [s for _, s, _ in g['final']] == [float(s) for _, s, _ in fin]          # bit-exact float64
        assert g['overflow'] == bool(info.get('overflow')) and (cap < 4096) == g['overflow']
    # the words come out through the history chain
    words = do.words_of(fin[0], hist, tree)
    assert all(isinstance(wl, list) and wl for wl in words)


@pytest.mark.parametrize('beam,cap', [(0.85, 4096), (0.6, 300)])
def test_both_decode_kernels_agree_on_left_to_right_units(eng, lex, monkeypatch, beam, cap):
    """Left-to-right unit matrices with uneven self-loop weights: the lane-per-token kernel (hmm_decode_lr.hip, the default for
    such units) and the general kernel (PCL_DEC_GENERAL=1: 8 lanes per token, any matrices) both equal the restatement bit for
    bit -- the terms the general kernel adds with ln A = -inf change no bit (Decoder.py:278-283)."""
    from poccala_amd import PCL_F64, synth
    lx, units, tree = lex
    mean, var, w, _ = model_for(units, 3, 13, 71)
    rng = np.random.default_rng(72)
    trans = []
    for _ in units:
        a = np.zeros((S, S))
        a[0, 1] = 1.0
        for r in range(1, S - 1):
            x = rng.uniform(0.05, 0.95)
            a[r, r], a[r, r + 1] = x, 1.0 - x
        trans.append(a)
    trans = np.stack(trans)
    frames, lens, begin = synth.make_frames(3, 80, 13, seed=73, ragged=True)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    fast = b.decode(beam=beam, candidate=6, max_tokens=cap)
    monkeypatch.setenv('PCL_DEC_GENERAL', '1')
    general = b.decode(beam=beam, candidate=6, max_tokens=cap)
    monkeypatch.delenv('PCL_DEC_GENERAL')
    b.close()
    for u in range(3):
        trace, info = [], {}
        fin, hist = do.decode(tree, list(trans), B[u][1:-1], beam=beam, candidate=6, max_tokens=cap, trace=trace, info=info)
        for g in (fast[u], general[u]):
            assert np.array_equal(g['n_tokens'], np.array(trace))
            assert g['history'] == [(int(p), int(n)) for p, n in hist]
            assert [(n, h) for n, _, h in g['final']] == [(n, h) for n, _, h in fin]
            assert [s for _, s, _ in g['final']] == [float(s) for _, s, _ in fin]
            assert g['overflow'] == bool(info.get('overflow'))


def test_decode_many_tokens_finish_in_one_frame(eng):
    """Every state scores alike and every unit hurries forward (self-loop 0.1), so all tokens made in one frame finish in one
    frame: thousands of donors at once, more per wavefront than the lane-per-token kernel keeps in LDS (its donor lists
    spill to HBM above 128 per wave), word-end donors and first-character re-seeding in the same frames, the capacity hit
    -- bit for bit the restatement (Decoder.py:91-143)."""
    from poccala_amd import PCL_F64, synth
    n_units = 183
    tree, lx = synth.make_pronunciation_tree(3000, n_units, seed=81)
    mean, var, w, _ = synth.make_model(n_units, 2, 13, seed=82)
    mean[:], var[:], w[:] = mean[0], var[0], w[0]                 # one GMM for every state
    a = np.zeros((S, S))
    a[0, 1] = 1.0
    for r in range(1, S - 1):
        a[r, r], a[r, r + 1] = 0.1, 0.9
    trans = np.stack([a] * n_units)
    frames, lens, begin = synth.make_frames(2, 22, 13, seed=83)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    got = b.decode(beam=1.0, candidate=4, max_tokens=8192)
    b.close()
    swing = 0
    for u in range(2):
        trace, info = [], {}
        fin, hist = do.decode(tree, list(trans), B[u][1:-1], beam=1.0, candidate=4, max_tokens=8192, trace=trace, info=info)
        g = got[u]
        assert np.array_equal(g['n_tokens'], np.array(trace)), (g['n_tokens'], trace)
        assert g['history'] == [(int(p), int(n)) for p, n in hist]
        assert [(n, h) for n, _, h in g['final']] == [(n, h) for n, _, h in fin]
        assert [s for _, s, _ in g['final']] == [float(s) for _, s, _ in fin]
        assert g['overflow'] == bool(info.get('overflow'))
        swing = max(swing, int(np.abs(np.diff(np.array(trace))).max()))
    assert swing > 2000                                           # (the frames this test is about did occur)


@pytest.mark.parametrize('n_units', [350, 600])
def test_decode_with_a_large_unit_inventory(eng, n_units):
    """More GMM states than the lane-per-token kernel prefetches per frame (J + 2 > 1024: the emission row is staged without the
    register prefetch) and, at 600 units, more LDS than the default 64 KB (unit table + two emission rows: the launch raises the
    kernel's dynamic-LDS limit) -- bit for bit the restatement (Decoder.py:91-167,250-288)."""
    from poccala_amd import PCL_F64, synth
    tree, lx = synth.make_pronunciation_tree(300, n_units, seed=61)
    mean, var, w, trans = synth.make_model(n_units, 2, 13, seed=62)
    trans = np.stack(trans)
    frames, lens, begin = synth.make_frames(2, 40, 13, seed=63, ragged=True)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    got = b.decode(beam=0.85, candidate=5, max_tokens=2048)
    b.close()
    for u in range(2):
        trace, info = [], {}
        fin, hist = do.decode(tree, list(trans), B[u][1:-1], beam=0.85, candidate=5, max_tokens=2048, trace=trace, info=info)
        g = got[u]
        assert np.array_equal(g['n_tokens'], np.array(trace))
        assert g['history'] == [(int(p), int(n)) for p, n in hist]
        assert [(n, h) for n, _, h in g['final']] == [(n, h) for n, _, h in fin]
        assert [s for _, s, _ in g['final']] == [float(s) for _, s, _ in fin]
        assert g['overflow'] == bool(info.get('overflow'))


def test_decode_properties(eng, lex):
    """A one-frame utterance returns exactly the first-step scores of the first-character nodes (ln pi + the best of entry
    row 0 and the node's emissions); the same call twice gives the same bits (no atomics, no timing dependence); and a
    beam of 1 never prunes: every frame keeps all unfinished tokens."""
    from poccala_amd import PCL_F64, synth
    lx, units, tree = lex
    mean, var, w, trans = model_for(units, 2, 13, 21)
    frames, lens, begin = synth.make_frames(3, 60, 13, seed=22, ragged=True)
    lens[2] = 1
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    one = b.decode(beam=0.85, candidate=5)
    two = b.decode(beam=0.85, candidate=5)
    free = b.decode(beam=1.0, candidate=5)
    b.close()
    for u in range(3):
        assert one[u]['final'] == two[u]['final'] and one[u]['history'] == two[u]['history']
        assert np.array_equal(one[u]['n_tokens'], two[u]['n_tokens'])
    # T = 1: nothing but the first step of every first-character node
    col = B[2][1:-1][:, 0]
    best = max(np.log(1.0 / (E * n + 2)) + max(0.0, max(col[int(uu) * E + k] for uu in row[:n] for k in range(E)))
               for row, n in ((tree['node_units'][r], tree['node_nunits'][r]) for r in tree['roots']))
    np.testing.assert_allclose(free[2]['final'][0][1], best, rtol=1e-12)
    assert free[2]['n_tokens'].tolist() == [len(tree['roots'])]
    assert one[2]['final'] == free[2]['final']


def test_decode_batch_dropin_and_errors(eng, lex):
    from poccala_amd import Decoder, PCL_F32, PoccalaHipError, synth
    lx, units, tree0 = lex
    mean, var, w, trans = model_for(units, 2, 13, 31)
    tree = Decoder.load_inventory(eng, units, mean, var, w, trans, lx)
    assert tree['names'] == tree0['names']
    frames, lens, begin = synth.make_frames(2, 50, 13, seed=32)
    out = Decoder.decode_batch([frames[:50], frames[50:]], tree, engine=eng, precision=PCL_F32)
    assert len(out) == 2 and all(np.isfinite(s) for _, s, _ in out)
    # a label-built (not all-state) batch is refused
    eng.load_frames(frames)
    b = eng.label_batch([[0, 1]], [50], [0])
    b.score(PCL_F32)
    with pytest.raises(PoccalaHipError):
        b.decode()
    b.close()
    # an inventory of the same shape keeps the tree (new transitions reach the decoder), a different one drops it
    eng.load_units(trans)
    b = eng.all_state_batch([50], [0])
    b.score(PCL_F32)
    assert len(b.decode()) == 1
    b.close()
    eng.load_units(trans[:-1])
    b = eng.all_state_batch([50], [0])
    b.score(PCL_F32)
    with pytest.raises(PoccalaHipError):
        b.decode()
    b.close()


def test_decode_follows_a_transition_mstep(eng, lex):
    """the decoder's device copy of ln A must follow pcl_mstep_transitions (and a re-upload of the units): train the
    transitions on a label-built batch in the SAME context, then decode -- bit for bit the restatement run with the NEW
    matrices, and different from the decode before the M-step."""
    from poccala_amd import PCL_F64, synth
    lx, units, tree = lex
    mean, var, w, trans = model_for(units, 3, 13, 51)
    frames, lens, begin = synth.make_frames(3, 70, 13, seed=52, ragged=True)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    eng.load_lexicon(tree)
    eng.load_frames(frames)

    def decode_all():
        b = eng.all_state_batch(lens, begin)
        b.score(PCL_F64)
        out, B = b.decode(candidate=4), b.get('B')
        b.close()
        return out, B
    before, B = decode_all()
    labels = synth.make_labels(3, 4, len(units), seed=53)
    lb = eng.label_batch(labels, lens, begin)
    lb.score(PCL_F64)
    lb.forward_backward(fix_pi=False)
    eng.hmm_acc_zero()
    lb.accumulate_hmm()
    eng.mstep_transitions()
    lb.close()
    new_trans = eng.units_download()
    assert not np.allclose(new_trans, trans)
    after, B2 = decode_all()
    assert all(np.array_equal(x, y) for x, y in zip(B, B2))      # same emissions: only the transitions moved
    eng.load_units(new_trans)                                     # the same matrices with NumPy's logarithm (the library's own
    exact, _ = decode_all()                                       # refresh uses libm's: equal up to the last bit of ln A)
    changed = False
    for u in range(3):
        fin, hist = do.decode(tree, list(new_trans), B2[u][1:-1], candidate=4, max_tokens=4096)
        for got, tight in ((after[u], False), (exact[u], True)):
            assert [(n, h) for n, _, h in got['final']] == [(n, h) for n, _, h in fin]
            assert got['history'] == [(int(p), int(n)) for p, n in hist]
            if tight:
                assert [s for _, s, _ in got['final']] == [float(s) for _, s, _ in fin]
            else:
                np.testing.assert_allclose([s for _, s, _ in got['final']], [float(s) for _, s, _ in fin], rtol=1e-13)
        changed |= [s for _, s, _ in after[u]['final']] != [s for _, s, _ in before[u]['final']]
    assert changed
    # the same through a re-upload of an inventory of the same shape (the tree stays)
    eng.load_units(trans)
    again, _ = decode_all()
    for u in range(3):
        assert again[u]['final'] == before[u]['final'] and again[u]['history'] == before[u]['history']


def test_lexicon_upload_rejects_malformed_trees(eng, lex):
    from poccala_amd import PoccalaHipError
    lx, units, tree = lex
    mean, var, w, trans = model_for(units, 2, 13, 61)
    eng.load_model(mean, var, w)
    eng.load_units(trans)
    bad = dict(tree); bad['child_ptr'] = tree['child_ptr'].copy(); bad['child_ptr'][0] = -1
    with pytest.raises(PoccalaHipError):
        eng.load_lexicon(bad)
    bad = dict(tree); bad['roots'] = np.concatenate([tree['roots'], tree['roots'][:1]])
    with pytest.raises(PoccalaHipError):
        eng.load_lexicon(bad)
    bad = dict(tree); bad['node_units'] = np.concatenate([tree['node_units'], tree['node_units'][:, :1]], axis=1)
    with pytest.raises(ValueError):
        eng.load_lexicon(bad)
    eng.load_lexicon(tree)


@pytest.mark.parametrize('D', [13, 20])
def test_decode_stream_equals_chunk_by_chunk(eng, lex, D):
    """The streaming pipeline (frames of chunk k+1 on the copy stream, scoring of chunk k on the main stream, token passing
    of chunk k-1 on the second stream, batches reused per chunk shape) returns, chunk by chunk, exactly what decode_batch
    returns for that chunk alone -- including a chunk of a different shape in the middle and a one-utterance tail.  D = 20 has no
    kernel instance of its own: the staged rows are padded to 26 on their way into the frame slot, as the resident
    upload pads them on the host."""
    from poccala_amd import Decoder, PCL_F32, synth
    lx, units, tree0 = lex
    mean, var, w, trans = model_for(units, 3, D, 41)
    tree = Decoder.load_inventory(eng, units, mean, var, w, trans, lx)
    rng = np.random.default_rng(42)
    shapes = [[40, 40, 40], [40, 40, 40], [25, 60], [40, 40, 40], [40, 40, 40], [33]]
    chunks = [[rng.standard_normal((t, D)).astype(np.float32) for t in sh] for sh in shapes]
    ref = [Decoder.decode_batch(ch, tree, engine=eng, precision=PCL_F32, candidate=4, max_tokens=600) for ch in chunks]
    got = list(Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, candidate=4, max_tokens=600))
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert len(g) == len(r)
        for (gw, gs, gd), (rw, rs, rd) in zip(g, r):
            assert gw == rw and gs == rs
            assert gd['final'] == rd['final'] and gd['history'] == rd['history'] and gd['overflow'] == rd['overflow']
            assert np.array_equal(gd['n_tokens'], rd['n_tokens'])
    # the resident path is untouched by the slots: a plain upload afterwards scores as before
    one = Decoder.decode_batch(chunks[0], tree, engine=eng, precision=PCL_F32, candidate=4, max_tokens=600)
    assert [x[1] for x in one] == [x[1] for x in ref[0]]
    assert list(Decoder.decode_stream(iter([]), tree, engine=eng)) == []


def test_c5_decode_at_shard_shape(eng):
    """BASELINE config 5's decode half at its real shape (Decoder.py:91-167): the per-GPU shard of 417 utterances x 300 frames,
    all 549 states x 4096 mixtures scored, a 20 k-word / 90 k-node pronunciation tree, 8192 live tokens per utterance.
    Too big for the CPU restatement as a whole, so: properties at full size -- live tokens within the cap, the overflow
    flag consistent (an utterance that never overflowed decodes identically under a larger cap), acyclic word-ending
    history chains, two runs bit-identical, the streamed pipeline equal to the resident run -- and one utterance x 72 frames
    bit for bit against oracle/decoder_oracle.py (recursion / pruning / loop pinned by golden G14; D1-D5 the builder's
    completion of the reference's dead code)."""
    from poccala_amd import Decoder, PCL_F32, synth
    c = synth.CONFIGS['C5shard']
    cap = 8192
    tree, lx = synth.make_pronunciation_tree(20000, c['units'])
    assert len(tree['names']) > 80000
    from _models import full_size_model
    mean, var, w, trans = full_size_model(c, 5)
    frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'], seed=6)
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F32)
    one = b.decode(max_tokens=cap, candidate=5)
    two = b.decode(max_tokens=cap, candidate=5)
    big = b.decode(max_tokens=16384, candidate=5)
    b.close()
    n_roots = len(tree['roots'])
    calm = 0
    for u in range(c['U']):
        r = one[u]
        assert r['final'] == two[u]['final'] and r['history'] == two[u]['history'] and np.array_equal(r['n_tokens'], two[u]['n_tokens'])
        nt = r['n_tokens']
        assert len(nt) == c['T'] and nt[0] == min(n_roots, cap) and nt.min() > 0 and nt.max() <= cap
        assert len(r['final']) == min(5, int(nt[-1]))
        sc = [s for _, s, _ in r['final']]
        assert sc == sorted(sc, reverse=True) and np.isfinite(sc).all()
        for i, (prev, node) in enumerate(r['history']):              # chains point backwards (acyclic) and end words
            assert -1 <= prev < i and tree['node_word'][node]
        assert all(-1 <= h < len(r['history']) for _, _, h in r['final'])
        if not r['overflow']:                                        # the capacity never bound: a larger one changes nothing
            calm += 1
            assert not big[u]['overflow'] and r['final'] == big[u]['final'] and np.array_equal(nt, big[u]['n_tokens'])
    print('C5 shard decode: %d of %d utterances within %d tokens, mean %.0f / max %d live tokens'
          % (calm, c['U'], cap, np.mean([r['n_tokens'].mean() for r in one]), max(r['n_tokens'].max() for r in one)))
    # streamed == resident (three chunks, the pipeline of Decoder.decode_stream)
    per = (c['U'] + 2) // 3
    chunks = [[frames[begin[u]:begin[u] + lens[u]] for u in range(k * per, min(c['U'], (k + 1) * per))] for k in range(3)]
    got = [x for ch in Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=cap, candidate=5) for x in ch]
    assert len(got) == c['U']
    for u in range(c['U']):
        assert got[u][2]['final'] == one[u]['final'] and got[u][2]['history'] == one[u]['history']
        assert np.array_equal(got[u][2]['n_tokens'], one[u]['n_tokens'])
    # one utterance against the restatement, from the emissions the device decoded
    tcut = 72
    eng.load_frames(frames)
    b = eng.all_state_batch(np.array([tcut], dtype=np.int32), np.array([begin[7]], dtype=np.int64))
    b.score(PCL_F32)
    B = b.get('B')[0]
    g = b.decode(max_tokens=cap, candidate=5)[0]
    b.close()
    trace, info = [], {}
    fin, hist = do.decode(tree, list(trans), B[1:-1], max_tokens=cap, candidate=5, trace=trace, info=info)
    assert np.array_equal(g['n_tokens'], np.array(trace))
    assert g['history'] == [(int(p), int(n)) for p, n in hist]
    assert [(n, h) for n, _, h in g['final']] == [(n, h) for n, _, h in fin]
    assert [s for _, s, _ in g['final']] == [float(s) for _, s, _ in fin]
    assert g['overflow'] == bool(info.get('overflow'))
    assert np.array_equal(g['n_tokens'], one[7]['n_tokens'][:tcut])           # a prefix of the full-length run


def test_c5_full_corpus_streamed():
    """BASELINE config 5 at the size it states: the 1M-frame corpus (24 chunks x 139 utterances x 300 frames = 1,000,800 frames)
    streamed through Decoder.decode_stream -- H2D of chunk k+1 beside the all-state scoring of chunk k beside the token passing of
    chunk k-1 (the reference reads one utterance at a time inside its worker, AcousticModel.py:723-768; Decoder.py:91-167).
    Properties over the whole stream (every utterance answered, live tokens within the cap, candidates ordered, history chains
    acyclic and word-ending) and streamed = resident bit for bit on three chunks; then a RAGGED stream (a new chunk shape every
    time, hence a new batch per chunk): the same equality, and the engine's page-locked memory stays bounded."""
    from poccala_amd import Decoder, Engine, PCL_F32, synth
    c = synth.CONFIGS['C5shard']
    cap, per, n_chunks = 8192, 139, 24
    tree, lx = synth.make_pronunciation_tree(20000, c['units'])
    from _models import full_size_model
    mean, var, w, trans = full_size_model(c, 5)
    eng = Engine(0)
    try:
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        eng.load_lexicon(tree)

        def corpus(ragged, n):
            for k in range(n):
                frames, lens, begin = synth.make_frames(per, c['T'], c['D'], seed=7000 + k, ragged=ragged)
                yield [frames[begin[u]:begin[u] + lens[u]] for u in range(per)]

        def check(res, lens):
            assert len(res) == len(lens)
            for (words, score, r), t in zip(res, lens):
                nt = r['n_tokens']
                assert len(nt) == t and nt.min() > 0 and nt.max() <= cap
                sc = [s for _, s, _ in r['final']]
                assert len(sc) == min(5, int(nt[-1])) and sc == sorted(sc, reverse=True) and np.isfinite(sc).all()
                for i, (prev, node) in enumerate(r['history']):
                    assert -1 <= prev < i and tree['node_word'][node]

        for ragged, n, keep in ((False, n_chunks, (0, 13, 23)), (True, 8, (2, 7))):
            kept, frames_done, pinned_mid = {}, 0, None
            for k, res in enumerate(Decoder.decode_stream(corpus(ragged, n), tree, engine=eng, precision=PCL_F32, max_tokens=cap, candidate=5)):
                lens = [len(r[2]['n_tokens']) for r in res]
                check(res, lens)
                frames_done += sum(lens)
                if k in keep:
                    kept[k] = res
                if k == 3:
                    pinned_mid = eng.pinned_bytes()
            assert k == n - 1
            if not ragged:
                assert frames_done == n_chunks * per * c['T'] == 1000800
            else:
                assert eng.pinned_bytes() <= 2 * pinned_mid + (1 << 20), (pinned_mid, eng.pinned_bytes())   # grow-only by doubling: bounded
            for k, res in kept.items():                                  # the same chunk, resident: identical results
                chunk = list(corpus(ragged, k + 1))[k]
                ref = Decoder.decode_batch(chunk, tree, engine=eng, precision=PCL_F32, max_tokens=cap, candidate=5)
                for a, b in zip(res, ref):
                    assert a[0] == b[0] and a[1] == b[1] and a[2]['final'] == b[2]['final'] and a[2]['history'] == b[2]['history']
                    assert np.array_equal(a[2]['n_tokens'], b[2]['n_tokens']) and a[2]['overflow'] == b[2]['overflow']
    finally:
        eng.close()

"""oracle/decoder_oracle.py held to golden G14 -- what the reference's Decoder.py itself produced when its runnable pieces
were driven in the build container (tests/golden/make_golden_decoder.py): Token.viterbi (Decoder.py:250-288), pruning
(:159-167), the token_passing frame loop (:91-111) and passing_in_word (:114-143).  CPU only."""
import os

import numpy as np
import pytest

from oracle import decoder_oracle as do
from oracle import poccala_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(HERE, 'golden', 'G14_decoder.npz'))


def all_state_emissions(g, x):
    """ln b_j(o_t) of every GMM state (J = 3 units, T) with the pinned scoring oracle (A4, golden G2/G3)."""
    rows = []
    for u in range(int(g['n_units'])):
        for k in range(int(g['S']) - 2):
            rows.append(po.gmm_point(x, g['mean'][u, k], g['var'][u, k], g['weight'][u, k]))
    return np.array(rows)


def test_token_viterbi_matches_reference(g):
    """Decoder.py:250-288: p, score, mark per frame; the emission column the reference's embedded() built."""
    trans = list(g['trans'])
    for w in range(int(g['tok_n'])):
        label = [int(u) for u in g['tok%d_label' % w]]
        b_all = all_state_emissions(g, g['tok%d_x' % w])
        tok = do.Token(float(g['tok%d_score0' % w]), 0, label, trans)
        tok.finished_rule = 'source'
        n = 3 * len(label) + 2
        for t in range(b_all.shape[1]):
            col = do.emission_column(label, b_all, t)
            ref_col = g['tok%d_bcol' % w][t]
            assert col[0] == ref_col[0] == 0.0 and np.isneginf(col[-1]) and np.isneginf(ref_col[-1])
            np.testing.assert_allclose(col[1:-1], ref_col[1:-1], rtol=1e-12)
            ret = tok.viterbi(col)
            np.testing.assert_allclose(tok.p[:-1], g['tok%d_p' % w][t][:-1], rtol=1e-12)
            assert np.isneginf(tok.p[-1]) and np.isneginf(g['tok%d_p' % w][t][-1])
            np.testing.assert_allclose(tok.score, g['tok%d_score' % w][t], rtol=1e-12)
            assert tok.mark == int(g['tok%d_mark' % w][t])
            assert ret is False and int(g['tok%d_ret' % w][t]) == -1        # the source returns None: not finished
        assert tok.mark == n - 2


def test_d1_counter_example(g):
    """D1, the one intentional deviation: the best state sits on the last emitting state (mark == N - 2) in every G14
    token for the last frames, and the source's own rule never reported the token finished; the D1 rule does."""
    trans = list(g['trans'])
    for w in range(int(g['tok_n'])):
        label = [int(u) for u in g['tok%d_label' % w]]
        n = 3 * len(label) + 2
        mark, ret = g['tok%d_mark' % w], g['tok%d_ret' % w]
        at_end = mark == n - 2
        assert at_end.any() and (ret == -1).all()
        b_all = all_state_emissions(g, g['tok%d_x' % w])
        tok = do.Token(0.0, 0, label, trans)                                # default rule
        d1 = [tok.viterbi(do.emission_column(label, b_all, t)) for t in range(b_all.shape[1])]
        assert d1 == list(at_end)


def test_pruning_matches_reference(g):
    """Decoder.py:159-167 on constructed lists: < 8 distinct scores, int(width * (1 - beam)) edges, ties across the cut."""
    seen_noop = seen_tie = False
    for c in range(int(g['prune_n'])):
        scores = [float(s) for s in g['prune%d_scores' % c]]
        drop = do.prune(scores)
        kept = [i for i in range(len(scores)) if i not in drop]
        assert kept == list(g['prune%d_kept' % c]), 'case %d' % c
        seen_noop |= len(kept) == len(scores) and len(scores) >= 8
        seen_tie |= len(set(scores)) < len(scores) and len(kept) < len(scores)
    assert seen_noop and seen_tie


def test_token_passing_loop_matches_reference(g):
    """Decoder.py:91-111 with the source's own finished rule on a flat tree (every token a first-character node, no
    children): per frame the ascending score list the source prints, and the surviving set after pruning."""
    states = g['tp_states']
    n = len(states)
    tree = dict(node_units=states.astype(np.int32), node_nunits=(states >= 0).sum(axis=1).astype(np.int32),
                roots=np.arange(n, dtype=np.int32), child_ptr=np.zeros(n + 1, dtype=np.int32),
                child_idx=np.zeros(0, dtype=np.int32), node_word=np.zeros(n, dtype=np.int32), words=[[] for _ in range(n)])
    b_all = all_state_emissions(g, g['tp_x'])
    log = []
    final, _ = do.decode(tree, list(g['trans']), b_all, candidate=n, finished='source', frame_log=log)
    assert len(log) == int(g['tp_frames'])
    for t, line in enumerate(log, start=1):
        assert [k for k, _ in line] == list(g['tp%d_keys' % t]), 'frame %d' % t
        np.testing.assert_allclose([s for _, s in line], g['tp%d_scores' % t], rtol=1e-12)
    assert sorted(k for k, _, _ in final) == sorted(int(k) for k in g['tp_final_keys'])
    ref = dict(zip((int(k) for k in g['tp_final_keys']), g['tp_final_scores']))
    for k, s, _ in final:
        np.testing.assert_allclose(s, ref[k], rtol=1e-12)
    assert len(final) < n                                                   # the beam did remove tokens


def test_hand_over_matches_reference(g):
    """Decoder.py:114-143: lower -> takes the donor's score, equal / higher -> keeps its own (strict >), absent -> a new
    token with the donor's score that steps at once; existing tokens keep their place in the dict."""
    donor = float(g['piw_donor_score'])
    before, after = g['piw_before'], g['piw_after']
    rules = [do.hand_over(donor, float(b)) for b in before]
    assert rules == ['take', 'keep', 'keep']
    for r, b, a in zip(rules, before, after):
        assert a == (donor if r == 'take' else b)
    assert list(g['piw_took_context'][:3]) == [1, 0, 0]
    assert do.hand_over(donor, None) == 'create' and int(g['piw_flag']) == 1
    label = [int(u) for u in g['piw_keys'][3] if u >= 0]
    b_all = all_state_emissions(g, g['piw_x'])
    new = do.Token(donor, 0, label, list(g['trans']))
    new.viterbi(do.emission_column(label, b_all, 1))                        # the frame of the hand-over
    np.testing.assert_allclose(new.p[:-1], g['piw_new_p'][:-1], rtol=1e-12)
    np.testing.assert_allclose(new.score, after[3], rtol=1e-12)
    assert list(g['piw_order']) == [0, 1, 2, 3]

"""CPU checks of the decoder's restatement (oracle/decoder_oracle.py) beyond what golden G14 pins (test_decoder_golden.py:
recursion, pruning, frame loop, in-word hand-over) -- the properties of the whole decode incl. the unpinned rules D1-D5,
so that the GPU kernel is compared against something that is itself held to account."""
import json
import os

import numpy as np
import pytest

from oracle import decoder_oracle as do
from oracle import poccala_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    from poccala_amd.Lexicon import PinYin, PronunciationLexicon
    from poccala_amd import synth
    g = json.load(open(os.path.join(HERE, 'golden', 'G13_lexicon.json')))
    path = str(tmp_path_factory.mktemp('lex') / 'Mandarin.dat')
    with open(path, 'w') as f:
        for k, v in g['table'].items():
            f.write('%s\t%s\n' % (k, v))
    py = PinYin(path)
    lx = PronunciationLexicon()
    lx.generate_lexicon(words=g['words'][:60], pinyin=py)
    units = sorted({u for w in g['words'][:60] for r in py.word2pinyin(w) for x in r for u in x.split(',')})
    tree = lx.compile({u: i for i, u in enumerate(units)})
    trans = [synth.flat_start_transmat() for _ in units]
    rng = np.random.default_rng(3)
    b_all = rng.standard_normal((len(units) * 3, 50)) * 4 - 50
    return tree, trans, b_all


def test_token_step_is_the_max_recursion(setup):
    """Token.viterbi against the shared A16 restatement (poccala_oracle.token_viterbi_step, Decoder.py:270-285)."""
    tree, trans, b_all = setup
    node = int(tree['roots'][0])
    units = [int(u) for u in tree['node_units'][node][:tree['node_nunits'][node]]]
    tok = do.Token(0.0, node, units, trans)
    la, lpi = do.sentence_hmm(units, trans)
    p, total = lpi, 0.0
    for t in range(6):
        col = do.emission_column(units, b_all, t)
        tok.viterbi(col)
        p, point = po.token_viterbi_step(p, la, col, first=(t == 0))
        total += point
        np.testing.assert_array_equal(tok.p, p)
        assert tok.score == total and tok.mark == int(np.argmax(p))


def test_beam_and_history(setup):
    tree, trans, b_all = setup
    tr_free, tr_beam = [], []
    free, h_free = do.decode(tree, trans, b_all, beam=1.0, candidate=5, trace=tr_free)
    fin, hist = do.decode(tree, trans, b_all, beam=0.85, candidate=5, trace=tr_beam)
    assert tr_free[0] == tr_beam[0] == len(tree['roots'])
    assert all(a >= b for a, b in zip(tr_free, tr_beam)) and min(tr_beam) > 0          # the beam only removes; D4 keeps it alive
    assert free[0][1] >= fin[0][1]
    assert [s for _, s, _ in fin] == sorted((s for _, s, _ in fin), reverse=True)
    for prev, node in hist:
        assert prev < len(hist) and tree['node_word'][node]
    words = do.words_of(fin[0], hist, tree)
    assert all(w for w in words)
    # deterministic
    again, _ = do.decode(tree, trans, b_all, beam=0.85, candidate=5)
    assert again == fin
    # capacity: hand-overs beyond the frame's slots are dropped and reported
    info = {}
    small, _ = do.decode(tree, trans, b_all, beam=0.85, candidate=5, max_tokens=len(tree['roots']) + 3, info=info)
    assert info.get('overflow') and small[0][1] <= free[0][1]

"""The lexicon half of SURVEY section 8(f) rank 3 against what the reference's own classes produced (golden G13, made by
tests/golden/make_golden_lexicon.py from Lexicon/PinYin.py:58-132 and Lexicon/PronunciationLexicon.py:45-94)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def g13(tmp_path_factory):
    g = json.load(open(os.path.join(HERE, 'golden', 'G13_lexicon.json')))
    d = tmp_path_factory.mktemp('lex')
    path = str(d / 'Mandarin.dat')
    with open(path, 'w') as f:                       # the reference's table format: HEX<TAB>PIN1 PIN2 ...
        for k, v in g['table'].items():
            f.write('%s\t%s\n' % (k, v))
    g['table_path'] = path
    return g


def test_word2pinyin_matches_reference_for_every_flag_combination(g13):
    from poccala_amd.Lexicon import PinYin
    py = PinYin(g13['table_path'])
    n = 0
    for s, row in zip(g13['strings'], g13['word2pinyin']):
        for fl, ref in zip(g13['flags'], row):
            got = py.word2pinyin(s, *fl)
            if got is not None and not fl[3]:
                got = [sorted(x) for x in got]       # the reference builds these through a set
            assert got == ref, (s, fl)
            n += 1
    assert n == len(g13['strings']) * 16


def test_pronunciation_tree_matches_reference(g13, tmp_path):
    from poccala_amd.Lexicon import PinYin, PronunciationLexicon
    py = PinYin(g13['table_path'])
    lex = PronunciationLexicon()
    tree = lex.generate_lexicon(words=g13['words'], pinyin=py)
    assert tree == g13['tree']                       # nested dicts: same keys, same word lists in the same order
    # insertion order of the children is what the decoder's expansion order (and so its tie-breaks) follows
    def order(a, b):
        assert list(a.keys()) == list(b.keys())
        for k in a:
            if k != 'word':
                order(a[k], b[k])
    order(tree, g13['tree'])
    # from a directory of word lists, as the reference reads them, and through the pickle round trip
    os.makedirs(tmp_path / 'data')
    with open(tmp_path / 'data' / 'w.txt', 'w') as f:
        f.write('\n'.join(g13['words']) + '\n')
    lex2 = PronunciationLexicon()
    lex2.generate_lexicon(path=str(tmp_path / 'data') + '/', savepath=str(tmp_path / 'lex.pkl'), pinyin=py)
    lex3 = PronunciationLexicon()
    lex3.init_lexicon(str(tmp_path / 'lex.pkl'))
    assert lex2.lexicon == g13['tree'] and lex3.lexicon == g13['tree'] and lex2.size == len(g13['words'])


def test_compiled_tree_is_consistent(g13):
    from poccala_amd.Lexicon import PinYin, PronunciationLexicon
    py = PinYin(g13['table_path'])
    lex = PronunciationLexicon()
    lex.generate_lexicon(words=g13['words'], pinyin=py)
    units = sorted({u for s in g13['strings'] for r in (py.word2pinyin(s) or []) for x in r for u in x.split(',')} |
                   {u for w in g13['words'] for r in py.word2pinyin(w) for x in r for u in x.split(',')})
    index = {u: i for i, u in enumerate(units)}
    c = lex.compile(index)
    n = len(c['names'])
    assert c['dropped'] == [] and n == len(c['node_parent']) == len(c['node_word'])
    # every word is reachable: walk the flat tree along each reading combination
    first = {}
    for r in c['roots']:
        first.setdefault(c['names'][r], r)
    for w in g13['words']:
        p = py.word2pinyin(w)
        node = first[p[0][0]]
        for level in p[1:]:
            kids = c['child_idx'][c['child_ptr'][node]:c['child_ptr'][node + 1]]
            node = [k for k in kids if c['names'][k] == level[0]][0]
            assert c['node_parent'][node] >= 0
        assert w in c['words'][node] and c['node_word'][node] == 1
    for i in range(n):
        lab = c['names'][i].split(',')
        assert c['node_nunits'][i] == len(lab) and [units[u] for u in c['node_units'][i][:len(lab)]] == lab
    # a unit outside the inventory removes the branch, not the tree
    small = dict(index)
    gone = units[len(units) // 2]
    del small[gone]
    c2 = lex.compile(small)
    assert c2['dropped'] and all(gone not in nm.split(',') for nm in c2['names'])

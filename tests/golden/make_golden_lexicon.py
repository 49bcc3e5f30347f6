#!/usr/bin/env python3
"""Golden vectors for the lexicon half of SURVEY section 8(f) rank 3, made by RUNNING the reference classes
Lexicon.PinYin.PinYin (word2pinyin, Lexicon/PinYin.py:58-132) and Lexicon.PronunciationLexicon.PronunciationLexicon
(generate_lexicon, PronunciationLexicon.py:45-94) on the reference's Mandarin.dat.  Build container only.
Writes tests/golden/G13_lexicon.json: the input strings / word list, the table lines those characters need (so the test
does not depend on the reference tree at run time), and the reference's outputs.  No source text is stored.

    python tests/golden/make_golden_lexicon.py
"""
import itertools
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

WORDS = ['中国', '语音', '识别', '语音识别', '声学', '模型', '你好', '我们', '月亮', '女儿', '军队', '学习', '安全', '长大', '成长',
         '重要', '重复', '银行', '行走', '音乐', '快乐', '北京', '上海', '广州', '重庆', '好的', '的确', '目的', '了解', '完了',
         '一', '一个', '不', '不要', '儿子', '而且', '啊', '嗯', '欧洲', '爱', '爱好', '恩情', '昂贵', '哦', '饿', '二',
         '云', '雨', '鱼', '远', '原来', '语言', '元', '王', '万', '为什么', '文化', '我', '五', '外面', '温暖',
         '家', '角色', '觉得', '睡觉', '参加', '参差', '人参', '乐', '着', '着急', '看着', '传说', '传记', '朝阳', '朝代', '出差', '差不多']


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, REF + '/Lexicon')          # PinYin.py reads sys.path[0] + '/Mandarin.dat' at import
    from Lexicon.PinYin import PinYin
    from Lexicon.PronunciationLexicon import PronunciationLexicon
    table_path = REF + '/Lexicon/Mandarin.dat'
    py = PinYin(table_path)
    # random strings over the whole table on top of the hand-picked words
    rng = np.random.default_rng(13)
    keys = [line.split('\t')[0] for line in open(table_path)]
    chars = [chr(int(k, 16)) for k in keys]
    pick = [chars[i] for i in rng.choice(len(chars), 400, replace=False)]
    strings = WORDS + [''.join(pick[i:i + n]) for i, n in zip(range(0, 390, 3), itertools.cycle([1, 2, 3]))] + ['abc', '中a']
    flags = list(itertools.product([True, False], repeat=4))          # separate, check_tone, extend, show_tone_mark
    w2p = []
    for s in strings:
        row = []
        for fl in flags:
            r = py.word2pinyin(s, *fl)
            if r is not None and not fl[3]:
                r = [sorted(x) for x in r]                             # built through a set: order unspecified
            row.append(r)
        w2p.append(row)
    # the tree
    words = WORDS + [''.join(pick[i:i + n]) for i, n in zip(range(0, 240, 3), itertools.cycle([2, 3, 2]))]
    tmp = tempfile.mkdtemp(prefix='pcl_lex_')
    os.makedirs(tmp + '/data')
    with open(tmp + '/data/words.txt', 'w') as f:
        f.write('\n'.join(words) + '\n')
    lex = PronunciationLexicon()
    lex.generate_lexicon(path=tmp + '/data/', savepath=tmp + '/lex.pkl')
    # only the table lines the fixture's characters need travel with it
    need = set(''.join(strings + words))
    table = {}
    for line in open(table_path):
        k, v = line.strip('\n').split('\t')
        try:
            if chr(int(k, 16)) in need:
                table[k] = v
        except ValueError:
            pass
    out = dict(strings=strings, flags=[list(f) for f in flags], word2pinyin=w2p, words=words, tree=lex.lexicon, table=table)
    with open(os.path.join(HERE, 'G13_lexicon.json'), 'w') as f:
        json.dump(out, f, ensure_ascii=False, separators=(',', ':'))
    print('wrote G13_lexicon.json:', len(strings), 'strings,', len(words), 'words,', len(table), 'table lines,', os.path.getsize(os.path.join(HERE, 'G13_lexicon.json')), 'bytes')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Randomised soak of the token-passing kernel against its CPU restatement (bit-exact): random sub-lexicons of the golden
fixture (different trees), models, utterance lengths, beams, distinct-score thresholds, candidate counts and token caps
(down to a handful, so that overflow and near-empty frames occur).  usage: decode_fuzz.py [cases] [seed]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from poccala_amd import Engine, PCL_F64, synth
from poccala_amd.Lexicon import PinYin, PronunciationLexicon
from oracle import decoder_oracle as do

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'G13_lexicon.json')))
with tempfile.NamedTemporaryFile('w', suffix='.dat', delete=False) as f:
    for k, v in g['table'].items():
        f.write('%s\t%s\n' % (k, v))
py = PinYin(f.name)
os.unlink(f.name)
eng = Engine(0)
bad = 0
for case in range(cases):
    nw = int(rng.integers(3, len(g['words']) + 1))
    words = list(rng.choice(g['words'], size=nw, replace=False))
    if rng.random() < 0.5:                                     # extra random strings: deeper / wider trees
        chars = sorted({ch for w in g['words'] for ch in w})
        words += [''.join(rng.choice(chars, size=rng.integers(1, 5))) for _ in range(int(rng.integers(1, 300)))]
    lx = PronunciationLexicon()
    lx.generate_lexicon(words=words, pinyin=py)
    units = sorted({u for w in words for r in (py.word2pinyin(w) or []) for x in r for u in x.split(',')})
    tree = lx.compile({u: i for i, u in enumerate(units)})
    M, D = int(rng.integers(1, 5)), int(rng.choice([13, 26, 39]))
    mean, var, w, trans = synth.make_model(len(units), M, D, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.5:                                     # dense unit matrices (skips, back transitions)
        trans = []
        for _ in units:
            a = np.zeros((5, 5)); a[0, 1:3] = [0.7, 0.3]; a[1:-1, 1:] = rng.dirichlet(np.ones(4), size=3); trans.append(a)
    elif rng.random() < 0.6:                                   # left-to-right units with random self-loop weights (the lane-per-token kernel)
        trans = []
        for _ in units:
            a = np.zeros((5, 5)); a[0, 1] = 1.0
            for r_ in (1, 2, 3):
                x_ = rng.uniform(0.05, 0.95); a[r_, r_] = x_; a[r_, r_ + 1] = 1.0 - x_
            trans.append(a)
    trans = np.stack(trans)
    U = int(rng.integers(1, 5))
    lens = rng.integers(1, 70, size=U).astype(np.int32)
    frames = (rng.standard_normal((int(lens.sum()), D)) * rng.uniform(0.5, 2.0)).astype(np.float32)
    begin = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    beam = float(rng.choice([0.5, 0.7, 0.85, 0.95, 1.0]))
    md = int(rng.choice([1, 2, 8, 20]))
    cand = int(rng.integers(1, 8))
    cap = int(rng.choice([3, 17, 64, 257, 1024, 1500, 4096]))
    eng.load_model(mean, var, w); eng.load_units(trans); eng.load_lexicon(tree); eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    b.score(PCL_F64)
    B = b.get('B')
    got = b.decode(beam=beam, min_distinct=md, candidate=cand, max_tokens=cap)
    b.close()
    for u in range(U):
        trace, info = [], {}
        fin, hist = do.decode(tree, list(trans), B[u][1:-1], beam=beam, candidate=cand, min_distinct=md, max_tokens=cap, trace=trace, info=info)
        gu = got[u]
        ok = (np.array_equal(gu['n_tokens'], np.array(trace)) and gu['history'] == [(int(p), int(n)) for p, n in hist]
              and [(n, h) for n, _, h in gu['final']] == [(n, h) for n, _, h in fin] and [s for _, s, _ in gu['final']] == [float(s) for _, s, _ in fin]
              and gu['overflow'] == bool(info.get('overflow')))
        if not ok:
            bad += 1
            print('MISMATCH case %d utt %d: words %d nodes %d roots %d T %d beam %g md %d cand %d cap %d' % (case, u, lx.size, len(tree['names']), len(tree['roots']), lens[u], beam, md, cand, cap))
            print('   tokens', gu['n_tokens'][:10].tolist(), trace[:10])
print('%d cases, %d mismatching utterances' % (cases, bad))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Golden G14: what CAN be executed of the reference's token-passing decoder (Decoder.py), by RUNNING it.

Build container only (imports /root/reference).  Decoder.py fails to import at `from LanguageModel.Ngram import Ngram`
(Decoder.py:17); with an empty stand-in module in sys.modules (the same device make_golden.py uses for `pyaudio`) the
module loads, and these parts run against the live classes:

  Token.__init__ / Token.viterbi (Decoder.py:221-288)   through a duck-typed `am` whose `embedded(label, data_index, alter)`
        answers the two calls Token makes (alter 29 -> [states, observation, A, pi], alter 2 -> [B]) with what the
        reference's own AcousticModel.embedded builds from the reference's own LHMM / GMM objects;
  pruning (Decoder.py:159-167)                          a free function over the module global `tokens`;
  token_passing (Decoder.py:91-111)                     the frame loop, its per-frame score list captured through the
        module-level `print`;
  passing_in_word (Decoder.py:114-143)                  the hand-over of a score to the children of a tree node.

Nothing of the reference's text is stored: the fixture holds the seeded inputs and the numbers the reference produced.

    python tests/golden/make_golden_decoder.py          # rewrites tests/golden/G14_decoder.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, RecLog, diag_cov, rand_gmm   # noqa: E402

S = 5


def main():
    scratch, util, LHMM, Clustering, AcousticModel = import_reference()
    ngram_pkg = types.ModuleType('LanguageModel')
    ngram_mod = types.ModuleType('LanguageModel.Ngram')
    ngram_mod.Ngram = type('Ngram', (), {})
    ngram_pkg.Ngram = ngram_mod
    sys.modules['LanguageModel'] = ngram_pkg
    sys.modules['LanguageModel.Ngram'] = ngram_mod
    import Decoder                                                          # the reference's module
    GMM = Clustering.GMM
    real_am = AcousticModel(RecLog(), 'XIF_tone', processes=1, console=False, state_num=S)

    rng = np.random.default_rng(1401)
    n_units, m, d, T = 5, 3, 13, 40
    names = ['u%d' % i for i in range(n_units)]
    params, trans, unit = [], [], {}
    for u, name in enumerate(names):
        tr = np.zeros((S, S))
        tr[0][1] = 1.
        for j in range(1, S - 1):
            stay = rng.uniform(0.3, 0.8)                                    # trained-looking rows, not the flat start
            tr[j][j], tr[j][j + 1] = stay, 1. - stay
        gmms, par = [], []
        for k in range(S - 2):
            mean, var, w = rand_gmm(rng, m, d)
            mean = mean * 2.0 + 3.0 * u + k                                 # states far enough apart for the argmax to move
            par.append((mean, var, w))
            gmms.append(GMM(RecLog(), dimension=d, mix_level=m, alpha=w.copy(), mean=mean.copy(),
                            covariance=diag_cov(var), gmm_id=k))
        prof = [AcousticModel.VirtualState(1.)] + gmms + [AcousticModel.VirtualState(0.)]
        hmm = LHMM({i: name for i in range(S)}, S, RecLog(), transmat=tr.copy(), profunc=prof, fix_code=0)
        unit[name] = [None, hmm]
        params.append(par)
        trans.append(tr)

    class DuckAM(object):
        """The two members of the old AcousticModel API Token touches (Decoder.py:232,245-247): `unit` and
        `embedded(label, data_index, alter)` over five parts [states, observation, A, B, pi]."""

        def __init__(self):
            self.unit = unit

        def embedded(self, label, data_index, alter):
            hmms = [unit[x][1] for x in label]
            out = []
            if alter & 16:
                out.append(real_am.embedded(label, hmms, data_index, alter=8)[0])
            if alter & 8:
                out.append(None)                                            # `complex_observation`: never read by Token
            if alter & 4:
                out.append(real_am.embedded(label, hmms, data_index, alter=4)[0])
            if alter & 2:
                out.append(real_am.embedded(label, hmms, data_index, alter=2)[0])
            if alter & 1:
                out.append(real_am.embedded(label, hmms, data_index, alter=1)[0])
            return out

    am = DuckAM()

    def sample_path(label, t_total):
        """frames that walk the label's emitting states in order, so the best state reaches the last emitting one."""
        rows = []
        states = [(names.index(x), k) for x in label for k in range(S - 2)]
        per = t_total // len(states)
        for i, (u, k) in enumerate(states):
            n = per if i < len(states) - 1 else t_total - per * (len(states) - 1)
            mean, var, w = params[u][k]
            comp = rng.choice(m, size=n, p=w)
            rows.append(mean[comp] + rng.standard_normal((n, d)) * np.sqrt(var[comp]))
        return np.concatenate(rows)

    out = dict(n_units=n_units, M=m, D=d, S=S,
               mean=np.array([[p[0] for p in par] for par in params]),      # (units, 3, M, D)
               var=np.array([[p[1] for p in par] for par in params]),
               weight=np.array([[p[2] for p in par] for par in params]),
               trans=np.array(trans))

    # ---------------------------------------------------------------- G14a  Token.viterbi, frame by frame
    words = [['u0'], ['u1', 'u2'], ['u3', 'u4'], ['u2', 'u2']]
    out['tok_n'] = len(words)
    for wi, label in enumerate(words):
        x = sample_path(label, T)
        unit_time = {k: -1 for k in names}
        tok = Decoder.Token(-1.5 * wi, ','.join(label), [], [], am, {})
        n_states = (S - 2) * len(label) + 2
        p_hist, score, mark, ret, bcol = [], [], [], [], []
        for t in range(T):
            r = tok.viterbi([x[t:t + 1]], t, unit_time)
            p_hist.append(np.array(tok._Token__p_list, dtype=np.float64))
            bcol.append(np.array(tok._Token__embedded_list[3][:, 0], dtype=np.float64))
            score.append(float(tok.score))
            mark.append(int(tok.mark))
            ret.append(-1 if r is None else int(bool(r)))
        mark = np.array(mark)
        assert (mark == n_states - 2).any(), 'the walk should reach the last emitting state'
        assert all(r == -1 for r in ret), 'under the live API the source rule never fires'
        out['tok%d_label' % wi] = np.array([names.index(u) for u in label])
        out['tok%d_score0' % wi] = -1.5 * wi
        out['tok%d_x' % wi] = x
        out['tok%d_p' % wi] = np.array(p_hist)
        out['tok%d_bcol' % wi] = np.array(bcol)
        out['tok%d_score' % wi] = np.array(score)
        out['tok%d_mark' % wi] = mark
        out['tok%d_ret' % wi] = np.array(ret)                               # -1 = the source returned None

    # ---------------------------------------------------------------- G14b  pruning on constructed score lists
    class Dummy(object):
        def __init__(self, s):
            self.score = s

    cases = []
    base = rng.standard_normal(200) * 10 - 300
    for width in (5, 7, 8, 9, 13, 14, 20, 27, 40, 60, 100, 133):
        cases.append(base[:width].copy())
    cases.append(np.repeat(base[:7], 3))                                    # 21 tokens, 7 distinct scores: nothing is pruned
    cases.append(np.concatenate([np.repeat(base[:7], 3), base[7:8]]))        # 22 tokens, 8 distinct: int(22 * 0.15..) = 3
    tie = base[:20].copy()
    srt = np.argsort(tie, kind='stable')
    tie[srt[1]] = tie[srt[2]] = tie[srt[3]] = tie[srt[4]]                    # a four-way tie across the cut (3 of 20 go)
    cases.append(tie)
    tie2 = base[:40].copy()
    tie2[::2] = tie2[0]                                                      # half the tokens share one score
    cases.append(tie2)
    cases.append(np.full(30, -12.5))                                         # one distinct score
    out['prune_n'] = len(cases)
    for ci, sc in enumerate(cases):
        Decoder.tokens.clear()
        for i, s in enumerate(sc):
            Decoder.tokens['k%d' % i] = Dummy(float(s))
        score_list = [['k%d' % i, float(s)] for i, s in enumerate(sc)]
        score_list.sort(key=lambda q: q[1])                                  # as token_passing does before the call (:108)
        Decoder.pruning(score_list)
        out['prune%d_scores' % ci] = np.array(sc)
        out['prune%d_kept' % ci] = np.array(sorted(int(k[1:]) for k in Decoder.tokens.keys()))
    Decoder.tokens.clear()

    # ---------------------------------------------------------------- G14c  token_passing: the frame loop + pruning
    states = [['u0'], ['u1'], ['u2'], ['u3'], ['u4'], ['u0', 'u1'], ['u1', 'u2'], ['u2', 'u3'], ['u3', 'u4'],
              ['u4', 'u0'], ['u0', 'u2'], ['u1', 'u3'], ['u2', 'u4'], ['u3', 'u0']]
    Tp = 24
    x = np.concatenate([sample_path(['u1', 'u2'], Tp // 2), sample_path(['u3'], Tp - Tp // 2)])
    unit_time = {k: -1 for k in names}
    Decoder.unit_time = unit_time
    Decoder.mfcc = x
    Decoder.t = 0
    Decoder.tokens.clear()
    for label in states:                                                     # generate_first_word's tail (:86-88), score 0
        key = ','.join(label)
        Decoder.tokens[key] = Decoder.Token(0.0, key, [], [], am, {})
        Decoder.tokens[key].viterbi([x[0:1]], 0, unit_time)
    lines = []
    Decoder.print = lambda *a: lines.append(a)
    Decoder.token_passing()
    del Decoder.print
    key_id = {','.join(label): i for i, label in enumerate(states)}
    out['tp_x'] = x
    out['tp_states'] = np.array([[names.index(l[0]), names.index(l[1]) if len(l) > 1 else -1] for l in states])
    out['tp_frames'] = len(lines)
    for (t, n, sl) in lines:                                                 # per frame: the ascending [key, score] list
        out['tp%d_keys' % t] = np.array([key_id[k] for k, _ in sl])
        out['tp%d_scores' % t] = np.array([s for _, s in sl])
    out['tp_final_keys'] = np.array([key_id[k] for k in Decoder.tokens.keys()])
    out['tp_final_scores'] = np.array([tk.score for tk in Decoder.tokens.values()])
    assert len(Decoder.tokens) < len(states), 'pruning should have removed tokens'

    # ---------------------------------------------------------------- G14d  passing_in_word: hand-over to the children
    Decoder.tokens.clear()
    Decoder.t = 7
    frame = [x[7:8]]
    unit_time = {k: -1 for k in names}
    Decoder.unit_time = unit_time
    lex = {'u0,u1': {'u2': {}, 'u3': {}, 'u4': {}, 'u1,u2': {}, 'word': ['w']}}
    donor = Decoder.Token(-40.0, 'u0,u1', ['ctx'], ['stk'], am, lex)
    for key, s in (('u2', -55.0), ('u3', -40.0), ('u4', -20.0)):             # lower / equal / higher than the donor
        tk = Decoder.Token(s, key, [], [], am, {})
        tk.viterbi([x[6:7]], 6, unit_time)                                   # they have a recursion state of their own
        tk.score = s                                                         # exactly lower / equal / higher
        Decoder.tokens[key] = tk
    before = {k: (tk.score, np.array(tk._Token__p_list)) for k, tk in Decoder.tokens.items()}
    Decoder.print = lambda *a: None
    flag = Decoder.passing_in_word(donor, frame)
    del Decoder.print
    out['piw_flag'] = int(bool(flag))
    out['piw_x'] = x[6:8]
    out['piw_donor_score'] = -40.0
    out['piw_keys'] = np.array([[2, -1], [3, -1], [4, -1], [1, 2]])
    out['piw_before'] = np.array([before['u2'][0], before['u3'][0], before['u4'][0]])
    out['piw_after'] = np.array([Decoder.tokens[k].score for k in ('u2', 'u3', 'u4', 'u1,u2')])
    out['piw_order'] = np.array([['u2', 'u3', 'u4', 'u1,u2'].index(k) for k in Decoder.tokens.keys()])
    for k in ('u2', 'u3', 'u4'):                                             # the receiving token keeps its own recursion state
        assert np.array_equal(before[k][1], Decoder.tokens[k]._Token__p_list)
    out['piw_took_context'] = np.array([int(Decoder.tokens[k].context == ['ctx']) for k in ('u2', 'u3', 'u4', 'u1,u2')])
    out['piw_new_p'] = np.array(Decoder.tokens['u1,u2']._Token__p_list)     # the created token took its first step at once
    Decoder.tokens.clear()

    path = os.path.join(HERE, 'G14_decoder.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()

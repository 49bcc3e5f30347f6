"""CPU restatement of the reference's token-passing decoder -- TEST INFRASTRUCTURE.
PARITY: recursion + pruning + frame loop + in-word hand-over PINNED (golden G14); the rules D1-D5 below UNPINNED.

Only tests/, __graft_entry__.smoke() and bench tools may import this; the product (poccala_amd/) never does.

The reference's Decoder.py is dead code as a program: it imports a module that does not exist (`LanguageModel.Ngram`,
Decoder.py:17) and its `main` / `generate_first_word` call AcousticModel / LHMM methods of an older API
(`am.initialize_unit`, `hmm.change_T`, `hmm.q_function`, Decoder.py:73-75,195-197).  With a stand-in for the missing
module its pieces DO run (tests/golden/make_golden_decoder.py, build container only), and golden G14 holds what the
reference itself produced for them; tests/test_decoder_golden.py holds this file to G14:

  pinned     Token.viterbi (Decoder.py:250-288) -> `Token.viterbi`: p, score, mark per frame for one- and two-unit
             tokens over 40 frames (rtol 1e-12; marks exact), driven through the reference's own AcousticModel.embedded;
  pinned     pruning (Decoder.py:159-167) -> `prune`: 17 constructed score lists (fewer than 8 distinct scores, the
             int(width * (1 - beam)) edges, ties across the cut, one shared score): surviving sets exact;
  pinned     token_passing (Decoder.py:91-111) -> `decode(finished='source')` on a flat tree: the per-frame ascending
             score lists and the surviving token sets of 14 tokens over 24 frames, exact keys / rtol 1e-12 scores;
  pinned     passing_in_word (Decoder.py:114-143) -> `hand_over`: an existing token takes the donor's score only when it
             is STRICTLY better and keeps its own recursion state; a missing one is created with the donor's score and
             takes its first step at once; the return flag = "words end here";
  D1         is the one INTENTIONAL deviation from executable source, with its golden counter-example: in G14 the best
             state reaches the last emitting state of every token and the source's rule (`mark == len(states) - 1`)
             never fires (tok*_ret are all None), so no token would ever be handed over.
  unpinned   D2-D5 (what the source cannot do at all: node keys, first-word seeding, word-to-word hand-over, the order of
             steps and hand-overs inside a frame) and the capacity rule.

What is restated, line by line, and where the source cannot run, the smallest well-defined rule that fills the gap:

  restated   Token.viterbi (Decoder.py:250-288): first frame p = ln pi + B[:,0]; later p_j = max_i(p_i + ln A_ij) + B_j;
             score += max_j p_j; `mark` = first argmax.  The token's HMM is AcousticModel.embedded of the node's units
             (Token.__init__ :232, embedded :957-1014): uniform pi, entry row ln 1 = 0, exit row ln 0 = -inf.
  restated   token_passing (:91-111): frame by frame, tokens in insertion order; a finished token hands its SCORE to
             new tokens for the children of its tree node (passing_in_word :114-143), each of which runs its first
             frame at once, and is deleted; tokens created in a frame take no part in that frame's pruning.
  restated   pruning (:159-167): nothing happens while fewer than 8 distinct scores exist; otherwise the
             int(width * (1 - beam)) lowest-scoring tokens (stable ascending sort) are deleted, beam = 0.85 (:34).
  restated   transfer (:175-187): the `candidate` best tokens at the end, and the words of those whose node ends words.
  D1 (gap)   "finished" is `mark == len(states) - 1` in the source (:276,:287); under the live API the exit state scores
             -inf (AcousticModel.py:219) and can never be the argmax, so the test is made on the last EMITTING state:
             finished <=> mark >= N - 2.
  D2 (gap)   tokens are keyed by tree NODE.  The source keys its dict by the pinyin string of the node (`tokens[state]`,
             :88,:142), which merges different paths that end in the same syllable and then keeps only the better score
             (:126-134) while continuing the other token's recursion -- and is where its one-token-per-key bookkeeping
             breaks.  With node keys a node receives at most one token in an utterance (a tree node has one parent).
  D3 (gap)   generate_first_word (:63-88) ranks first syllables with `hmm.q_function()` on 20 frames -- a method that no
             longer exists.  Here every first-character node starts with a token of score 0 at frame 0.
  D4 (gap)   passing_between_word (:146-156) is a stub (it needs the missing n-gram model and ends in `Token()`), so in
             the source a token that finishes a word simply disappears and the token set dies out after a few frames.
             Here a finished token whose node ends words hands its score to EVERY first-character node (a uniform
             language model) and records the word in a history chain (the source's empty `context` / `stack`).
  D5 (gap)   the source walks its token dict sequentially, so whether a hand-off meets the receiving token before or
             after that token's own step of the frame depends on dict order.  Here a frame is: (1) every live token
             takes its step; (2) every finished token hands its (after-step) score to its targets; a target without a
             live unfinished token gets a new token that takes its first step at once (:135-140), a target with one
             keeps its recursion and takes the score if it is STRICTLY better (:126-134), several donors -> the best,
             the earliest on ties; (3) finished tokens are deleted; (4) pruning over the tokens that were alive before
             the frame (new ones are exempt, as in the source where they are not in score_list).
  capacity   (not in the source) the device holds at most max_tokens live tokens per utterance: a creation that would exceed
             it is dropped and reported (info['overflow']); at frame 0 only the first max_tokens first-character nodes start.
"""
import numpy as np

NEG_INF = -np.inf


def sentence_hmm(units, unit_trans, s=5):
    """ln A (N,N) and ln pi (N,) of AcousticModel.embedded for a node's units (AcousticModel.py:957-1014)."""
    e = s - 2
    n = e * len(units) + 2
    a = np.zeros((n, n))
    a[:s - 1, :s] = unit_trans[units[0]][:-1]
    for i, u in enumerate(units):
        lo = i * e + 1
        a[lo:lo + e, lo - 1:lo - 1 + s] = unit_trans[u][1:-1]
    with np.errstate(divide='ignore'):
        return np.log(a), np.log(np.ones(n) / n)


def emission_column(units, b_all, t, s=5):
    """B[:, t] of the node's embedded HMM: entry 0, the units' emitting states, exit -inf."""
    e = s - 2
    rows = [0.0]
    for u in units:
        rows.extend(b_all[u * e + k, t] for k in range(e))
    rows.append(NEG_INF)
    return np.array(rows)


class Token(object):
    finished_rule = 'D1'

    def __init__(self, score, node, units, unit_trans, s=5):
        self.score = score
        self.node = node
        self.units = units
        self.la, self.lpi = sentence_hmm(units, unit_trans, s)
        self.p = None
        self.mark = -1

    def viterbi(self, bcol):
        """Decoder.py:250-288.  Returns True when the best state is the last emitting one (D1)."""
        n = len(self.lpi)
        if self.p is None:
            self.p = self.lpi + bcol                                           # :270
        else:
            new = np.empty(n)
            for j in range(n):                                                 # :278-282
                new[j] = (self.p + self.la[:, j]).max()
            self.p = new + bcol                                                # :283
        point = self.p.max()                                                   # info(), :263-268
        self.mark = int(np.where(self.p == point)[0][0])
        self.score += point                                                    # :285
        if self.finished_rule == 'source':                                     # :276,:287 -- never true under the live API
            return self.mark == n - 1
        return self.mark >= n - 2                                              # D1


def prune(scores, beam=0.85, min_distinct=8):
    """Decoder.pruning (:159-167) over the scores of the tokens alive before the frame, in token order.
    Returns the set of token positions that are deleted: none while fewer than `min_distinct` distinct scores exist,
    otherwise the int(width * (1 - beam)) first of the stable ascending sort (:108)."""
    if len(set(scores)) < min_distinct:
        return set()
    ranked = sorted(range(len(scores)), key=lambda i: scores[i])
    return set(ranked[:int(len(scores) * (1 - beam))])


def hand_over(score, live_score):
    """passing_in_word's rule for ONE target (:123-140).  `live_score` is the target's live token score or None.
    Returns 'create' (no live token: a new one starts from `score` and steps at once), 'take' (the live token takes the
    score, its recursion state stays) or 'keep'."""
    if live_score is None:
        return 'create'
    return 'take' if score > live_score else 'keep'


def decode(tree, unit_trans, b_all, beam=0.85, candidate=5, min_distinct=8, s=5, max_tokens=None, trace=None, info=None,
           finished='D1', frame_log=None):
    """`finished`: 'D1' (default) or 'source' = the source's own test, under which no token ever finishes and the loop is
    exactly token_passing (:91-111) -- the form golden G14 pins.  `frame_log`, if a list, receives per frame the ascending
    [(node, score)] list of the tokens that stepped and did not finish (what the source prints at :111).
    tree: the dict PronunciationLexicon.compile returns; b_all (J,T): ln b_j(o_t) of every GMM state.
    Returns (final, history): final = [(node, score, hist)] of the `candidate` best tokens after the last frame
    (descending, ties in token order); history = [(previous entry or -1, word-end node)], the chain `hist` points into."""
    T = b_all.shape[1]
    units_of = [[int(u) for u in row[:n]] for row, n in zip(tree['node_units'], tree['node_nunits'])]
    roots = [int(r) for r in tree['roots']]
    kids = lambda n: [int(c) for c in tree['child_idx'][tree['child_ptr'][n]:tree['child_ptr'][n + 1]]]
    tokens = []                                                                # live tokens in creation order
    history = []

    def Token_(*a):
        tok = Token(*a)
        tok.finished_rule = finished
        return tok
    for r in roots:                                                            # D3
        if max_tokens is not None and len(tokens) >= max_tokens:               # device capacity: the first max_tokens first-character
            if info is not None:                                               # nodes start, the rest is reported as overflow
                info['overflow'] = True
            break
        tok = Token_(0.0, r, units_of[r], unit_trans, s)
        tok.hist = -1
        tok.viterbi(emission_column(tok.units, b_all, 0, s))
        tokens.append(tok)
    if trace is not None:
        trace.append(len(tokens))
    for t in range(1, T):
        n_start = len(tokens)
        done = [tok.viterbi(emission_column(tok.units, b_all, t, s)) for tok in tokens]          # (1) every token steps
        live = {tok.node: tok for tok, d in zip(tokens, done) if not d}
        offers = {}                                                            # target node -> (score, hist), first best donor
        order = []
        for tok, d in zip(tokens, done):                                       # (2) hand-offs, donors in token order
            if not d:
                continue
            targets = [(c, tok.hist) for c in kids(tok.node)]                  # passing_in_word, :114-143
            if tree['node_word'][tok.node]:                                    # D4: every first-character node; the history entry
                targets += [(r, ('word', tok)) for r in roots]                 # is made once, for the donor that wins (below)
            for node, hist in targets:
                if node not in offers:
                    offers[node] = (tok.score, hist)
                    order.append(node)
                elif tok.score > offers[node][0]:
                    offers[node] = (tok.score, hist)
        winner = None                                                          # all roots receive the same best word-end donor
        for node in order:
            h = offers[node][1]
            if isinstance(h, tuple):
                if winner is None:
                    history.append((h[1].hist, h[1].node))
                    winner = len(history) - 1
                offers[node] = (offers[node][0], winner)
        created = []
        for node in order:
            score, hist = offers[node]
            rule = hand_over(score, live[node].score if node in live else None)
            if rule == 'take':                                                 # :126-134 (the recursion state is kept)
                live[node].score = score
                live[node].hist = hist
            elif rule == 'keep':
                pass
            elif max_tokens is not None and n_start + len(created) >= max_tokens:
                if info is not None:
                    info['overflow'] = True                                    # device capacity: the frame's token slots are used up
            else:
                new = Token_(score, node, units_of[node], unit_trans, s)
                new.hist = hist
                new.viterbi(emission_column(new.units, b_all, t, s))           # :138-139
                created.append(new)
        old = [tok for tok, d in zip(tokens, done) if not d]                   # (3)
        if frame_log is not None:
            frame_log.append(sorted(((tok.node, tok.score) for tok in old), key=lambda q: q[1]))
        drop = prune([tok.score for tok in old], beam, min_distinct)           # (4) pruning, :159-167
        old = [tok for i, tok in enumerate(old) if i not in drop]
        tokens = old + created
        if trace is not None:
            trace.append(len(tokens))
    best = sorted(range(len(tokens)), key=lambda i: -tokens[i].score)[:candidate]                 # transfer, :175-187 (stable)
    return [(tokens[i].node, tokens[i].score, tokens[i].hist) for i in best], history


def words_of(final_entry, history, tree):
    """The word sequence behind one final token: the chain of word-end nodes (each a list of homophones), then the
    words of the token's own node if words end there (transfer, Decoder.py:183-186)."""
    node, _, hist = final_entry
    chain = []
    while hist >= 0:
        hist, wnode = history[hist]
        chain.append(tree['words'][wnode])
    chain.reverse()
    if tree['node_word'][node]:
        chain.append(tree['words'][node])
    return chain

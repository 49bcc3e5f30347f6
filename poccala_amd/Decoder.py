"""Drop-in for the runnable part of the reference's Decoder.py (frame-synchronous token passing over the pronunciation
tree), batched on the GPU.

The reference module cannot be imported (it needs `LanguageModel.Ngram`, which is not in the repository, and an older
AcousticModel API).  Mirrored here: the module-level `beam = 0.85` (Decoder.py:34), Token.viterbi's
recursion and score (:250-288), token_passing / passing_in_word / pruning (:91-167) and transfer (:175-187) -- on the
device, for many utterances at once (hmm_decode.hip through pcl_batch_decode).  The gaps that had to be filled because the
source cannot run (D1-D5: exit test on the last emitting state, tokens keyed by tree node, all first characters start,
word ends re-seed the first characters with a uniform language model, "all step, then all hand over" frames) are listed in
include/poccala_hip.h and in the tests' CPU restatement.  PARITY UNPINNED against the reference.
"""
import numpy as np

from ._lib import PCL_F32
from .runtime import default_engine

beam = 0.85          # Decoder.py:34


def load_inventory(engine, unit_names, means, variances, weights, unit_trans, lexicon):
    """Upload what the decoder needs: the GMM states (unit-major, S-2 per unit), the unit transition matrices and the
    pronunciation tree compiled against `unit_names` (Lexicon.PronunciationLexicon).  Returns the compiled tree."""
    engine.load_model(means, variances, weights)
    engine.load_units(np.asarray(unit_trans))
    tree = lexicon.compile({u: i for i, u in enumerate(unit_names)})
    engine.load_lexicon(tree)
    return tree


def decode_batch(data_list, tree, engine=None, precision=PCL_F32, beam_=None, candidate=5, max_tokens=4096):
    """data_list: MFCC matrices (T_u, D).  Scores every GMM state for every frame (the decoder has no label) and runs the
    token passing.  Returns per utterance (words, score, detail): `words` = the word sequence behind the best final
    token -- each entry the list of homophones of a word-end node -- as transfer() reports it (Decoder.py:183-186)."""
    engine = engine or default_engine()
    lens = np.array([len(d) for d in data_list], dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    engine.load_frames(np.concatenate([np.asarray(d) for d in data_list], axis=0))
    b = engine.all_state_batch(lens, begin)
    b.score(precision)
    res = b.decode(beam if beam_ is None else beam_, 8, candidate, max_tokens)
    b.close()
    out = []
    for r in res:
        words, score = [], -np.inf
        if r['final']:
            node, score, h = r['final'][0]
            while h >= 0:
                h, wn = r['history'][h]
                words.append(tree['words'][wn])
            words.reverse()
            if tree['node_word'][node]:
                words.append(tree['words'][node])
        out.append((words, score, r))
    return out

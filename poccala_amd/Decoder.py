"""Drop-in for the runnable part of the reference's Decoder.py (frame-synchronous token passing over the pronunciation
tree), batched on the GPU.

The reference module cannot be imported (it needs `LanguageModel.Ngram`, which is not in the repository, and an older
AcousticModel API).  Mirrored here: the module-level `beam = 0.85` (Decoder.py:34), Token.viterbi's
recursion and score (:250-288), token_passing / passing_in_word / pruning (:91-167) and transfer (:175-187) -- on the
device, for many utterances at once (hmm_decode.hip through pcl_batch_decode).  The gaps that had to be filled because the
source cannot run (D1-D5: exit test on the last emitting state, tokens keyed by tree node, all first characters start,
word ends re-seed the first characters with a uniform language model, "all step, then all hand over" frames) are listed in
include/poccala_hip.h and in the tests' CPU restatement.  Parity: recursion, pruning, frame loop and in-word hand-over are pinned
by golden G14 (the reference's own pieces, run with a stand-in for the missing import); D1-D5 are the builder's completion.
"""
import os

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


def _report(res, tree):
    """Per utterance (words, score, detail): the word sequence behind the best final token, as transfer() reports it
    (Decoder.py:183-186) -- each entry the list of homophones of a word-end node."""
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


def decode_batch(data_list, tree, engine=None, precision=PCL_F32, beam_=None, candidate=5, max_tokens=4096):
    """data_list: MFCC matrices (T_u, D).  Scores every GMM state for every frame (the decoder has no label) and runs the
    token passing.  Returns per utterance (words, score, detail)."""
    engine = engine or default_engine()
    lens = np.array([len(d) for d in data_list], dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    engine.load_frames(np.concatenate([np.asarray(d) for d in data_list], axis=0))
    b = engine.all_state_batch(lens, begin)
    b.score(precision)
    res = b.decode(beam if beam_ is None else beam_, 8, candidate, max_tokens)
    b.close()
    return _report(res, tree)


def decode_stream(chunks, tree, engine=None, precision=PCL_F32, beam_=None, candidate=5, max_tokens=4096):
    """The streaming form (BASELINE config 5: a corpus that is fed chunk by chunk).  `chunks` yields lists of (T_u, D)
    float32 MFCC matrices; for every chunk, in order, decode_batch's result list is yielded.  Three legs overlap:
      copy stream    frames of chunk k+1 travel into the frame slot that is not being scored (Engine.stage_frames)
      main stream    every GMM state x every frame of chunk k is scored
      second stream  chunk k-1 is decoded (token passing), its results come down
    and the host packs chunk k+2 into page-locked memory meanwhile.  Batches are kept for the last few chunk shapes (the
    lengths of a chunk's utterances) and reused in rotation; a new shape costs a batch creation, which takes its buffers from
    the library's device-memory pool and copies its descriptors on the main stream -- it does not wait for the decoder
    running on the second stream."""
    engine = engine or default_engine()
    bm = beam if beam_ is None else beam_
    # batches (three per chunk shape, the four most recent shapes) and the page-locked staging buffer live with the ENGINE: a
    # second stream of the same chunk shapes (or the same stream read in several calls) starts warm; Engine.close() frees them
    pool, pinned = engine._stream_pool, engine._stream_pinned

    def pack(chunk):
        lens = np.array([len(d) for d in chunk], dtype=np.int32)
        begin = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
        rows, dim = int(lens.sum()), int(np.asarray(chunk[0]).shape[1])
        if pinned[0] is None or pinned[0].size < rows * dim:
            # grow-only, by doubling; the old buffer is free to go: every copy staged from it was waited for by swap_frames()
            grown = max(rows * dim, 2 * (pinned[0].size if pinned[0] is not None else 0))
            if pinned[0] is not None:
                engine.pinned_free(pinned[0])
            pinned[0] = engine.pinned_empty((grown,), np.float32)
        view = pinned[0][:rows * dim].reshape(rows, dim)
        np.concatenate([np.asarray(d, dtype=np.float32) for d in chunk], axis=0, out=view)
        engine.stage_frames(view)
        return lens, begin

    def batch_for(lens, begin, k, busy):
        key = (lens.tobytes(), engine.J, engine.n_nodes)
        ring = pool.pop(key, None) or [None, None, None]
        pool[key] = ring                                           # (most recently used last)
        for old in [q for q in pool if q != key][:max(0, len(pool) - 4)]:
            if not any(x is y for x in pool[old] if x is not None for y in busy):   # never a batch whose decoder is still running
                for x in pool.pop(old):
                    if x is not None:
                        x.close()
        if ring[k % 3] is None:
            ring[k % 3] = engine.all_state_batch(lens, begin)
        return ring[k % 3]

    # The host runs ONE chunk ahead of the GPU: when it waits for the decoder of chunk k-1, the scoring of chunk k+1 is already
    # queued behind the scoring of chunk k and the decoder of chunk k behind its scoring -- the main stream never runs dry and
    # decode(k) starts beside score(k+1) the moment score(k) is done (three batches per chunk shape in rotation: one decoding,
    # one scoring, one being fetched).
    it = iter(chunks)
    first = next(it, None)
    if first is None:
        return
    inflight = []
    # The scoring of chunk k leaves room for the token passing of chunk k-1 (Engine.score_occupancy) when that chunk's utterances are all
    # of one length: its 417 decode workgroups then need the CUs for the whole chunk and take turns with the scoring workgroups unless
    # those leave registers (+11 % on config 5's corpus).  Ragged chunks free CUs as their short utterances finish -- their token passing
    # already runs in its stand-alone time, and a thinner scoring kernel only costs (-3 %).  POCCALA_STREAM_OCCUPANCY=0: never (A/B).
    thin = os.environ.get('POCCALA_STREAM_OCCUPANCY', '2') != '0'

    def room_for(prev_lens):
        engine.score_occupancy(2 if thin and prev_lens is not None and len(prev_lens) and int(prev_lens.min()) == int(prev_lens.max()) else 0)
    try:
        layout = pack(first)
        engine.swap_frames()                                       # chunk 0 is the current frame matrix
        b = batch_for(layout[0], layout[1], 0, inflight)
        b.score(precision)
        prev = layout[0]
        nxt = next(it, None)
        if nxt is not None:
            layout = pack(nxt)                                     # host packing + H2D of chunk 1 beside the GPU work
        k = 0
        while b is not None:
            b.decode_launch(bm, 8, candidate, max_tokens)          # decode(k): second stream, behind score(k)
            inflight.append(b)
            b = None
            if nxt is not None:
                engine.swap_frames()                               # chunk k+1 is the current frame matrix
                b = batch_for(layout[0], layout[1], k + 1, inflight)
                room_for(prev)                                     # (chunk k is being decoded beside this launch)
                b.score(precision)                                 # score(k+1): main stream, behind score(k)
                prev = layout[0]
                nxt = next(it, None)
                if nxt is not None:
                    layout = pack(nxt)                             # chunk k+2 on its way (behind the last readers of its slot)
            if len(inflight) > 1 or b is None:
                done = inflight.pop(0)
                yield _report(done.decode_unpack(done.decode_fetch()), tree)   # waits for THAT decoder only
            k += 1
        while inflight:
            done = inflight.pop(0)
            yield _report(done.decode_unpack(done.decode_fetch()), tree)
    finally:
        engine.score_occupancy(0)
        inflight[:] = []                                           # (an abandoned stream: the batches stay in the engine's pool)

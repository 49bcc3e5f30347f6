#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: the per-GPU shard of the streaming 1M-frame corpus (417 utterances x 300 frames, 39-dim,
4096-mix, 183 XIF_tone-sized units = 549 GMM states scored for EVERY frame) + token-passing decode over a pronunciation
tree.  Streaming: the shard is cut into chunks; chunk k+1's frames go up (H2D) and chunk k-1's results come down while
chunk k is scored and decoded -- here measured as per-chunk phases; the GPU-resident rate is what the scoring roofline is
quoted on.  The tree is synthetic (the reference ships no word list): words of 1-4 characters drawn from the characters of
the golden lexicon fixture, read through the reference's Mandarin.dat rules (poccala_amd.Lexicon).

usage: c5_decode_bench.py [utterances] [mixtures] [words] [chunks] [max live tokens per utterance]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c = dict(synth.CONFIGS['C5shard'])
U = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
M = int(sys.argv[2]) if len(sys.argv) > 2 else c['M']
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
CH = int(sys.argv[4]) if len(sys.argv) > 4 else 3
CAP = int(sys.argv[5]) if len(sys.argv) > 5 else 2048      # live tokens per utterance (the reference's 15 % rule alone lets the set grow with the tree)
T, D, units_n = c['T'], c['D'], c['units']

tree, lx = synth.make_pronunciation_tree(NW, units_n)
print('tree: %d words -> %d nodes, %d first-character nodes, %d units in use of %d' % (lx.size, len(tree['names']), len(tree['roots']), len(set(tree['node_units'].ravel()) - {-1}), units_n))

mean, var, w, trans = synth.make_model(units_n, M, D)
frames, lens, begin = synth.make_frames(U, T, D)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_lexicon(tree)

# ---- resident: whole shard in HBM
eng.load_frames(frames)
b = eng.all_state_batch(lens, begin)
b.score(PCL_F32); res = b.decode(max_tokens=CAP); eng.sync()
eng.kernel_time('score'); eng.kernel_time('decode')
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    b.score(PCL_F32)
    res = b.decode(max_tokens=CAP)
eng.sync()
wall = (time.perf_counter() - t0) / reps
sc_ms, k1 = eng.kernel_time('score'); de_ms, k2 = eng.kernel_time('decode')
sc_ms /= k1; de_ms /= k2
pairs = U * T * units_n * 3
flop = pairs * M * (3 * D + 4)
ntok = np.concatenate([r['n_tokens'] for r in res])
steps = int(ntok.sum())
print('resident: %.1f ms/shard = %.3f M frames/s per GPU | score %.1f ms (%.0f TFLOP/s algorithmic, frac %.3f of 838.9) | decode %.1f ms '
      '(%.1f M token-steps/s, mean %.0f / max %d live tokens, overflow %d utterances)'
      % (wall * 1e3, U * T / wall / 1e6, sc_ms, flop / sc_ms / 1e9, flop / sc_ms / 1e9 / 838.9, de_ms, steps / de_ms / 1e3, ntok.mean(), ntok.max(),
         sum(r['overflow'] for r in res)))
print('decode ~%d bytes per token step (step: p in + out 128, score / flags / units 32; prune ~6 passes over score + flag 72; compaction 60) -> %.0f GB/s of 8000'
      % (292, steps * 292 / de_ms / 1e6))
b.close()

# ---- streaming: the shard arrives in CH chunks through poccala_amd.Decoder.decode_stream -- H2D of chunk k+1 on the copy stream,
#      scoring of chunk k on the main stream, token passing of chunk k-1 (and its results D2H) on the second stream
from poccala_amd import Decoder
per = (U + CH - 1) // CH
chunks = [[frames[begin[u]:begin[u] + lens[u]] for u in range(k * per, min(U, (k + 1) * per))] for k in range(CH)]
chunks = [c_ for c_ in chunks if c_]
for rep in range(2):                                               # the first pass creates the two batches and the slots
    t0 = time.perf_counter()
    outs = list(Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP))
    eng.sync()
    t_stream = time.perf_counter() - t0
same = all(o[2]['final'] == r['final'] for o, r in zip([x for ch in outs for x in ch], res))
print('streaming in %d chunks of %d utterances (copy / score / decode legs on three streams): %.1f ms/shard = %.3f M frames/s per GPU; results %s the resident run'
      % (len(chunks), per, t_stream * 1e3, U * T / t_stream / 1e6, 'equal' if same else 'DIFFER FROM'))
# the legs one after the other, for comparison
t_h2d = t_sc = t_de = 0.0
for ch in chunks:
    fl = np.concatenate(ch, axis=0)
    ln = np.array([len(x) for x in ch], dtype=np.int32)
    bg = np.concatenate([[0], np.cumsum(ln[:-1].astype(np.int64))]).astype(np.int64)
    t1 = time.perf_counter(); eng.load_frames(fl); t2 = time.perf_counter()
    bb = eng.all_state_batch(ln, bg)
    t3 = time.perf_counter(); bb.score(PCL_F32); eng.sync(); t4 = time.perf_counter()
    bb.decode(max_tokens=CAP); t5 = time.perf_counter()
    bb.close()
    t_h2d += t2 - t1; t_sc += t4 - t3; t_de += t5 - t4
print('the same chunks with the legs back to back: H2D %.1f ms + score %.1f ms + decode incl. results D2H %.1f ms = %.3f M frames/s per GPU'
      % (t_h2d * 1e3, t_sc * 1e3, t_de * 1e3, U * T / (t_h2d + t_sc + t_de) / 1e6))
# a ragged stream: every utterance a different length, so every chunk is a new shape -- a batch is created (and an old one dropped)
# per chunk, out of the device-memory pool, while the previous chunk's decoder runs
rl = np.random.default_rng(77).integers(150, T + 1, size=U)
rchunks = [[frames[begin[u]:begin[u] + rl[u]] for u in range(k * per, min(U, (k + 1) * per))] for k in range(CH)]
rchunks = [c_ for c_ in rchunks if c_]
for rep in range(2):
    t0 = time.perf_counter()
    routs = list(Decoder.decode_stream(iter(rchunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP))
    eng.sync()
    t_rag = time.perf_counter() - t0
t_b2b = 0.0
for ch in rchunks:
    t1 = time.perf_counter()
    Decoder.decode_batch(ch, tree, engine=eng, precision=PCL_F32, max_tokens=CAP)
    t_b2b += time.perf_counter() - t1
print('ragged stream (%d chunks, utterances of 150-%d frames, %d frames): %.1f ms through decode_stream = %.3f M frames/s; chunk by chunk with decode_batch: %.1f ms'
      % (len(rchunks), T, int(rl.sum()), t_rag * 1e3, rl.sum() / t_rag / 1e6, t_b2b * 1e3))
best = res[0]['final'][0] if res[0]['final'] else None
print('utterance 0: best token', best, 'history entries', len(res[0]['history']))

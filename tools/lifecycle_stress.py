#!/usr/bin/env python3
"""Context / batch lifecycle under load (one pcl_destroy faulted at a fixture teardown in round 5 and never again).

    python tools/lifecycle_stress.py --iters 200 [--seed S]

Every iteration: create a context, upload a model whose statistics block is large enough for the asynchronous pcl_stats_zero
(>= 64 MB: the auxiliary stream's memset), upload frames, create 3 label batches, and -- WITHOUT any synchronisation in between --
queue score + forward-backward (second stream) + accumulate (producer on the auxiliary stream) + per-unit merge + an asynchronous
result fetch (download stream) + stats_zero for the next E-step, then tear down in one of four ways chosen at random:

  close   Engine.close(): batches destroyed with work in flight (buried, reaped), page-locked result buffers freed, pcl_destroy
  raw     pcl_destroy(ctx) with the batches ALIVE and nothing waited for (the C caller who forgot everything), pageable destinations
  drop    the Python objects are simply dropped (garbage collection order decides: Batch.__del__ / Engine.__del__)
  mixed   half of the batches closed by hand, one fetch waited for, then Engine.close()

Two contexts are alive at once: context k is torn down while context k + 1 has its work in flight on the same device (the shared
device-memory pool hands k's blocks to k + 1).  Results of every fetch that was waited for are compared with a reference computed
once (same model, same frames): a torn copy or a block recycled too early shows as a mismatch, a use-after-free as a GPU fault.
Prints one JSON line; exit code 0 = every iteration clean."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--units', type=int, default=150)
    ap.add_argument('--mix', type=int, default=256)
    ap.add_argument('--utts', type=int, default=48)
    args = ap.parse_args()
    from poccala_amd import Engine, PCL_F32, synth
    rng = np.random.default_rng(args.seed)
    units, M, D, U, T, L = args.units, args.mix, 39, args.utts, 120, 6
    mean, var, w, trans = synth.make_model(units, M, D, seed=3)
    frames, lens, begin = synth.make_frames(3 * U, T, D, seed=4, ragged=True)
    labels = synth.make_labels(3 * U, L, units, seed=5)
    tr = np.stack(trans)
    assert units * 3 * M * (2 * D + 1) * 8 >= (64 << 20), 'the statistics block must take the asynchronous pcl_stats_zero path'

    def batches_of(eng):
        return [eng.label_batch(labels[U * k:U * (k + 1)], lens[U * k:U * (k + 1)], begin[U * k:U * (k + 1)]) for k in range(3)]

    def launch(eng, bs, pinned):
        """everything queued, nothing waited for; returns the fetch destinations"""
        eng.stats_zero()
        outs = []
        for k, b in enumerate(bs):
            b.score(PCL_F32)
            b.forward_backward(fix_pi=False)
            b.accumulate(PCL_F32)
            b.accumulate_hmm()
            if pinned:
                bufs = b.result_buffers(want=('logp', 'lgamma'), slot=k)
            else:
                sh = b._result_shapes()
                bufs = dict(logp=np.empty(sh['logp'][0]), lgamma=np.empty(sh['lgamma'][0]))
            b.fetch_async(bufs)
            outs.append(bufs)
        eng.stats_zero()                                  # the next E-step's zero, on the side stream, behind the accumulate passes
        return outs

    # reference results, once, with everything waited for
    e0 = Engine(0)
    e0.load_model(mean, var, w); e0.load_units(tr); e0.load_frames(frames)
    ref = []
    for b in batches_of(e0):
        b.score(PCL_F32); b.forward_backward(fix_pi=False)
        ref.append((b.get('logp').copy(), np.concatenate([l.T.reshape(-1) for l in b.get('lgamma')])))
    e0.close()

    def make():
        e = Engine(0)
        e.load_model(mean, var, w); e.load_units(tr); e.load_frames(frames)
        return e, batches_of(e)

    counts, checked, t0 = {}, 0, time.time()
    prev = None                                          # (engine, batches, outs, mode) of the context torn down NEXT iteration
    for it in range(args.iters + 1):
        cur = None
        if it < args.iters:
            mode = ['close', 'raw', 'drop', 'mixed'][int(rng.integers(4))]
            e, bs = make()
            outs = launch(e, bs, pinned=(mode != 'raw'))
            cur = (e, bs, outs, mode)
        if prev is not None:                             # tear the OLDER context down while the newer one's work is in flight
            e, bs, outs, mode = prev
            counts[mode] = counts.get(mode, 0) + 1
            if mode == 'close':
                e.close()
            elif mode == 'raw':
                ctx, e._ctx = e._ctx, None               # the Python objects become husks: nothing of theirs runs after this
                for b in bs:
                    b._b = None
                e._pinned = []
                rc = e._lib.pcl_destroy(ctx)
                assert rc == 0
                for k, o in enumerate(outs):             # pcl_destroy drained the streams: the copies have landed, whole
                    assert np.array_equal(o['logp'], ref[k][0]) and np.array_equal(o['lgamma'], ref[k][1], equal_nan=True), ('raw', it, k)
                    checked += 1
            elif mode == 'drop':
                del bs, outs
                prev = None
                del e
                gc.collect()
            else:
                bs[0].close()
                bs[1].fetch_wait()
                assert np.array_equal(outs[1]['logp'], ref[1][0]) and np.array_equal(outs[1]['lgamma'], ref[1][1], equal_nan=True), ('mixed', it)
                checked += 1
                e.close()
        prev = cur
        if it % 20 == 0:
            print('lifecycle_stress: iteration %d / %d (%.0f s)' % (it, args.iters, time.time() - t0), file=sys.stderr, flush=True)
    print(json.dumps(dict(ok=True, iterations=args.iters, teardown_modes=counts, fetches_checked=checked, seconds=round(time.time() - t0, 1),
                          model='%d states x %d mixtures, statistics block %.0f MB' % (units * 3, M, units * 3 * M * (2 * D + 1) * 8 / 2 ** 20))))


if __name__ == '__main__':
    main()

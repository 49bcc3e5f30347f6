"""Lifecycle and ordering hazards of the asynchronous parts of the boundary:
a second recursion queued behind a pending result fetch, stale views of a result slot that has grown, a slot shared by two batches
with copies in flight, and contexts torn down with everything in flight (tools/lifecycle_stress.py, a subprocess)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(eng, U, units=40, M=64, T=300, L=20, seed=31):
    from poccala_amd import synth
    mean, var, w, trans = synth.make_model(units, M, 39, seed=seed)
    frames, lens, begin = synth.make_frames(U, T, 39, seed=seed + 1)
    labels = synth.make_labels(U, L, units, seed=seed + 2)
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_frames(frames)
    return labels, lens, begin


def test_a_second_recursion_waits_for_the_result_copies_of_the_first():
    """score -> forward-backward -> fetch_async -> forward-backward AGAIN (other settings) with no fetch_wait in between, default
    knobs: the second recursion runs on the second stream, which under the fewer-markers shortcut follows the scoring's event and
    not the main stream's head -- it must wait for the copies itself, or the host receives ln gamma / ln P(O) torn between the
    two recursions (rounds 1-4 ordered it through the main stream).  40 MB of ln gamma per fetch: the copy takes
    longer than the recursion."""
    from poccala_amd import Engine, PCL_F32
    eng = Engine(0)
    try:
        labels, lens, begin = _setup(eng, 256)
        b = eng.label_batch(labels, lens, begin)
        b.score(PCL_F32)
        b.forward_backward(fix_pi=False)
        first = (b.get('logp').copy(), np.concatenate([l.T.reshape(-1) for l in b.get('lgamma')]))
        b.forward_backward(fix_pi=True, threshold=0.0)
        second = (b.get('logp').copy(), np.concatenate([l.T.reshape(-1) for l in b.get('lgamma')]))
        assert not np.array_equal(first[1], second[1], equal_nan=True)             # (the two recursions do differ: the test can see a tear)
        for rep in range(6):
            b.score(PCL_F32)
            b.forward_backward(fix_pi=False)
            bufs = b.result_buffers(want=('logp', 'lgamma'))
            b.fetch_async(bufs)
            b.forward_backward(fix_pi=True, threshold=0.0)                         # no fetch_wait: the device has to order this
            if rep % 2:
                b.viterbi()                                                        # ... and a third recursion behind it
            b.fetch_wait()
            assert np.array_equal(bufs['logp'], first[0]) and np.array_equal(bufs['lgamma'], first[1], equal_nan=True), 'torn results, repetition %d' % rep
            assert np.array_equal(np.concatenate([l.T.reshape(-1) for l in b.get('lgamma')]), second[1], equal_nan=True)
        # the accumulate pass reads ln gamma on the main stream: a recursion queued behind it must not overwrite what it reads
        b.score(PCL_F32); b.forward_backward(fix_pi=False)
        eng.stats_zero(); b.accumulate(PCL_F32)
        want = eng.stats_download(moments=False)
        for rep in range(4):
            b.score(PCL_F32); b.forward_backward(fix_pi=False)
            eng.stats_zero(); b.accumulate(PCL_F32)
            b.forward_backward(fix_pi=True, threshold=0.0)                         # (second stream, right behind the pass)
            got = eng.stats_download(moments=False)
            assert np.array_equal(got['acc'], want['acc']) and np.array_equal(got['alpha_acc'], want['alpha_acc']), 'repetition %d' % rep
        b.close()
    finally:
        eng.close()


def test_result_slots_stale_views_and_two_batches_on_one_slot():
    """result_buffers() hands out views of ONE engine-owned block per slot.  A dict from before the slot grew still passes
    fetch_async's size checks: its block must still be page-locked memory (retired, not freed).  Two batches asking for the same
    slot share memory: while one has un-waited copies into it the other is refused."""
    from poccala_amd import Engine, PCL_F32
    eng = Engine(0)
    try:
        labels, lens, begin = _setup(eng, 64)
        small = eng.label_batch(labels[:8], lens[:8], begin[:8])
        big = eng.label_batch(labels, lens, begin)
        for b in (small, big):
            b.score(PCL_F32); b.forward_backward(fix_pi=False)
        old = small.result_buffers(want=('logp', 'lgamma'), slot=5)
        held = eng.pinned_bytes()
        new = big.result_buffers(want=('logp', 'lgamma'), slot=5)                  # the slot grows: a new block
        assert eng.pinned_bytes() > held                                           # ... and the old one is still held
        assert old['lgamma'].__array_interface__['data'][0] != new['lgamma'].__array_interface__['data'][0]
        small.fetch_async(old)                                                     # the stale dict: a DMA into the RETIRED block
        small.fetch_wait()
        assert np.array_equal(old['logp'], small.get('logp'))
        assert np.array_equal(old['lgamma'], np.concatenate([l.T.reshape(-1) for l in small.get('lgamma')]), equal_nan=True)
        big.fetch_async(new)                                                       # un-waited copies into slot 5 ...
        with pytest.raises(RuntimeError, match='slot 5'):
            small.result_buffers(want=('logp',), slot=5)                           # ... another batch may not have it
        small.result_buffers(want=('logp',), slot=6)                               # (another slot is fine)
        assert big.result_buffers(want=('logp', 'lgamma'), slot=5)['logp'].shape == new['logp'].shape      # (the owner itself may ask again)
        big.fetch_wait()
        assert np.array_equal(new['logp'], big.get('logp'))
        again = small.result_buffers(want=('logp',), slot=5)                       # free once the copies were waited for
        small.fetch_async(again)
        small.close()                                                              # closing with copies in flight waits for them
        assert np.array_equal(again['logp'], old['logp'])
        big.close()
    finally:
        eng.close()


def test_contexts_torn_down_with_everything_in_flight():
    """tools/lifecycle_stress.py in a subprocess (a GPU fault takes the process, not the suite): 40 contexts, two alive at a time,
    torn down four ways with score / forward-backward / accumulate / fetch / stats_zero queued and nothing waited for.  The soak
    (tools/gpu_soak.sh) runs 200 iterations x 3 seeds."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'lifecycle_stress.py'), '--iters', '40', '--seed', '1'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d['ok'] and d['iterations'] == 40 and d['fetches_checked'] > 0 and len(d['teardown_modes']) == 4, d

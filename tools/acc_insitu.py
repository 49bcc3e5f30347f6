"""The accumulate pass timed alone and right after scoring / forward-backward (is the E-step's accumulate slower than the pass alone?)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
U, M, units, D, T, L = 1024, 2048, 1000, 39, 300, 20
BENCH = len(sys.argv) > 1                      # any argument: the model / frames / labels bench.py generates (two resident batches)
if BENCH:
    mean, var, w, trans = synth.make_model(units, M, D, seed=1)
    frames, lens, begin = synth.make_frames(2 * U, T, D, seed=0)
    labels = synth.make_labels(2 * U, L, units, seed=2)
    lens, begin, labels = lens[:U], begin[:U], labels[:U]
else:
    mean, var, w, trans = synth.make_model(units, M, D)
    frames, lens, begin = synth.make_frames(U, T, D)
    labels = synth.make_labels(U, L, units)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
b.score(PCL_F32); b.forward_backward(fix_pi=False); eng.stats_zero(); b.accumulate(PCL_F32); eng.sync()
def run(tag, pre):
    eng.kernel_time('accumulate')
    for _ in range(4):
        pre(); eng.stats_zero(); b.accumulate(PCL_F32); eng.sync()
    ms, k = eng.kernel_time('accumulate'); print('%-60s accumulate %.2f ms' % (tag, ms / k))
run('accumulate alone', lambda: None)
run('after score (synced)', lambda: (b.score(PCL_F32), eng.sync()))
run('after score + forward-backward (synced)', lambda: (b.score(PCL_F32), b.forward_backward(fix_pi=False), eng.sync()))
run('after score + forward-backward (not synced: the E-step)', lambda: (b.score(PCL_F32), b.forward_backward(fix_pi=False)))
run('after 5 x score (hot)', lambda: ([b.score(PCL_F32) for _ in range(5)], eng.sync()))
run('after a 50 ms pause', lambda: time.sleep(0.05))
if BENCH:                                      # bench.py's sequence: a second resident batch, alternating steps, Viterbi, then the E-step
    f2, l2, b2 = synth.make_frames(2 * U, T, D, seed=0)
    lab2 = synth.make_labels(2 * U, L, units, seed=2)
    bb = eng.label_batch(lab2[U:], l2[U:], b2[U:])
    for k in range(10):
        (b if k % 2 == 0 else bb).score(PCL_F32); (b if k % 2 == 0 else bb).forward_backward(fix_pi=False)
    eng.sync()
    run('with a second resident batch, after alternating steps', lambda: (b.score(PCL_F32), b.forward_backward(fix_pi=False)))
    b.viterbi(); eng.sync()
    run('... and after a Viterbi pass on the batch', lambda: (b.score(PCL_F32), b.forward_backward(fix_pi=False)))
    b.accumulate_hmm(); eng.sync()
    run('... and after accumulate_hmm', lambda: (b.score(PCL_F32), b.forward_backward(fix_pi=False)))
    run('... accumulate + accumulate_hmm in one E-step', lambda: (b.accumulate_hmm(), b.score(PCL_F32), b.forward_backward(fix_pi=False)))

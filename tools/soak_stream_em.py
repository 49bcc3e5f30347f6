import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth, Decoder
hip = C.CDLL('libamdhip64.so')
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
c = dict(synth.CONFIGS['C5shard'])
tree, lx = synth.make_pronunciation_tree(5000, c['units'])
mean, var, w, trans = synth.make_model(c['units'], 256, c['D'])
frames, lens, begin = synth.make_frames(96, 120, c['D'])
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_lexicon(tree)
chunks = [[frames[begin[u]:begin[u] + lens[u]] for u in range(k * 32, (k + 1) * 32)] for k in range(3)]
ref = None
for rep in range(40):
    outs = list(Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=2048))
    sig = [o[2]['final'] for ch in outs for o in ch]
    if ref is None: ref = sig
    assert sig == ref
    if rep in (1, 20, 39): print('pass %d free %.0f MiB' % (rep, free_mb()))
# E-step loop soak: model round trips
mean, var, w, trans = synth.make_model(20, 64, 39, seed=3)
frames, lens, begin = synth.make_frames(64, 100, 39, seed=4)
labels = synth.make_labels(64, 5, 20, seed=5)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)
b = eng.label_batch(labels, lens, begin)
for it in range(30):
    b.refresh_transitions(); b.score(PCL_F32); b.forward_backward(); eng.stats_zero(); b.accumulate_hmm()
    if it % 2: b.accumulate_exchange(PCL_F32, 1e-3, 0, True, n_chunks=4)
    else:
        b.accumulate(PCL_F32); eng.em_exchange(1e-3, 0, True)
    if it in (1, 15, 29): print('EM iteration %d free %.0f MiB logP %.6f' % (it, free_mb(), float(np.sum(b.get('logp')))))
b.close(); eng.close(); print('soak ok')

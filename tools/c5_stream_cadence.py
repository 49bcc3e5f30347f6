import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth, Decoder
c = dict(synth.CONFIGS['C5shard'])
CH, U, M, CAP = 12, 1668, 4096, 8192
tree, lx = synth.make_pronunciation_tree(20000, c['units'])
mean, var, w, trans = synth.make_model(c['units'], M, c['D'])
frames, lens, begin = synth.make_frames(U, c['T'], c['D'])
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_lexicon(tree)
per = (U + CH - 1) // CH
chunks = [[frames[begin[u]:begin[u] + lens[u]] for u in range(k * per, min(U, (k + 1) * per))] for k in range(CH)]
for rep in range(3):
    t0 = time.perf_counter(); stamps = []
    for out in Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP):
        stamps.append((time.perf_counter() - t0) * 1e3)
    eng.sync(); tot = (time.perf_counter() - t0) * 1e3
    print('pass %d: %.1f ms; yields at' % (rep, tot), ' '.join('%.0f' % s for s in stamps))

# where the host blocks: wall time of every engine / batch call of one more pass
import functools
log = []
def timed(name, f):
    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, (t - T0) * 1e3, (time.perf_counter() - t) * 1e3)); return r
    return g
from poccala_amd import engine as E_
for cls, names in ((E_.Engine, ['stage_frames', 'swap_frames', 'all_state_batch']), (E_.Batch, ['score', 'decode_launch', 'decode_fetch'])):
    for nme in names:
        setattr(cls, nme, timed(nme, getattr(cls, nme)))
T0 = time.perf_counter()
for out in Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP):
    log.append(('yield', (time.perf_counter() - T0) * 1e3, 0.0))
eng.sync()
for nme, at, dur in log[:60]:
    print('%8.1f  %-16s %7.2f ms' % (at, nme, dur))

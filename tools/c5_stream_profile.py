#!/usr/bin/env python3
"""Where the host's time goes in the streamed C5 run (Decoder.decode_stream): cProfile of two streamed passes over one shard."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth, Decoder
c = dict(synth.CONFIGS['C5shard'])
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 3
U, M, CAP = int(sys.argv[2]) if len(sys.argv) > 2 else 417, 4096, 8192
tree, lx = synth.make_pronunciation_tree(20000, c['units'])
mean, var, w, trans = synth.make_model(c['units'], M, c['D'])
frames, lens, begin = synth.make_frames(U, c['T'], c['D'])
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_lexicon(tree)
per = (U + CH - 1) // CH
chunks = [[frames[begin[u]:begin[u] + lens[u]] for u in range(k * per, min(U, (k + 1) * per))] for k in range(CH)]
for rep in range(2):
    t0 = time.perf_counter()
    outs = list(Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP))
    eng.sync()
    print('pass %d: %.1f ms' % (rep, (time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
outs = list(Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=CAP))
eng.sync()
t = time.perf_counter() - t0
pr.disable()
print('profiled pass: %.1f ms' % (t * 1e3))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)

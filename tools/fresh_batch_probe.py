#!/usr/bin/env python3
"""A corpus sweep hands every step a NEW (label, data) (AcousticModel.py:664-681, 861-870): the headline loop with the label
batch of every step created inside the loop (new labels; frames resident) against the same loop on resident batches.
Prints ms per step of both, the host time of every call, and whether the results of a fresh batch equal the resident ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'C4shard'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # batches alive at once
c = synth.CONFIGS[name]
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
frames, lens_all, begin_all = synth.make_frames(c['U'] * 2, c['T'], c['D'], seed=1000)
U = c['U']
pool = [synth.make_labels(U, c['L'], c['units'], seed=100 + k) for k in range(8)]     # 8 different label sets in rotation
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_units(np.stack(trans)); eng.load_frames(frames)

def resident(fetch=False, nres=2):
    bs = [eng.label_batch(pool[k], lens_all[U * (k % 2):U * (k % 2 + 1)], begin_all[U * (k % 2):U * (k % 2 + 1)]) for k in range(nres)]
    res = [b.result_buffers(('logp',), slot=k) for k, b in enumerate(bs)]
    for b in bs:
        b.score(PCL_F32); b.forward_backward()
    eng.sync()
    def step(k):
        b = bs[k % nres]
        if fetch and k >= nres:
            b.fetch_wait()
        b.score(PCL_F32); b.forward_backward()
        if fetch:
            b.fetch_async(res[k % nres])
    for k in range(5):
        step(k)
    eng.sync()
    eng.kernel_time('score'); eng.kernel_time('fb')
    t0 = time.perf_counter()
    for k in range(5, 5 + steps):
        step(k)
    eng.sync()
    el = time.perf_counter() - t0
    kt = (eng.kernel_time('score'), eng.kernel_time('fb'))
    for b in bs:
        b.close()
    print('  resident(fetch=%s, %d batches): %.3f ms/step, score kernel %.3f ms, fb span %.3f ms' % (fetch, nres, el / steps * 1e3, kt[0][0] / max(kt[0][1], 1), kt[1][0] / max(kt[1][1], 1)))
    return el / steps * 1e3

def fresh(check=False):
    live, host = [], dict(create=0.0, score=0.0, fb=0.0, close=0.0)
    logp = {}
    def one(k, timed):
        t = time.perf_counter()
        b = eng.label_batch(pool[k % len(pool)], lens_all[U * (k % 2):U * (k % 2 + 1)], begin_all[U * (k % 2):U * (k % 2 + 1)])
        t1 = time.perf_counter(); b.score(PCL_F32)
        t2 = time.perf_counter(); b.forward_backward()
        res = b.result_buffers(('logp',), slot=k % (depth + 1)); b.fetch_async(res)
        t3 = time.perf_counter()
        live.append((k, b, res))
        if len(live) > depth:
            kk, old, r = live.pop(0)
            old.fetch_wait()
            if check and kk < 8:
                logp[kk] = r['logp'].copy()
            old.close()
        t4 = time.perf_counter()
        if timed:
            host['create'] += t1 - t; host['score'] += t2 - t1; host['fb'] += t3 - t2; host['close'] += t4 - t3
    for k in range(8):
        one(k, False)
    eng.sync()
    eng.kernel_time('score'); eng.kernel_time('fb')
    t0 = time.perf_counter()
    for k in range(8, 8 + steps):
        one(k, True)
    eng.sync()
    el = time.perf_counter() - t0
    kt = (eng.kernel_time('score'), eng.kernel_time('fb'))
    print('  fresh: %.3f ms/step, score kernel %.3f ms, fb span %.3f ms' % (el / steps * 1e3, kt[0][0] / max(kt[0][1], 1), kt[1][0] / max(kt[1][1], 1)))
    for _, b, _r in live:
        b.close()
    live.clear()
    return el / steps * 1e3, {k: v / steps * 1e3 for k, v in host.items()}, logp

if os.environ.get('PROBE_TRACE'):
    resident(fetch=True); resident(fetch=False); f, host, logp = fresh(); eng.close(); sys.exit(0)
r = resident()
f, host, logp = fresh(check=True)
r2 = resident()
f2, host2, _ = fresh()
if os.environ.get('PROBE_TRACE'):
    resident(fetch=True); resident(fetch=False); sys.exit(0)
resident(fetch=True); resident(fetch=True, nres=4); resident(nres=4)
print('%s: resident %.3f / %.3f ms per step; fresh batch every step %.3f / %.3f ms (ratio %.3f); host ms per step %s' % (name, r, r2, f, f2, min(r, r2) / min(f, f2), {k: round(v, 3) for k, v in host2.items()}))
# same bits as a resident batch of the same labels
ok = True
for k, lp in logp.items():
    b = eng.label_batch(pool[k % len(pool)], lens_all[U * (k % 2):U * (k % 2 + 1)], begin_all[U * (k % 2):U * (k % 2 + 1)])
    b.score(PCL_F32); b.forward_backward()
    ok = ok and bool(np.array_equal(b.get('logp'), lp))
    b.close()
print('fresh == resident bit for bit:', ok)
eng.close()

#!/usr/bin/env python3
"""Board power and clock (rocm-smi) while the accumulate pass / the scoring kernel run back to back on the C4 shard:
is the E-step's accumulate pass power limited like the scoring kernel?  usage: acc_power.py [seconds]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import Engine, PCL_F32, synth
from poccala_amd.engine import make_sentence_batch

def smi():
    try:
        out = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        sclk = next((v for k, v in card.items() if 'sclk' in k.lower()), None)
        power = next((v for k, v in card.items() if 'power' in k.lower()), None)
        return float(str(sclk).strip('()').lower().replace('mhz', '')), float(power)
    except Exception as e:
        return None

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
U, M, units, D, T, L = 1024, 2048, 1000, 39, 300, 20
mean, var, w, trans = synth.make_model(units, M, D)
frames, lens, begin = synth.make_frames(U, T, D)
labels = synth.make_labels(U, L, units)
eng = Engine(0); eng.enable_timing(True)
eng.load_model(mean, var, w); eng.load_frames(frames)
b, n = make_sentence_batch(eng, labels, lens, begin, trans)
b.score(PCL_F32); b.forward_backward(fix_pi=False)
eng.stats_zero(); b.accumulate(PCL_F32); eng.sync()
print('idle', smi())
for name, fn in (('accumulate', lambda: b.accumulate(PCL_F32)), ('score', lambda: b.score(PCL_F32))):
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            r = smi()
            if r: samples.append(r)
            time.sleep(0.05)
    eng.kernel_time(name)
    th = threading.Thread(target=sampler, daemon=True); th.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(4): fn()
        eng.sync(); k += 4
    stop.set(); th.join(timeout=5)
    ms, cnt = eng.kernel_time(name)
    p = [s[1] for s in samples]; c = [s[0] for s in samples]
    print('%s: %.2f ms/launch over %d launches; power median %.0f W (min %.0f max %.0f), sclk field median %.0f MHz, %d samples'
          % (name, ms / max(cnt, 1), cnt, np.median(p), min(p), max(p), np.median(c), len(p)))

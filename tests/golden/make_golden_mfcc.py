#!/usr/bin/env python3
"""Golden vectors for the MFCC front-end (next row f4) by RUNNING the reference's AudioProcessing.MFCC
(StatisticalModel/AudioProcessing.py:184-448) on synthetic signals.  Build container only.

    python tests/golden/make_golden_mfcc.py        # writes tests/golden/G10_mfcc.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


class FakeWav(object):
    """Just enough of wave.Wave_read for MFCC.params (AudioProcessing.py:118-126)."""

    def __init__(self, framerate, nframes):
        self._p = (1, 2, framerate, nframes, 'NONE', 'not compressed')

    def getparams(self):
        return self._p


def synth_signal(rng, n, rate):
    t = np.arange(n) / rate
    s = (3000 * np.sin(2 * np.pi * 220 * t) + 1500 * np.sin(2 * np.pi * 1330 * t + 0.3) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
         + 800 * rng.standard_normal(n))
    s = np.round(s).astype(np.int16)
    s[s == 0] = 1          # init_audio deletes zero samples (AudioProcessing.py:176); keep the fixture explicit
    return s


def main():
    import_reference()
    from StatisticalModel.AudioProcessing import AudioProcessing
    rng = np.random.default_rng(1010)
    out = {}
    for tag, (n, rate) in (('a', (9000, 16000)), ('b', (5123, 8000))):
        sig = synth_signal(rng, n, rate)
        m = AudioProcessing.MFCC(13)
        m._MFCC__wdata = sig
        m._MFCC__wav = FakeWav(rate, n)
        out['signal_' + tag] = sig
        out['rate_' + tag] = np.int64(rate)
        out['mfcc13_' + tag] = m.mfcc()
        out['mfcc13_noenergy_' + tag] = m.mfcc(cal_energy=False)
        out['mfcc26_' + tag] = m.mfcc(d1=True)
        out['mfcc39_' + tag] = m.mfcc(d1=True, d2=True)
        # intermediates that pin the quirks
        pe = m.pre_emphasis(m.data)
        fb = m.frame_blocking(pe, rate)
        out['nframes_' + tag] = np.int64(fb.shape[0])
        win = m.hamming_window(fb.copy())
        spec = m.fft(win, 512)
        fbank, energy = m.mel_filter_bank(spec, rate, nfft=512)
        out['spec_rows_' + tag] = spec[[0, 1, fb.shape[0] // 2, fb.shape[0] - 1]]
        out['fbank_' + tag] = fbank
        out['energy_' + tag] = energy
    np.savez_compressed(os.path.join(HERE, 'G10_mfcc.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', v))


if __name__ == '__main__':
    main()

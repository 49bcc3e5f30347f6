"""Drop-in for the feature-extraction part of StatisticalModel/AudioProcessing.py: `AudioProcessing.MFCC`
(wav -> (T, 13/26/39) float64 MFCC matrix), computed on the GPU (csrc/mfcc.hip) through `pcl_mfcc`.

This is SURVEY.md section 8(f) row 4, the step BEFORE the hot path.  Mirrored: MFCC(vec_num), init_audio,
data, params, mfcc(sampletime, overlap, nfft, cal_energy, d1, d2) (AudioProcessing.py:100-448); added:
`mfcc_batch` for many signals in one launch.  Recording / playback (pyaudio) and the VAD are out of scope.
The reference's conventions are kept exactly (they are pinned by tests/golden/G10_mfcc.npz):
the "Hamming window" is one factor per frame, the spectrum is the rFFT magnitude, every mel filter is two
rising ramps, the DCT kernel is cos(pi (2k-1) j / 2N) * 2/sqrt(N), c0 is ln(sum of magnitudes).
"""
import ctypes as C
import math
import wave

import numpy as np

from .._lib import as_c, ptr
from ..runtime import default_engine


def frame_count(n_samples, framerate, sampletime=0.025, overlap=0.5):
    size = int(framerate * sampletime)
    step = int(size * overlap)
    return 1 + math.ceil((n_samples - size) / step)                     # AudioProcessing.py:216-219


def mel_filter_matrix(samplerate, nfft=512, filterbanks=26, low_hz=0.0, high_hz=None):
    """(filterbanks, nfft/2+1) frequency responses as mel_filter_bank builds them (AudioProcessing.py:312-335):
    centres linear in mel = 2595 ln(1 + f/700), FFT bins floor((nfft+1) f / rate), and for filter i a ramp from 0
    on [bin_i, bin_i+1) followed by ANOTHER ramp from 0 on [bin_i+1, bin_i+2)."""
    high_hz = high_hz or samplerate / 2
    lo_mel, hi_mel = 2595 * math.log(1 + low_hz / 700), 2595 * math.log(1 + high_hz / 700)
    centres_hz = 700 * (np.exp(np.linspace(lo_mel, hi_mel, filterbanks + 2) / 2595) - 1)
    bins = np.floor((nfft + 1) / samplerate * centres_hz)
    resp = np.zeros((filterbanks, nfft // 2 + 1))
    for i in range(filterbanks):
        for a, b in ((i, i + 1), (i + 1, i + 2)):
            start, stop = int(bins[a]), int(bins[b])
            for j in range(start, stop):
                resp[i][j] = (j - start) / (bins[b] - bins[a])
    return resp


def dct_basis(filterbanks, rank):
    """(rank, filterbanks): 2/sqrt(N) cos(pi (2k-1) j / (2N)) (AudioProcessing.py:362-369)."""
    k = np.arange(filterbanks)[None, :]
    j = np.arange(rank)[:, None]
    return (2 / filterbanks ** 0.5) * np.cos(np.pi * (2 * k - 1) * j / (2 * filterbanks))


def mfcc_batch(signals, framerate, vec_num=13, sampletime=0.025, overlap=0.5, nfft=512, filterbanks=26, cal_energy=True,
               d1=False, d2=False, engine=None):
    """MFCC matrices of many signals in one launch: list of (T_u, vec_num * {1,2,3}) float64 arrays."""
    eng = engine or default_engine()
    sigs = [np.asarray(s, dtype=np.float64).reshape(-1) for s in signals]
    off = np.concatenate([[0], np.cumsum([len(s) for s in sigs])]).astype(np.int64)
    frames = [frame_count(len(s), framerate, sampletime, overlap) for s in sigs]
    rows = int(sum(frames))
    dim = vec_num * (3 if (d1 and d2) else 2 if d1 else 1)
    n = np.arange(nfft)
    twc, tws = as_c(np.cos(2 * np.pi * n / nfft), np.float64), as_c(-np.sin(2 * np.pi * n / nfft), np.float64)
    resp = as_c(mel_filter_matrix(framerate, nfft, filterbanks), np.float64)
    dct = as_c(dct_basis(filterbanks, vec_num), np.float64)
    flat = as_c(np.concatenate(sigs), np.float64)
    out = np.empty((rows, dim))
    flags = (1 if cal_energy else 0) | (2 if d1 else 0) | (4 if (d1 and d2) else 0)
    eng._check(eng._lib.pcl_mfcc(eng._ctx, len(sigs), ptr(flat), ptr(off), int(framerate), float(sampletime), float(overlap),
                                 int(nfft), int(filterbanks), int(vec_num), flags, ptr(twc), ptr(tws), ptr(resp), ptr(dct),
                                 ptr(out), C.c_int64(rows)))
    cuts = np.cumsum(frames)[:-1]
    return np.split(out, cuts)


class AudioProcessing(object):
    class MFCC(object):
        def __init__(self, vec_num=13):
            self.__wav = None
            self.__wdata = None
            self.__params = None
            self.__vec_num = vec_num

        @property
        def data(self):
            return self.__wdata

        @property
        def wav(self):
            return self.__wav

        @property
        def params(self):
            """(nchannels, sampwidth, framerate, nframes, comptype, compname) (AudioProcessing.py:118-126)."""
            return self.__params if self.__params is not None else self.__wav.getparams()

        def init_audio(self, wav=None, path=None, show_pic=False):
            """Read 16-bit PCM; stereo keeps the larger of the two channel samples; zero samples are removed
            (AudioProcessing.py:128-176)."""
            self.__wav = wav if wav is not None else wave.open(path, 'rb')
            self.__params = None
            raw = self.__wav.readframes(self.__wav.getnframes())
            data = np.frombuffer(raw, dtype=np.short).copy()
            if self.__wav.getnchannels() == 2:
                data = data.reshape(-1, 2)
                data = np.maximum(data[:, 0], data[:, 1])
            self.__wdata = data[data != 0]

        def set_signal(self, samples, framerate):
            """Use an in-memory signal instead of a wav file."""
            self.__wdata = np.asarray(samples)
            self.__params = (1, 2, int(framerate), len(self.__wdata), 'NONE', 'not compressed')

        def mfcc(self, sampletime=0.025, overlap=0.5, nfft=512, cal_energy=True, d1=False, d2=False):
            return mfcc_batch([self.__wdata], self.params[2], self.__vec_num, sampletime, overlap, nfft, 26, cal_energy,
                              d1, d2)[0]

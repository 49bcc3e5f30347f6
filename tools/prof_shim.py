#!/usr/bin/env python3
"""cProfile of one flush of the deferred worker shim on config 2's shapes (128 utterances): where do the ~50 ms go?"""
import cProfile, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poccala_amd import synth
from poccala_amd.AcousticModel.AcousticModel import AcousticModel
from poccala_amd.Exceptions import NullLog
from poccala_amd.StatisticalModel.Clustering import Clustering
from poccala_amd.StatisticalModel.LHMM import LHMM
c = synth.CONFIGS['C2']
mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=3)
log = NullLog()
def unit_hmm(u):
    gm = [Clustering.GMM(log, dimension=c['D'], mix_level=c['M'], alpha=w[u * 3 + k], mean=mean[u * 3 + k], covariance=var[u * 3 + k], gmm_id=k, precision='f32') for k in range(3)]
    prof = [AcousticModel.VirtualState(1.)] + gm + [AcousticModel.VirtualState(0.)]
    return LHMM({i: str(u) for i in range(5)}, 5, log, transmat=trans[u].copy(), profunc=prof)
units = {str(u): unit_hmm(u) for u in range(c['units'])}
n = int(sys.argv[1]) if len(sys.argv) > 1 else c['U']
fr, ln, bg = synth.make_frames(n, c['T'], c['D'], seed=6)
lab = [[str(int(u)) for u in l] for l in synth.make_labels(n, c['L'], c['units'], seed=7)]
xs = [fr[bg[k]:bg[k] + ln[k]] for k in range(n)]
tmp = tempfile.mkdtemp(prefix='poccala_shim_')
am = AcousticModel(log, 'XIF_tone', state_num=5, parameters_path=tmp)
am.worker_units = units
am.flush_frames = 1 << 30
def once():
    for k in range(n):
        am.multi_embedded_training_1(lab[k], xs[k], False, False, k + 1, n, 0)
    return am.flush_workers()
once()
t0 = time.perf_counter(); once(); print('flush: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
shutil.rmtree(tmp, ignore_errors=True)

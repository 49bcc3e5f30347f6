import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import poccala_oracle as po
from poccala_amd import Engine, PCL_F32, PCL_F64, synth
from poccala_amd.engine import make_sentence_batch
units, M, D, U, T, L = 5, 16, 39, 6, 60, 4
mean, var, w, trans = synth.make_model(units, M, D, seed=301)
frames, lens, begin = synth.make_frames(U, T, D, seed=302, ragged=True)
labels = synth.make_labels(U, L, units, seed=303)
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_frames(frames)
b, n = make_sentence_batch(eng, labels, lens, begin, trans)
for P, name in ((PCL_F64, 'f64'), (PCL_F32, 'f32')):
    b.score(P); b.forward_backward(); eng.stats_zero(); b.accumulate(P)
    st = eng.stats_download()
    B, lg = b.get('B'), b.get('lgamma')
    J = mean.shape[0]
    ref = dict(acc=np.zeros((J, M)), alpha_acc=np.zeros(J), mean_acc=np.zeros((J, M, D)), cov_acc=np.zeros((J, M, D)))
    for u, lab in enumerate(labels):
        x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
        for pos, unit in enumerate(lab):
            for k in range(3):
                j = unit * 3 + k; row = 1 + pos * 3 + k
                rec = po.gmm_component_loglik(x, mean[j], var[j], w[j])      # (T,M)
                g = np.exp(rec + (lg[u][row] - B[u][row])[:, None])
                ref['acc'][j] += g.sum(0)
                ref['alpha_acc'][j] += np.exp(lg[u][row]).sum()
                ref['mean_acc'][j] += (g[:, :, None] * (x[:, None, :] + 100.0)).sum(0)
                ref['cov_acc'][j] += (g[:, :, None] * (x[:, None, :] - mean[j][None]) ** 2).sum(0)
    for key in ref:
        m = ref[key] != 0
        print(name, key, 'max rel err vs numpy-from-GPU-posteriors', np.abs(st[key][m] / ref[key][m] - 1).max())

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import poccala_oracle as po
from poccala_amd import Engine, PCL_F32, PCL_F64
rng = np.random.default_rng(3)
M, D, T = 16, 39, 1
mean = rng.standard_normal((1, M, D)); var = rng.uniform(0.5, 2, (1, M, D)); w = rng.dirichlet(np.ones(M))[None]
x = rng.standard_normal((T, D)).astype(np.float32)
eng = Engine(0)
eng.load_model(mean, var, w); eng.load_frames(x)
b = eng.batch([3], [T], [0])
b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
A = np.array([[0, 1, 0], [0, .5, .5], [0, 0, 0.]])
with np.errstate(divide='ignore'):
    b.set_transitions([np.log(A)], [np.log(np.ones(3) / 3)])
b.score(PCL_F64); b.forward_backward(fix_pi=True); eng.stats_zero(); b.accumulate(PCL_F64)
st = eng.stats_download()
B, lg = b.get('B')[0], b.get('lgamma')[0]
rec = po.gmm_component_loglik(x.astype(np.float64), mean[0], var[0], w[0])
g = np.exp(rec + (lg[1] - B[1])[:, None]).sum(0)
print('lg', lg[:, 0], 'B', B[:, 0])
print('gpu/ref - 1:', st['acc'][0] / g - 1)
print('sum gpu', st['acc'][0].sum(), 'alpha', st['alpha_acc'][0], 'sum ref', g.sum())

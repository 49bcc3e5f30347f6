import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import test_gpu_fuzz_hmm as t
from oracle import poccala_oracle as po
from poccala_amd import Engine
np.set_printoptions(precision=5, linewidth=220)
e = Engine(0)
for seed, only in ((14, 0), (15, 0), (19, 6), (22, 0), (49, 0), (2, None), (5, None)):
    c = t.draw(seed)
    U = len(c['hmms'])
    N = np.array([h[0].shape[0] for h in c['hmms']], dtype=np.int32); T = np.array([b.shape[1] for b in c['Bs']], dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(T[:-1])]).astype(np.int64)
    e.load_frames(np.zeros((int(T.sum()), max(e.D, 1)), dtype=np.float32))
    b = e.batch(N, T, begin)
    with np.errstate(divide='ignore'):
        b.set_transitions([np.log(h[0]) for h in c['hmms']], [np.log(h[1]) for h in c['hmms']])
    b.set_emissions(c['Bs']); b.forward_backward(fix_pi=c['fix_pi'], threshold=c['threshold']); b.viterbi()
    out = {k: b.get(k) for k in ('ksai', 'gamma', 'logp', 'npass', 'qtrace', 'path', 'point', 'lgamma')}
    b.close()
    print('==== seed', seed, 'fix_pi', c['fix_pi'], 'thr', c['threshold'])
    for u in (range(U) if only is None else [only]):
        a, pi, kind = c['hmms'][u]; B = c['Bs'][u]
        with np.errstate(all='ignore'):
            bw = po.baum_welch(a, pi, [B], fix_code=1 if c['fix_pi'] else 0, threshold=c['threshold'])
            rp, rpath = po.viterbi(a, pi, B)
        nan_dev, nan_or = np.isnan(out['ksai'][u]), np.isnan(bw['ksai'])
        msg = []
        if int(out['npass'][u]) != int(bw['n_pass']): msg.append('npass dev %d oracle %d; q dev %s oracle %s' % (out['npass'][u], bw['n_pass'], out['qtrace'][u][:6] if np.ndim(out['qtrace'][u]) else out['qtrace'][u], bw['q_trace'][:6]))
        if not np.array_equal(nan_dev, nan_or): msg.append('ksai NaN: dev %d oracle %d of %d; -inf dev %d oracle %d' % (nan_dev.sum(), nan_or.sum(), nan_or.size, np.isneginf(out['ksai'][u]).sum(), np.isneginf(bw['ksai']).sum()))
        if not np.array_equal(out['path'][u].astype(np.float64), rpath): 
            d = np.nonzero(out['path'][u] != rpath)[0]
            msg.append('path differs at %d frames, first %d: dev %d oracle %d' % (len(d), d[0], out['path'][u][d[0]], rpath[d[0]]))
        if not (rp == out['point'][u]): msg.append('point dev %r oracle %r' % (out['point'][u], rp))
        if msg: print(' utt', u, kind, 'N', N[u], 'T', T[u], 'logp dev', out['logp'][u], 'oracle', bw['logp'][0], '|', ' ; '.join(msg))

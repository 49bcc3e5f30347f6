"""CPU oracle for the Poccala GMM-HMM hot path  --  TEST INFRASTRUCTURE ONLY.

A float64 NumPy restatement of the algorithm the reference runs on its hot path
(SURVEY.md section 8a, rows A1..A16).  It exists to check the HIP kernels and to
serve as the timed CPU baseline in bench.py.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; the product package
(poccala_amd/) never does and fails loudly when the HIP library is missing.

Pinning: every function below is checked in tests/test_oracle_golden.py against
golden vectors produced by running the reference itself in the build container
(tests/golden/make_golden.py, groups G1..G9; make_golden_multi.py / _regroup.py G11, G12), rtol <= 1e-10; round 5 added
the reference's own runs on what the randomised GPU tests draw: G16 (make_golden_kinds.py: mixtures at the 1e-6
variance floor, variances over four decades in a state, weights down to 1e-12, E-step and M-step) and G15
(make_golden_edges.py: utterances of one to three frames, P(O) = 0, a zero-occupancy mixture, Viterbi with
end_state_back on 2-6 states, GMM.point on a frame 30 sigma from the state's centre).  `token_viterbi_step`
(A16) is pinned through golden G14 (tests/golden/make_golden_decoder.py runs the
reference's Decoder.Token.viterbi with a stand-in for the one module it cannot
import; tests/test_decoder_golden.py + test_decoder_oracle.py).

Each function cites the reference file:line it follows (paths relative to the
reference checkout).  Reference quirks Q1..Q9 are kept.
Everything is written vectorised; the `faithful_*` functions reproduce the
reference's per-frame / per-mixture loop nest and exist only so the CPU
baseline can be timed the way the reference would actually run.
"""
import math

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)
NEG_INF = -np.inf


# --------------------------------------------------------------------------
# A2 / A3  log-sum-exp primitives                     StatisticalModel/util.py:54-92
# --------------------------------------------------------------------------
def lse(v, axis=None):
    """max-shifted log-sum-exp; returns the max itself when |max| is inf
    (quirk Q4, util.py:62-66).  `axis=None` reduces everything (util.py:77);
    an int reduces that axis."""
    v = np.asarray(v, dtype=np.float64)
    with np.errstate(all='ignore'):
        mx = np.max(v, axis=axis, keepdims=True)
        safe = np.where(np.isinf(mx), 0.0, mx)
        s = np.sum(np.exp(v - safe), axis=axis, keepdims=True)
        out = np.where(np.isinf(mx), mx, safe + np.log(s))
    if axis is None:
        return float(out.reshape(()))
    return np.squeeze(out, axis=axis)


def lse_rows(a):
    """log_sum_exp(p_list, vector=True): reduce each first-axis slice entirely
    (util.py:68-75)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return a.copy()  # each "row" is a scalar: cal(scalar) == scalar unless inf (same value)
    return lse(a.reshape(a.shape[0], -1), axis=1)


def matrix_lse(mats, axis_x):
    """matrix_log_sum_exp: elementwise LSE over a list of equal-shape matrices,
    first `axis_x` rows (util.py:80-92)."""
    stack = np.stack([np.asarray(m, dtype=np.float64)[:axis_x] for m in mats], axis=0)
    return lse(stack, axis=0)


def logaddexp_q4(a, b):
    """Two-operand form of `lse` with the same inf behaviour."""
    return lse(np.stack([np.asarray(a, np.float64), np.asarray(b, np.float64)], axis=0), axis=0)


# --------------------------------------------------------------------------
# A1  diagonal Gaussian log-density                   StatisticalModel/util.py:20-31
# --------------------------------------------------------------------------
def gaussian_logpdf(y, mean, var):
    """log branch of gaussian_function, INCLUDING quirk Q1: the constant is
    -D/2 ln(2 pi) - 1/2 * sum(var) (sum of variances, util.py:29), not the
    log-determinant.  y (..., D), mean (D,), var (D,) diagonal."""
    y = np.asarray(y, np.float64)
    d = y.shape[-1]
    x = y - mean
    const = -d / 2.0 * LOG_2PI - 0.5 * np.sum(var)
    return const - 0.5 * np.sum(x * (1.0 / var) * x, axis=-1)


def gmm_component_loglik(x, mean, var, w):
    """The reference's `record`: ln w_m + A1(x_t; mu_m, var_m) for every frame
    and mixture (Clustering.py:755-760).  x (T,D); mean,var (M,D); w (M,) -> (T,M)."""
    x = np.asarray(x, np.float64)
    if x.shape[-1] != mean.shape[1]:
        raise ValueError('data dimension %d != model dimension %d' % (x.shape[-1], mean.shape[1]))
    d = mean.shape[1]
    with np.errstate(divide='ignore'):
        const = np.log(w) - d / 2.0 * LOG_2PI - 0.5 * np.sum(var, axis=1)       # (M,)
    diff = x[:, None, :] - mean[None, :, :]                                      # (T,M,D)
    quad = np.sum(diff * (1.0 / var)[None] * diff, axis=2)                        # (T,M)
    return const[None, :] - 0.5 * quad


def gmm_point(x, mean, var, w, record=False):
    """A4  Clustering.GMM.point(log=True) for a block of frames
    (Clustering.py:740-767).  Returns ln b(x_t) (T,), and the record if asked."""
    comp = gmm_component_loglik(x, mean, var, w)
    out = lse(comp, axis=1)
    return (out, comp) if record else out


def gmm_point_blocked(x, mean, var, w, m_chunk=128, t_chunk=16):
    """gmm_point evaluated in blocks of t_chunk frames x m_chunk mixtures, so that the (t, m, D) temporaries of
    gmm_component_loglik stay in a core's cache (16 x 128 x 39 float64 = 640 KB) instead of streaming through DRAM
    (25 x 2048 x 39 = 16 MB per temporary, which is what every core of a many-core host then fights over).  The arithmetic per
    (frame, mixture) is gmm_component_loglik's, element for element, and the log-sum-exp runs over the assembled (T, M) record:
    the same bits as gmm_point (tests/test_oracle_golden.py).  Timing use (bench.py cpu_baseline)."""
    x = np.asarray(x, np.float64)
    T, M = x.shape[0], mean.shape[0]
    comp = np.empty((T, M))
    for t0 in range(0, T, t_chunk):
        for m0 in range(0, M, m_chunk):
            comp[t0:t0 + t_chunk, m0:m0 + m_chunk] = gmm_component_loglik(x[t0:t0 + t_chunk], mean[m0:m0 + m_chunk], var[m0:m0 + m_chunk], w[m0:m0 + m_chunk])
    return lse(comp, axis=1)


def gmm_point_gemm(x, mean, var, w):
    """The same log-likelihoods through the EXPANDED quadratic form, one float64 GEMM per state:
    ln w - D/2 ln 2pi - 1/2 sum var - 1/2 sum_d (x_d^2 / var_d - 2 x_d mu_d / var_d + mu_d^2 / var_d).  Not the reference's
    order of operations (util.py:22-31 subtracts first) -- it is the formulation the GPU kernels use, and what an optimised CPU
    implementation of this path would do (BLAS); agrees with gmm_point to ~1e-12 relative on well-scaled data
    (tests/test_oracle_golden.py).  Timing use (bench.py cpu_baseline, the strongest CPU leg)."""
    x = np.asarray(x, np.float64)
    d = mean.shape[1]
    xe = np.concatenate([x * x, x, np.ones((x.shape[0], 1))], axis=1)               # (T, 2D+1)
    with np.errstate(divide='ignore'):
        k = np.log(w) - d / 2.0 * LOG_2PI - 0.5 * var.sum(1) - 0.5 * (mean * mean / var).sum(1)
    p = np.concatenate([-0.5 / var, mean / var, k[:, None]], axis=1)                 # (M, 2D+1)
    return lse(xe @ p.T, axis=1)


def faithful_gmm_point(x_t, mean, var, w):
    """Same arithmetic as gmm_point for ONE frame, with the reference's loop
    nest: one NumPy-level Gaussian evaluation per mixture, list append, scalar
    log-sum-exp (Clustering.py:755-761 <- util.py:22-31).  Timing use only."""
    d = mean.shape[1]
    vals = []
    for m in range(mean.shape[0]):
        diff = x_t - mean[m]
        diag = var[m]
        head = -d / 2 * LOG_2PI - 0.5 * np.sum(diag)
        tail = -0.5 * np.dot(diff * (1. / diag), diff)
        vals.append(np.log(w[m]) + (head + tail))
    top = np.max(vals)
    if abs(top) == float('inf'):
        return top
    return top + np.log(np.sum(np.exp(vals - top)))


# --------------------------------------------------------------------------
# A5 / A6  unit-HMM emission matrix                   StatisticalModel/LHMM.py:163-187
# --------------------------------------------------------------------------
def unit_observation(x, gmms):
    """cal_observation_pro for one utterance: rows 1..S-2 are the unit's GMM
    states, row 0 is the entry VirtualState(1.) -> ln 1 = 0, row S-1 the exit
    VirtualState(0.) -> -inf (AcousticModel.py:216-219,1039-1043).
    gmms: list of (mean, var, w).  Returns B_p (S,T)."""
    t = x.shape[0]
    rows = [np.zeros(t)]
    for (mean, var, w) in gmms:
        rows.append(gmm_point(x, mean, var, w))
    rows.append(np.full(t, NEG_INF))
    return np.array(rows)


def flat_start_transmat(s=5):
    """init_unit's transition matrix (AcousticModel.py:174-181)."""
    a = np.zeros((s, s))
    a[0, 1] = 1.0
    for j in range(1, s - 1):
        a[j, j] = 0.5
        a[j, j + 1] = 0.5
    return a


# --------------------------------------------------------------------------
# A7  sentence-level ("embedded") HMM                 AcousticModel/AcousticModel.py:957-1014
# --------------------------------------------------------------------------
def embedded(label, unit_trans, unit_b, s=5):
    """label: list of unit names (L); unit_trans: L x (S,S); unit_b: L x (S,T).
    Returns (states dict, A (N,N), B (N,T), pi (N,)), N = (S-2) L + 2."""
    n_units = len(label)
    e = s - 2
    n = e * n_units + 2
    names = [label[0]]
    for u in label:
        names.extend([u] * e)
    names.append(label[-1])
    states = dict(enumerate(names))                                   # :968-976
    a = np.zeros((n, n))
    a[:s - 1, :s] = unit_trans[0][:-1]                                # :981
    for i in range(n_units):
        lo = i * e + 1
        hi = (i + 1) * e + 1
        a[lo:hi, lo - 1:lo - 1 + s] = unit_trans[i][1:-1]             # :982-987
    rows = [unit_b[0][0:-1]]                                          # :993
    for i in range(1, n_units):
        rows.append(unit_b[i][1:-1])                                  # :994-997
    rows.append(unit_b[n_units - 1][-1:])                             # :999-1000
    b = np.concatenate(rows, axis=0)
    pi = np.ones(n) / n                                               # :1005
    return states, a, b, pi


# --------------------------------------------------------------------------
# A8 / A9  forward / backward                         StatisticalModel/LHMM.py:335-366
# --------------------------------------------------------------------------
def forward(a, pi, b):
    """alpha (N,T); alpha_0 = ln pi + B[:,0]; alpha_t(j) = LSE_i[alpha_{t-1}(i)+ln A_ij] + B[j,t]."""
    n, t = b.shape
    with np.errstate(divide='ignore'):
        la = np.log(a)
        lpi = np.log(pi)
    alpha = np.zeros((n, t))
    alpha[:, 0] = lpi + b[:, 0]
    for i in range(1, t):
        alpha[:, i] = lse(alpha[:, i - 1][:, None] + la, axis=0) + b[:, i]
    return alpha


def backward(a, b):
    """beta (N,T); beta_{T-1} = 0 (quirk Q8: zeros from LHMM.py:383, never
    overwritten); beta_t(i) = LSE_j[ln A_ij + B[j,t+1] + beta_{t+1}(j)]."""
    n, t = b.shape
    with np.errstate(divide='ignore'):
        la = np.log(a)
    beta = np.zeros((n, t))
    for i in range(t - 2, -1, -1):
        beta[:, i] = lse(la + b[:, i + 1][None, :] + beta[:, i + 1][None, :], axis=1)
    return beta


def faithful_forward_backward(a, pi, b):
    """forward+backward with the reference's loop nest (one scalar LSE per
    (t, j), LHMM.py:345-351; one row list per t, :360-366).  Timing use only."""
    n, t = b.shape
    with np.errstate(divide='ignore'):
        la = np.log(a)
        lpi = np.log(pi)

    def one(v):
        top = np.max(v)
        if abs(top) == float('inf'):
            return top
        return top + np.log(np.sum(np.exp(v - top)))

    alpha = np.zeros((n, t))
    beta = np.zeros((n, t))
    with np.errstate(all='ignore'):
        alpha[:, 0] = lpi + b[:, 0]
        for i in range(1, t):
            col = []
            for j in range(n):
                col.append(one(alpha[:, i - 1] + la[:, j]))
            alpha[:, i] = np.array(col) + b[:, i]
        for i in range(t - 2, -1, -1):
            col = []
            for j in range(n):
                col.append(one(la[j, :] + b[:, i + 1] + beta[:, i + 1]))
            beta[:, i] = np.array(col)
    return alpha, beta


# --------------------------------------------------------------------------
# A11  xi / gamma / pi                                StatisticalModel/LHMM.py:394-471
# --------------------------------------------------------------------------
def xi_gamma_pi(a, b, alpha, beta):
    """Un-normalised (quirk Q5) log xi (N,N), log gamma (N,), normalised log pi."""
    with np.errstate(divide='ignore', invalid='ignore'):
        la = np.log(a)
        # ((alpha_t(i) + ln a_ij) + B[j,t+1]) + beta_{t+1}(j)   LHMM.py:402-404
        cube = ((alpha[:, :-1].T[:, :, None] + la[None, :, :]) + b[:, 1:].T[:, None, :]) + beta[:, 1:].T[:, None, :]
        if cube.shape[0] == 0:
            ksai = np.full(a.shape, np.nan)
        else:
            ksai = lse(cube, axis=0)                                      # :431-440
        g = alpha[:, :-1] + beta[:, :-1]
        gamma = lse(g, axis=1) if g.shape[1] else np.full(a.shape[0], np.nan)   # :442-445
        p0 = alpha[:, 0] + beta[:, 0]
        log_pi = p0 - lse(p0)                                             # :447-452
    return ksai, gamma, log_pi


# --------------------------------------------------------------------------
# A10  Baum-Welch pass loop                           StatisticalModel/LHMM.py:526-544
# --------------------------------------------------------------------------
def baum_welch(a, pi, b_list, fix_code=0, threshold=0.64, max_pass=1000):
    """The reference's `baulm_welch` for an LHMM built with probmat=b_list
    (embedded HMMs: B is never re-scored, quirk Q6).  Only pi changes between
    passes (when bit0 of fix_code is clear).  Returns a dict with the final
    pass's alpha/beta lists, merged ksai/gamma, pi AFTER the final update,
    the Q trace as logged (first entry -inf) and the pass count."""
    fix_pi = bool(fix_code & 1)
    pi = np.array(pi, dtype=np.float64)
    q = NEG_INF
    trace = []
    n_pass = 0
    while True:
        trace.append(q)
        n_pass += 1
        alphas = [forward(a, pi, b) for b in b_list]                      # :390-392 (uses current pi)
        betas = [backward(a, b) for b in b_list]
        parts = [xi_gamma_pi(a, b, al, be) for b, al, be in zip(b_list, alphas, betas)]
        if len(b_list) > 1:                                               # :454-466
            ksai = matrix_lse([p[0] for p in parts], a.shape[0])
            gamma = matrix_lse([p[1].reshape(1, -1) for p in parts], 1).reshape(-1)
            if not fix_pi:
                pi = np.exp(matrix_lse([p[2].reshape(1, -1) for p in parts], 1)).reshape(-1)
        else:
            ksai, gamma = parts[0][0], parts[0][1]
            if not fix_pi:
                pi = np.exp(parts[0][2])
        q_new = lse(np.concatenate([al[:, -1] for al in alphas]))         # :417-422
        if q_new - q > threshold and n_pass < max_pass:                   # :539
            q = q_new
            continue
        break
    return dict(alpha=alphas, beta=betas, ksai=ksai, gamma=gamma, pi=pi,
                q_trace=np.array(trace), n_pass=n_pass, q_final=q_new,
                logp=[lse(al[:, -1]) for al in alphas])


# --------------------------------------------------------------------------
# A12 / A13  accumulators                              LHMM.py:473-507, :149-161; Clustering.py:653-680
# --------------------------------------------------------------------------
class UnitAcc(object):
    """Per label-position accumulators, log domain, initial -inf
    (LHMM.py:84-85, Clustering.py:96-101)."""

    def __init__(self, s, gmms):
        e = s - 2
        self.ksai_acc = np.full((e, s), NEG_INF)
        self.gamma_acc = np.full((e,), NEG_INF)
        self.gmm = []
        for (mean, var, w) in gmms:
            m, d = mean.shape
            self.gmm.append(dict(acc=np.full((m,), NEG_INF), alpha_acc=NEG_INF,
                                 mean_acc=np.full((m, d), NEG_INF), cov_acc=np.full((m, d), NEG_INF)))


def gmm_update_acc(acc, l_value, b_value, o_value, mean, var, w, bias=100.0):
    """A13 Clustering.GMM.update_acc.  l_value = ln gamma_t(j) (T,), b_value = ln b_j(o_t) (T,),
    o_value (T,D).  `acc` is one entry of UnitAcc.gmm and is updated in place."""
    with np.errstate(all='ignore'):
        rec = gmm_component_loglik(o_value, mean, var, w).T + (l_value - b_value)[None, :]   # (M,T) :660-661
        log_o = np.log(o_value.T + bias)                                                      # (D,T) :663
        acc['acc'] = lse(np.concatenate([rec, acc['acc'][:, None]], axis=1), axis=1)          # :665
        acc['alpha_acc'] = lse(np.append(l_value, acc['alpha_acc']))                          # :667
        m_terms = log_o[None, :, :] + rec[:, None, :]                                          # (M,D,T) :669-672
        acc['mean_acc'] = lse(np.concatenate([m_terms, acc['mean_acc'][:, :, None]], axis=2), axis=2)
        sq = np.log((o_value.T[None, :, :] - mean[:, :, None]) ** 2)                          # (M,D,T) :674-678
        c_terms = rec[:, None, :] + sq
        acc['cov_acc'] = lse(np.concatenate([c_terms, acc['cov_acc'][:, :, None]], axis=2), axis=2)


def update_acc(bw, b_list, data_list, unit_accs, unit_gmms, fix_code=0, s=5):
    """A12 LHMM.update_acc after `baum_welch`.  unit_accs: one UnitAcc per label
    position; unit_gmms: per position the list of (mean,var,w) of its emitting states."""
    fix_a = bool(fix_code & 4)
    fix_pdf = bool(fix_code & 2)
    e = s - 2
    ksai_view = bw['ksai'][1:-1, :]
    gamma_view = bw['gamma'][1:-1]
    for idx in range(len(b_list)):
        if not fix_pdf:
            with np.errstate(all='ignore'):
                l_all = bw['alpha'][idx] + bw['beta'][idx]                 # :486
                b_in = b_list[idx][1:-1, :]
                norm = lse(l_all, axis=0)                                   # :488  per frame over states
                l_in = l_all[1:-1]
        x0 = y0 = 0
        for pos, ua in enumerate(unit_accs):
            if not fix_a:                                                   # :492-496, :149-161
                ua.ksai_acc = logaddexp_q4(ua.ksai_acc, ksai_view[y0:y0 + e, x0:x0 + s])
                ua.gamma_acc = logaddexp_q4(ua.gamma_acc, gamma_view[y0:y0 + e])
            if not fix_pdf:                                                 # :497-505
                with np.errstate(all='ignore'):
                    l_states = l_in[y0:y0 + e, :] - norm[None, :]
                for i in range(e):
                    mean, var, w = unit_gmms[pos][i]
                    gmm_update_acc(ua.gmm[i], l_states[i], b_in[y0 + i], data_list[idx], mean, var, w)
            y0 += e
            x0 += e


# --------------------------------------------------------------------------
# A15  M-step                                          LHMM.py:509-524; Clustering.py:682-693
# --------------------------------------------------------------------------
def hmm_update_param(trans, ksai_acc, gamma_acc):
    """A[1:-1,:] = exp(ksai_acc - gamma_acc[:,None]); rows 0 and S-1 untouched."""
    out = np.array(trans, dtype=np.float64)
    with np.errstate(all='ignore'):
        out[1:-1, :] = np.exp(ksai_acc - gamma_acc.reshape(-1, 1))
    return out


def gmm_update_param(acc, c_covariance=1e-3, bias=100.0):
    """Returns (w, mean, var) from one accumulator dict; variances below
    c_covariance are floored (Clustering.py:688-692)."""
    with np.errstate(all='ignore'):
        w = np.exp(acc['acc'] - acc['alpha_acc'])
        mean = np.exp(acc['mean_acc'] - acc['acc'].reshape(-1, 1)) - bias
        var = np.exp(acc['cov_acc'] - acc['acc'].reshape(-1, 1))
    var = np.where(var < c_covariance, c_covariance, var)
    return w, mean, var


# --------------------------------------------------------------------------
# A14  Viterbi + discriminate                          LHMM.py:546-609; AcousticModel.py:937-955
# --------------------------------------------------------------------------
def viterbi(a, pi, prob, end_state_back=False):
    """Returns (point, path (T,) float64 of state indices).  Adds in the
    reference's order: (p_i + ln A_ij) -> max / first argmax -> + prob[j,t]
    (LHMM.py:577-584); end = first argmax of the final scores (:591-593)."""
    n, t = prob.shape
    with np.errstate(divide='ignore'):
        la = np.log(a)
        p = np.log(pi) + prob[:, 0]
    back = np.zeros((n, t), dtype=np.int64)
    last_j_argmax = 0
    for i in range(1, t):
        tmp = p[:, None] + la                                              # tmp[i_prev, j]
        best = tmp.max(axis=0)
        arg = np.argmax(tmp == best[None, :], axis=0)                      # first index equal to the max
        # a column whose max is NaN never occurs (no inf-inf on this path)
        back[:, i] = arg
        last_j_argmax = int(arg[-1])
        p = best + prob[:, i]
    path = np.zeros(t)
    if end_state_back:
        # quirk Q9 (LHMM.py:586-599): the score/end mark come from the last 4 states,
        # but backtracking starts from the stale `max_index` of the last inner loop.
        tail = p[-4:]
        end = len(p) - 4 + int(np.argmax(tail == tail.max()))
        point = p[end]
        cur = last_j_argmax if t > 1 else 0
    else:
        end = int(np.argmax(p == p.max()))
        point = p[end]
        cur = end
    for i in range(t - 1, -1, -1):
        path[i] = cur
        cur = back[cur, i]
    return float(point), path


def discriminate(unit, sequence):
    """Frame indices where sequence == unit, split into contiguous runs
    (AcousticModel.py:937-955).  Returned in time order."""
    loc = np.where(np.asarray(sequence) == unit)[0]
    if len(loc) == 0:
        return []
    cut = np.where(np.diff(loc) != 1)[0] + 1
    return np.split(loc, cut)


# --------------------------------------------------------------------------
# A16  token-passing max recursion  --  pinned by golden G14   Decoder.py:250-288
# --------------------------------------------------------------------------
def token_viterbi_step(p, la, b_col, first):
    """One frame of Decoder.Token.viterbi: first frame p = ln pi + B[:,0]
    (the caller passes ln pi as `p`), later p_j = max_i(p_i + ln A_ij) + B_j;
    returns (new p, max_j p_j).  Held to the reference's own Token.viterbi
    through decoder_oracle.Token (golden G14)."""
    if first:
        new = p + b_col
    else:
        new = (p[:, None] + la).max(axis=0) + b_col
    return new, float(new.max())


# --------------------------------------------------------------------------
# convenience: the reference's per-utterance E-step / alignment call stacks
# --------------------------------------------------------------------------
def score_label(x, label, model, s=5):
    """multi_embedded_training_1's scoring loop (AcousticModel.py:897-902):
    model[unit] = dict(trans=(S,S), gmms=[(mean,var,w)]*(S-2))."""
    unit_b = [unit_observation(x, model[u]['gmms']) for u in label]
    unit_trans = [model[u]['trans'] for u in label]
    return embedded(label, unit_trans, unit_b, s)


def estep_utterance(x, label, model, fix_code=0, s=5):
    """One utterance of the E-step (SURVEY call stack B).  Returns (bw, accs, sentence HMM)."""
    states, a, b, pi = score_label(x, label, model, s)
    bw = baum_welch(a, pi, [b], fix_code=fix_code)
    accs = [UnitAcc(s, model[u]['gmms']) for u in label]
    update_acc(bw, [b], [x], accs, [model[u]['gmms'] for u in label], fix_code=fix_code, s=s)
    return bw, accs, (states, a, b, pi)


def align_utterance(x, label, model, s=5):
    """One utterance of forced alignment (SURVEY call stack C)."""
    states, a, b, pi = score_label(x, label, model, s)
    point, path = viterbi(a, pi, b)
    names = np.array([states[int(k)] for k in path])
    return point, path, names


# --------------------------------------------------------------------------
# next row f2: what follows forced alignment in training scheme 1
#   AcousticModel.__eq_segment   AcousticModel.py:587-627   (private; pinned through its mangled name, golden G12)
#   AcousticModel.__get_gmmdata  AcousticModel.py:629-644
# --------------------------------------------------------------------------
def eq_segment_e(data, label):
    """mode 'e' (:606-613): an utterance cut into len(label) equal chunks of len(data) // len(label) frames, in label
    order; the remainder frames at the end belong to nobody.  Returns [(unit, block)]."""
    chunk = len(data) // len(label)
    return [(u, data[i * chunk:(i + 1) * chunk]) for i, u in enumerate(label)]


def eq_segment_g(data, n):
    """mode 'g' (:614-625): one block cut into n slices, the first n-1 of len(data) // n frames, the last takes the
    rest (so a block shorter than n gives n-1 empty slices and everything in the last)."""
    chunk = len(data) // n
    out = [data[k * chunk:(k + 1) * chunk] for k in range(n - 1)]
    out.append(data[(n - 1) * chunk:])
    return out


def get_gmmdata(blocks, gmm_num):
    """(:629-644) the k-th slices of all blocks of a unit, concatenated in block order: the data of GMM state k."""
    parts = [eq_segment_g(b, gmm_num) for b in blocks]
    return [np.concatenate([p[k] for p in parts], axis=0) for k in range(gmm_num)]


def regroup_frame_states(unit_seq, gmm_num):
    """Per-frame form of discriminate (:937-955) + __get_gmmdata for one aligned utterance: unit_seq[t] = unit of frame
    t (AcousticModel.viterbi's name sequence).  Returns k[t] = which of the unit's gmm_num GMM states frame t is given
    to.  Runs are maximal blocks of equal unit (the same unit twice in a row is ONE run, as np.where/np.diff see it)."""
    unit_seq = np.asarray(unit_seq)
    t_n = len(unit_seq)
    k = np.zeros(t_n, dtype=np.int64)
    start = 0
    for t in range(1, t_n + 1):
        if t == t_n or unit_seq[t] != unit_seq[start]:
            n = t - start
            chunk = n // gmm_num
            pos = np.arange(n)
            k[start:t] = gmm_num - 1 if chunk == 0 else np.minimum(pos // chunk, gmm_num - 1)
            start = t
    return k

// hmm_dp.hip -- log-domain HMM dynamic programming for gfx950 (MI355X): Baum-Welch forward/backward
// with xi/gamma/pi statistics, and Viterbi, batched one workgroup per sentence HMM.
//
// Replaces:
//   LHMM.__forward_algorithm   StatisticalModel/LHMM.py:335-351   HOT LOOP 2
//   LHMM.__backward_algorithm  StatisticalModel/LHMM.py:353-366   HOT LOOP 3
//   LHMM.__maximization        StatisticalModel/LHMM.py:426-471   HOT LOOP 4 (cal_ksai / cal_gamma / cal_pi)
//   LHMM.__expectation         StatisticalModel/LHMM.py:412-422
//   LHMM.baulm_welch           StatisticalModel/LHMM.py:526-544   (pass loop, quirk Q6)
//   LHMM.update_acc            StatisticalModel/LHMM.py:486-500   (per-frame posteriors l - sum_value)
//   LHMM.viterbi               StatisticalModel/LHMM.py:546-609   HOT LOOP 6
//
// Mapping: the T-long recursion is strictly sequential, so parallelism is utterances x states.
// One workgroup per utterance, lane i <-> state i (one 64-lane wavefront when N <= 64, which is
// the canonical N = 62 sentence HMM).  All state is float64 (alpha/beta reach -2e4,
// f32 would lose the posteriors).  Transitions are held sparse (CSR successors / CSC predecessors
// built on the host from ln A): entries with ln A = -inf contribute exp(-inf) = 0 to every
// log-sum-exp and can never win a max unless everything is -inf, so skipping them is exact;
// a sentence HMM built by AcousticModel.embedded has <= 2 non-zeros per row.  The running
// alpha/beta vectors are exchanged through LDS; emission rows are read time-major (one
// coalesced N-vector per step).
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

constexpr int MAXW = 16;  // waves per workgroup (N <= 1024)

struct Red {
    double buf[2][MAXW];
    int ibuf[2][MAXW];
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide reductions.  `slot` alternates so that consecutive calls need one barrier each.
__device__ __forceinline__ double block_max(double v, Red &red, int &slot) {
    v = wave_max(v);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return v;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red.buf[slot][w] = v;
    __syncthreads();
    double r = red.buf[slot][0];
    for (int k = 1; k < nw; ++k) r = fmax(r, red.buf[slot][k]);
    slot ^= 1;
    return r;
}
__device__ __forceinline__ double block_sum(double v, Red &red, int &slot) {
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return v;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red.buf[slot][w] = v;
    __syncthreads();
    double r = red.buf[slot][0];
    for (int k = 1; k < nw; ++k) r += red.buf[slot][k];
    slot ^= 1;
    return r;
}
// util.log_sum_exp over the block (quirk Q4: returns the max itself when it is +-inf, util.py:62-65)
__device__ __forceinline__ double block_lse(double v, Red &red, int &slot) {
    const double m = block_max(v, red, slot);
    if (isinf(m)) return m;
    const double s = block_sum(exp(v - m), red, slot);
    return m + log(s);
}

// ------------------------------------------------------------------------------------------------
// Baum-Welch pass loop for one utterance per workgroup.
//   KREG > 0: every state has at most KREG predecessors and successors (a sentence HMM built by
//             AcousticModel.embedded has 2): the transition lists and the xi accumulators live in
//             registers and nothing but the emission / alpha rows is read from memory in the loops.
//   KREG = 0: general sparse lists in global memory (dense or wide transition matrices).
// The recursion is a T-long dependent chain with one wave per SIMD at the bench's batch size, so the
// loops are software pipelined: the emission (and alpha) row of step t+1 is requested before step t
// is computed.  Backward exchanges  w_j = b_j(o_{t+1}) + beta_{t+1}(j)  through LDS.
// ------------------------------------------------------------------------------------------------
template <int KREG>
__global__ void hmm_fb_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                              const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                              const double *__restrict__ csr_val, const int *__restrict__ col_ptr,
                              const int *__restrict__ row_idx, const double *__restrict__ csc_val,
                              const double *__restrict__ logpi_in, double *__restrict__ alpha,
                              double *__restrict__ beta, double *__restrict__ lgam, double *__restrict__ xi_m,
                              double *__restrict__ xi_s, double *__restrict__ ksai, double *__restrict__ gamma_out,
                              double *__restrict__ pi_out, double *__restrict__ logp, double *__restrict__ qtrace,
                              int32_t *__restrict__ npass_out, int fix_pi, double threshold, const int *__restrict__ kmax) {
    if (pcl_fb_linear_ok(kmax, blockIdx.x, utts[blockIdx.x].T)) return;      // done by the scaled multi-wave kernels (hmm_fb_linear_mw.inc); block-uniform
    constexpr int KR = KREG > 0 ? KREG : 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Red red;
    const UttDesc d = utts[blockIdx.x];
    const int N = d.N, T = d.T;
    const int NP = blockDim.x;
    double *vec0 = smem;            // ping
    double *vec1 = smem + NP;       // pong
    double *lpi = smem + 2 * NP;    // current ln pi
    const int i = threadIdx.x;
    const bool act = i < N;
    int slot = 0;

    const double *B = Bt + d.b_off;
    double *A_ = alpha + d.b_off;
    double *Bv = beta + d.b_off;
    double *G = lgam + d.b_off;

    // this lane's predecessor (CSC) and successor (CSR) lists
    int pc0 = 0, pc1 = 0, sr0 = 0, sr1 = 0;
    if (act) {
        pc0 = col_ptr[d.ptr_off + i] + d.nnz_off;
        pc1 = col_ptr[d.ptr_off + i + 1] + d.nnz_off;
        sr0 = row_ptr[d.ptr_off + i] + d.nnz_off;
        sr1 = row_ptr[d.ptr_off + i + 1] + d.nnz_off;
    }
    const int npred = pc1 - pc0, nsucc = sr1 - sr0;
    int pidx[KR], sidx[KR];
    double pval[KR], sval[KR], xm[KR], xs[KR];
    if (KREG > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            pidx[k] = (k < npred) ? row_idx[pc0 + k] : 0;
            pval[k] = (k < npred) ? csc_val[pc0 + k] : -INFINITY;
            sidx[k] = (k < nsucc) ? col_idx[sr0 + k] : 0;
            sval[k] = (k < nsucc) ? csr_val[sr0 + k] : -INFINITY;
        }
    }
    lpi[i] = act ? logpi_in[d.vec_off + i] : -INFINITY;
    // dense xi output starts at -inf (LHMM.py:404: ln 0 entries stay -inf)
    for (long long e = i; e < (long long)N * N; e += NP) ksai[d.mat_off + e] = -INFINITY;
    __syncthreads();

    double q = -INFINITY;
    int npass = 0;
    for (;;) {
        // ---------------------------------------------------------------- forward (LHMM.py:335-351)
        double a = -INFINITY;
        double bcur_t = act ? B[i] : 0.0;                               // emission of step t, prefetched
        if (act) {
            a = lpi[i] + bcur_t;
            A_[i] = a;
        }
        vec0[i] = a;
        double bnext_t = (act && T > 1) ? B[(long long)N + i] : 0.0;
        __syncthreads();
        for (int t = 1; t < T; ++t) {
            const double *prev = (t & 1) ? vec0 : vec1;
            double *cur = (t & 1) ? vec1 : vec0;
            bcur_t = bnext_t;
            if (act && t + 1 < T) bnext_t = B[(long long)(t + 1) * N + i];   // in flight during this step
            if (act) {
                double r;
                if (KREG > 0) {
                    double v[KR];
                    double m = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KR; ++k) {
                        v[k] = prev[pidx[k]] + pval[k];
                        m = fmax(m, v[k]);
                    }
                    r = m;
                    if (!isinf(m)) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < KR; ++k) s += exp(v[k] - m);   // padded entries: exp(-inf) = 0
                        r = m + log(s);
                    }
                } else {
                    double m = -INFINITY;
                    for (int k = pc0; k < pc1; ++k) m = fmax(m, prev[row_idx[k]] + csc_val[k]);
                    r = m;
                    if (!isinf(m)) {
                        double s = 0.0;
                        for (int k = pc0; k < pc1; ++k) s += exp(prev[row_idx[k]] + csc_val[k] - m);
                        r = m + log(s);
                    }
                }
                a = r + bcur_t;
                A_[(long long)t * N + i] = a;
            }
            cur[i] = a;
            __syncthreads();
        }
        // Q = LSE_i alpha_{T-1}(i)   (LHMM.py:412-422, datasize == 1 on this path)
        const double qnew = block_lse(act ? a : -INFINITY, red, slot);
        bool final_pass = !(qnew - q > threshold) || (npass + 1 >= PCL_MAX_PASS);   // LHMM.py:539
        if (i == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
        ++npass;
        if (fix_pi && !final_pass && threshold >= 0.0 && npass < PCL_MAX_PASS) {
            // quirk Q6: with pi locked nothing changes between passes, so the next pass would
            // reproduce this one bit for bit and then stop (Q - Q = 0 <= threshold).  Take its
            // statistics now and report the pass the reference would have run.
            if (i == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
            ++npass;
            final_pass = true;
        }

        // ---------------------------------------------------------------- backward (LHMM.py:353-366)
        // beta_{T-1} = 0 for every state (quirk Q8).  The vector exchanged through LDS is
        //   w_j = b_j(o_{t+1}) + beta_{t+1}(j).
        double bcur = 0.0;                 // beta_t(i) of the step just computed
        double gm = -INFINITY, gs = 0.0;   // online LSE for gamma_i over t < T-1 (LHMM.py:442-445)
        if (final_pass && act) {
            if (KREG > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    xm[k] = -INFINITY;
                    xs[k] = 0.0;
                }
            } else {
                for (int k = sr0; k < sr1; ++k) {
                    xi_m[k] = -INFINITY;
                    xi_s[k] = 0.0;
                }
            }
            Bv[(long long)(T - 1) * N + i] = 0.0;
            // l[:,T-1] - sum_value[T-1]  (LHMM.py:486-500); sum_value[T-1] == Q
            G[(long long)(T - 1) * N + i] = a - qnew;
        }
        __syncthreads();   // everyone is done reading the forward vectors
        {
            double *cur = ((T - 1) & 1) ? vec1 : vec0;
            cur[i] = act ? B[(long long)(T - 1) * N + i] + 0.0 : -INFINITY;   // w at t+1 = T-1
        }
        double at_next = (final_pass && act && T > 1) ? A_[(long long)(T - 2) * N + i] : 0.0;   // alpha_t, prefetched
        double b_t = (act && T > 1) ? B[(long long)(T - 2) * N + i] : 0.0;                     // b_i(o_t), for w of the next step
        __syncthreads();
        for (int t = T - 2; t >= 0; --t) {
            const double *nxt = ((t + 1) & 1) ? vec1 : vec0;
            double *cur = (t & 1) ? vec1 : vec0;
            const double at = at_next, bt = b_t;
            if (act && t > 0) {
                b_t = B[(long long)(t - 1) * N + i];
                if (final_pass) at_next = A_[(long long)(t - 1) * N + i];
            }
            double l = -INFINITY;
            if (act) {
                double r;
                if (KREG > 0) {
                    double v[KR];
                    double m = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KR; ++k) {
                        v[k] = sval[k] + nxt[sidx[k]];
                        m = fmax(m, v[k]);
                    }
                    r = m;
                    if (!isinf(m)) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < KR; ++k) s += exp(v[k] - m);
                        r = m + log(s);
                    }
                    if (final_pass) {
                        // xi_ij (+)= alpha_t(i) + ln a_ij + b_j(o_{t+1}) + beta_{t+1}(j)   (LHMM.py:394-405)
#pragma unroll
                        for (int k = 0; k < KR; ++k) {
                            const double x = at + v[k];
                            if (x > xm[k]) {
                                xs[k] = xs[k] * exp(xm[k] - x) + 1.0;   // exp(-inf) = 0 on the first hit
                                xm[k] = x;
                            } else if (x > -INFINITY) {
                                xs[k] += exp(x - xm[k]);
                            }
                        }
                    }
                } else {
                    double m = -INFINITY;
                    for (int k = sr0; k < sr1; ++k) m = fmax(m, csr_val[k] + nxt[col_idx[k]]);
                    r = m;
                    if (!isinf(m)) {
                        double s = 0.0;
                        for (int k = sr0; k < sr1; ++k) s += exp(csr_val[k] + nxt[col_idx[k]] - m);
                        r = m + log(s);
                    }
                    if (final_pass) {
                        for (int k = sr0; k < sr1; ++k) {
                            const double x = at + (csr_val[k] + nxt[col_idx[k]]);
                            const double om = xi_m[k];
                            if (x > om) {
                                xi_s[k] = xi_s[k] * exp(om - x) + 1.0;
                                xi_m[k] = x;
                            } else if (x > -INFINITY) {
                                xi_s[k] += exp(x - om);
                            }
                        }
                    }
                }
                bcur = r;
                if (final_pass) {
                    l = at + r;
                    if (l > gm) {
                        gs = gs * exp(gm - l) + 1.0;
                        gm = l;
                    } else if (l > -INFINITY) {
                        gs += exp(l - gm);
                    }
                    Bv[(long long)t * N + i] = r;
                }
            }
            cur[i] = act ? bt + bcur : -INFINITY;            // w_i for step t-1
            if (final_pass) {
                const double norm = block_lse(l, red, slot);   // sum_value[t] (LHMM.py:488)
                if (act) G[(long long)t * N + i] = l - norm;
            }
            __syncthreads();
        }
        // ---------------------------------------------------------------- pi (LHMM.py:447-452,470-471)
        if (!fix_pi) {
            const double a0 = act ? A_[i] : -INFINITY;
            const double p0 = a0 + ((T > 1) ? bcur : 0.0);
            const double n0 = block_lse(act ? p0 : -INFINITY, red, slot);
            // the reference stores pi = exp(.) and takes np.log of it again on the next pass
            const double pv = exp(p0 - n0);
            __syncthreads();
            lpi[i] = act ? log(pv) : -INFINITY;
            if (final_pass && act) pi_out[d.vec_off + i] = pv;
            __syncthreads();
        } else if (final_pass && act) {
            pi_out[d.vec_off + i] = exp(lpi[i]);
        }
        if (final_pass) {
            if (act) {
                gamma_out[d.vec_off + i] = (gs > 0.0) ? gm + log(gs) : -INFINITY;
                if (KREG > 0) {
#pragma unroll
                    for (int k = 0; k < KR; ++k)
                        if (k < nsucc) ksai[d.mat_off + (long long)i * N + sidx[k]] = (xs[k] > 0.0) ? xm[k] + log(xs[k]) : -INFINITY;
                } else {
                    for (int k = sr0; k < sr1; ++k)
                        ksai[d.mat_off + (long long)i * N + col_idx[k]] = (xi_s[k] > 0.0) ? xi_m[k] + log(xi_s[k]) : -INFINITY;
                }
            }
            if (i == 0) {
                logp[blockIdx.x] = qnew;
                npass_out[blockIdx.x] = npass;
                for (int k = npass; k < PCL_MAX_PASS; ++k) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + k] = NAN;
            }
            break;
        }
        q = qnew;
    }
}

// ------------------------------------------------------------------------------------------------
// The same pass loop for sentence HMMs of at most 64 states with at most two predecessors / successors per state (every HMM
// AcousticModel.embedded builds), on TWO wavefronts: within a pass the forward and the backward recursion do not depend on
// each other (the reference runs them one after the other, LHMM.py:388-392), so wave 0 runs alpha while wave 1 runs beta --
// each keeps its running vector in registers and reads its two neighbours' values with cross-lane reads, no barrier per frame --
// and what needs both (xi, gamma, the per-frame posteriors of the final pass; LHMM.py:394-405,431-445,486-500) is computed
// afterwards in parallel over t by both waves with online log-sum-exps that are merged at the end.  The T-long dependent
// chain is walked 3 times per utterance instead of 6.  log(e^a + e^b) is taken as max + log1p(exp(min - max)): one exp
// and one log1p per state and frame instead of two exps and a log.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lse2(double a, double b) {
    const double m = fmax(a, b);
    if (isinf(m)) return m;                                   // util.log_sum_exp, quirk Q4
    return m + log1p(exp(fmin(a, b) - m));
}
// The same through a table in LDS: the two-term log-sum-exp IS the step of the T-long dependent chain, and exp + log1p in
// float64 are ~140 dependent instructions of it.  f(d) = log1p(exp(-d)), d = |a - b| >= 0, is smooth and every derivative is a
// polynomial in s = 1 / (1 + e^d): f1 = -s, f2 = q = s (1 - s), f3 = -q (1 - 2s), f4 = q (1 - 6q), f5 = -q (1 - 2s)(1 - 12q),
// f6 = q (1 - 30q + 120q^2).  The table holds (f, s) at d_k = k / 32 (host libm, float64); a sixth-order Taylor step of at most
// 1/64 leaves < 1e-17 -- ~25 instructions.  Past d = 40, f < 5e-18 is dropped.
constexpr int SP_N = 1281;
constexpr double SP_MAX = 39.98;
__device__ __forceinline__ double lse2_tab(double a, double b, const double2 *tab) {
    // (branch-free: two conditional branches in every step of a latency-bound chain cost more than the arithmetic they skip)
    const double m = fmax(a, b);
    const bool plain = isinf(m) || !(m - fmin(a, b) < SP_MAX);   // util.log_sum_exp, quirk Q4: the maximum itself; or nothing to add
    const double d = plain ? 0.0 : m - fmin(a, b);
    const int k = __double2int_rn(d * 32.0);
    const double x = fma(-(double)k, 1.0 / 32.0, d);
    const double2 e = tab[k];
    const double sg = e.y, q = fma(-sg, sg, sg), h = fma(-2.0, sg, 1.0), qh = q * h;
    const double c6 = q * fma(q, fma(120.0, q, -30.0), 1.0) * (1.0 / 720.0);
    const double c5 = qh * fma(-12.0, q, 1.0) * (-1.0 / 120.0);
    const double c4 = q * fma(-6.0, q, 1.0) * (1.0 / 24.0);
    const double c3 = qh * (-1.0 / 6.0);
    double r = fma(x, c6, c5);
    r = fma(x, r, c4);
    r = fma(x, r, c3);
    r = fma(x, r, 0.5 * q);
    r = fma(x, r, -sg);
    return plain ? m : m + fma(x, r, e.x);
}
// exp(x) for x <= 0 (every log-sum-exp term is taken relative to a maximum): k = rint(x log2 e), r = x - k ln 2 in two pieces,
// a degree-12 Taylor polynomial on |r| <= ln 2 / 2 (remainder 1.7e-16 relative), v_ldexp_f64 -- 19 instructions where the
// library's takes ~40; below -745 the result is 0, as the library's.
__device__ __forceinline__ double exp_neg(double x) {
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return (x > -745.0) ? ldexp(p, (int)k) : 0.0;
}
__device__ __forceinline__ void online_lse(double x, double &m, double &s) {
    if (x > m) {
        s = s * exp_neg(m - x) + 1.0;                         // exp(-inf) = 0 on the first finite value
        m = x;
    } else if (x > -INFINITY) {
        s += exp_neg(x - m);
    }
}
__device__ __forceinline__ double wave_lse(double v) {
    const double m = wave_max(v);
    if (isinf(m)) return m;
    return m + log(wave_sum(exp_neg(v - m)));
}

#ifndef PCL_FB_AHEAD
#define PCL_FB_AHEAD 4
#endif
constexpr int FB_AHEAD = PCL_FB_AHEAD;      // frames of emissions in flight ahead of the recursion (hmm_fb2_kernel)

// the value of the lane before / after this one: a DPP wave shift, one VALU move per half (lane 0 / 63 keep their own)
__device__ __forceinline__ double lane_before(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);          // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_after(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);          // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void hmm_fb2_body(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                                             const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                             const double *__restrict__ csr_val, const int *__restrict__ col_ptr,
                                             const int *__restrict__ row_idx, const double *__restrict__ csc_val,
                                             const double *__restrict__ logpi_in, double *__restrict__ alpha,
                                             double *__restrict__ beta, double *__restrict__ pi_out, double *__restrict__ logp,
                                             double *__restrict__ qtrace, int32_t *__restrict__ npass_out, int fix_pi,
                                             double threshold, const double2 *__restrict__ softplus) {
#ifndef PCL_FB_NOPRIO
    // a latency-bound chain of few instructions, usually beside the scoring kernel's waves on the same SIMD: issue first
    __builtin_amdgcn_s_setprio(3);
#endif
    __shared__ double a0s[64], b0s[64], lpi[64];
    __shared__ double2 sp[SP_N];
    for (int k = threadIdx.x; k < SP_N; k += 128) sp[k] = softplus[k];
    __shared__ double s_q;
    const UttDesc d = utts[blockIdx.x];
    const int N = d.N, T = d.T;
    const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool act = i < N;
    const double *B = Bt + d.b_off;
    double *A_ = alpha + d.b_off, *Bv = beta + d.b_off;
    int pidx[2] = {0, 0}, sidx[2] = {0, 0};
    double pval[2] = {-INFINITY, -INFINITY}, sval[2] = {-INFINITY, -INFINITY};
    // A sentence HMM is left to right (AcousticModel.embedded, AcousticModel.py:979-989): a state is reached from the state before
    // it and from itself, and reaches itself and the state after it.  Then the neighbour's value comes through a DPP wave shift
    // -- a VALU move -- instead of two ds_bpermute round trips on every step of the chain; any other sparsity keeps the bpermute.
    bool near = true;
    bool pself[2] = {true, true}, sself[2] = {true, true};          // operand k is the lane's own value (else the neighbour's)
    if (act) {
        const int pc0 = col_ptr[d.ptr_off + i] + d.nnz_off, pc1 = col_ptr[d.ptr_off + i + 1] + d.nnz_off;
        const int sr0 = row_ptr[d.ptr_off + i] + d.nnz_off, sr1 = row_ptr[d.ptr_off + i + 1] + d.nnz_off;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < pc1 - pc0) {
                pidx[k] = row_idx[pc0 + k];
                pval[k] = csc_val[pc0 + k];
                pself[k] = pidx[k] == i;
                near = near && (pidx[k] == i || pidx[k] == i - 1);
            }
            if (k < sr1 - sr0) {
                sidx[k] = col_idx[sr0 + k];
                sval[k] = csr_val[sr0 + k];
                sself[k] = sidx[k] == i;
                near = near && (sidx[k] == i || sidx[k] == i + 1);
            }
        }
    }
    const bool shift = __all(near) != 0;                            // (both waves hold the same states: the same answer)
    if (w == 0) lpi[i] = act ? logpi_in[d.vec_off + i] : -INFINITY;
    __syncthreads();
    double q = -INFINITY;
    int npass = 0;
    for (;;) {
        if (w == 0) {
            // ---------------------------------------------------------------- forward (LHMM.py:335-351)
            double a = -INFINITY;
            if (act) {
                a = lpi[i] + B[i];
                A_[i] = a;
            }
            a0s[i] = a;
            // the emissions of FB_AHEAD frames are in flight while FB_AHEAD steps of the chain run: a step is ~300 cycles, an L2 round
            // trip 2-3 times that (one frame ahead, every step waited for its b_j(o_t): ~1000 cycles per step)
            double bq[FB_AHEAD], bn[FB_AHEAD];
#pragma unroll
            for (int k = 0; k < FB_AHEAD; ++k) bq[k] = (act && 1 + k < T) ? B[(long long)(1 + k) * N + i] : 0.0;
            for (int tb = 1; tb < T; tb += FB_AHEAD) {
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) bn[k] = (act && tb + FB_AHEAD + k < T) ? B[(long long)(tb + FB_AHEAD + k) * N + i] : 0.0;
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) {
                    const int t = tb + k;
                    if (t < T) {
                        // the running vector stays in registers: a lane fetches its two predecessors' values with a DPP wave shift
                        // (left-to-right HMMs) or a cross-lane read (ds_bpermute) -- no LDS write, fence and read per step of the chain
                        double p0, p1;
                        if (shift) {
                            const double pb = lane_before(a);
                            p0 = pself[0] ? a : pb;
                            p1 = pself[1] ? a : pb;
                        } else {
                            p0 = __shfl(a, pidx[0], 64);
                            p1 = __shfl(a, pidx[1], 64);
                        }
                        if (act) {
                            a = lse2_tab(p0 + pval[0], p1 + pval[1], sp) + bq[k];
                            A_[(long long)t * N + i] = a;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) bq[k] = bn[k];
            }
            const double qn = wave_lse(act ? a : -INFINITY);                     // Q (LHMM.py:412-422, datasize == 1 on this path)
            if (i == 0) s_q = qn;
        } else {
            // ---------------------------------------------------------------- backward (LHMM.py:353-366); beta_{T-1} = 0 (quirk Q8)
            if (act) Bv[(long long)(T - 1) * N + i] = 0.0;
            double wv = act ? B[(long long)(T - 1) * N + i] + 0.0 : -INFINITY;                // w_j = b_j(o_{t+1}) + beta_{t+1}(j)
            double bcur = 0.0, bq[FB_AHEAD], bn[FB_AHEAD];                                    // (emissions FB_AHEAD frames ahead, as above)
#pragma unroll
            for (int k = 0; k < FB_AHEAD; ++k) bq[k] = (act && T - 2 - k >= 0) ? B[(long long)(T - 2 - k) * N + i] : 0.0;
            for (int tb = T - 2; tb >= 0; tb -= FB_AHEAD) {
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) bn[k] = (act && tb - FB_AHEAD - k >= 0) ? B[(long long)(tb - FB_AHEAD - k) * N + i] : 0.0;
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) {
                    const int t = tb - k;
                    if (t >= 0) {
                        double n0, n1;
                        if (shift) {
                            const double na = lane_after(wv);
                            n0 = sself[0] ? wv : na;
                            n1 = sself[1] ? wv : na;
                        } else {
                            n0 = __shfl(wv, sidx[0], 64);
                            n1 = __shfl(wv, sidx[1], 64);
                        }
                        if (act) {
                            bcur = lse2_tab(sval[0] + n0, sval[1] + n1, sp);
                            Bv[(long long)t * N + i] = bcur;
                        }
                        wv = act ? bq[k] + bcur : -INFINITY;
                    }
                }
#pragma unroll
                for (int k = 0; k < FB_AHEAD; ++k) bq[k] = bn[k];
            }
            b0s[i] = (T > 1) ? bcur : 0.0;
        }
        __syncthreads();
        const double qnew = s_q;
        bool final_pass = !(qnew - q > threshold) || (npass + 1 >= PCL_MAX_PASS);   // LHMM.py:539
        if (threadIdx.x == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
        ++npass;
        if (fix_pi && !final_pass && threshold >= 0.0 && npass < PCL_MAX_PASS) {
            // quirk Q6: with pi locked the next pass would reproduce this one bit for bit and then stop
            if (threadIdx.x == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
            ++npass;
            final_pass = true;
        }
        // ---------------------------------------------------------------- pi (LHMM.py:447-452,470-471), wave 0
        if (w == 0) {
            if (!fix_pi) {
                const double p0 = act ? a0s[i] + b0s[i] : -INFINITY;
                const double n0 = wave_lse(p0);
                const double pv = exp(p0 - n0);                  // the reference stores pi = exp(.) and takes np.log of it again
                lpi[i] = act ? log(pv) : -INFINITY;
                if (final_pass && act) pi_out[d.vec_off + i] = pv;
            } else if (final_pass && act) {
                pi_out[d.vec_off + i] = exp(lpi[i]);
            }
        }
        if (final_pass) {
            // xi / gamma / the per-frame posteriors are parallel over t: hmm_post_kernel, eight waves per utterance, right behind
            if (threadIdx.x == 0) {
                logp[blockIdx.x] = qnew;
                npass_out[blockIdx.x] = npass;
                for (int k = npass; k < PCL_MAX_PASS; ++k) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + k] = NAN;
            }
            break;
        }
        q = qnew;
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void hmm_fb2_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                                                     const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                                     const double *__restrict__ csr_val, const int *__restrict__ col_ptr,
                                                     const int *__restrict__ row_idx, const double *__restrict__ csc_val,
                                                     const double *__restrict__ logpi_in, double *__restrict__ alpha,
                                                     double *__restrict__ beta, double *__restrict__ pi_out, double *__restrict__ logp,
                                                     double *__restrict__ qtrace, int32_t *__restrict__ npass_out, int fix_pi,
                                                     double threshold, const double2 *__restrict__ softplus) {
    hmm_fb2_body(utts, Bt, row_ptr, col_idx, csr_val, col_ptr, row_idx, csc_val, logpi_in, alpha, beta, pi_out, logp, qtrace, npass_out, fix_pi,
                 threshold, softplus);
}

// ------------------------------------------------------------------------------------------------
// Viterbi (LHMM.py:546-609).  Bit-exact in float64: adds in the reference's order
// (p_i + ln A_ij) -> max with first-index tie-break -> + prob[j,t].
// ------------------------------------------------------------------------------------------------
__global__ void hmm_viterbi_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                                   const int *__restrict__ col_ptr, const int *__restrict__ row_idx,
                                   const double *__restrict__ csc_val, const double *__restrict__ logpi,
                                   unsigned short *__restrict__ bp, int32_t *__restrict__ path,
                                   double *__restrict__ point, int end_state_back) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Red red;
    __shared__ int s_end, s_stale;
    const UttDesc d = utts[blockIdx.x];
    const int N = d.N, T = d.T;
    const int NP = blockDim.x;
    double *vec0 = smem, *vec1 = smem + NP;
    const int i = threadIdx.x;
    const bool act = i < N;
    const double *B = Bt + d.b_off;
    unsigned short *BP = bp + d.b_off;
    int pc0 = 0, pc1 = 0;
    if (act) {
        pc0 = col_ptr[d.ptr_off + i] + d.nnz_off;
        pc1 = col_ptr[d.ptr_off + i + 1] + d.nnz_off;
    }
    double p = -INFINITY;
    if (act) p = logpi[d.vec_off + i] + B[i];   // LHMM.py:571
    vec0[i] = p;
    if (i == 0) s_stale = 0;
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const double *prev = (t & 1) ? vec0 : vec1;
        double *cur = (t & 1) ? vec1 : vec0;
        if (act) {
            // max over ALL source states of prev[i'] + ln A[i',j]; non-stored entries are -inf, and
            // the first index equal to the max is index 0 when the max is -inf (LHMM.py:577-583).
            double best = -INFINITY;
            int arg = 0;
            for (int k = pc0; k < pc1; ++k) {
                const double v = prev[row_idx[k]] + csc_val[k];
                if (v > best) {
                    best = v;
                    arg = row_idx[k];
                }
            }
            BP[(long long)t * N + i] = (unsigned short)arg;
            if (i == N - 1 && t == T - 1) s_stale = arg;   // quirk Q9: stale max_index of the last inner loop
            p = best + B[(long long)t * N + i];             // LHMM.py:584
        }
        cur[i] = p;
        __syncthreads();
    }
    // end state: first argmax of the final scores (LHMM.py:591-593) or of the last 4 (:587-589)
    {
        double v = act ? p : -INFINITY;
        if (end_state_back && act && i < N - 4) v = -INFINITY;
        int slot = 0;
        const double m = block_max(v, red, slot);
        // lowest index attaining the max; NaN never occurs on this path
        int cand = (act && v == m && !(end_state_back && i < N - 4)) ? i : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if ((threadIdx.x & 63) == 0) red.ibuf[0][threadIdx.x >> 6] = cand;
        __syncthreads();
        if (i == 0) {
            int e = 0x7fffffff;
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) e = min(e, red.ibuf[0][k]);
            if (e == 0x7fffffff) e = end_state_back ? max(N - 4, 0) : 0;
            // fewer than four states with end_state_back: the reference's len(p_list) - 4 + (first argmax of p_list[-4:]) is a NEGATIVE index
            // there, which NumPy wraps (LHMM.py:587-588) -- only `point` sees it, the backtrack starts from the stale index either way
            if (end_state_back && N < 4) {
                int r = N - 4 + e;
                if (r < 0) r += N;
                e = max(r, 0);
            }
            s_end = e;
        }
        __syncthreads();
    }
    if (i == 0) {
        const double *fin = ((T - 1) & 1) ? vec1 : vec0;
        point[blockIdx.x] = fin[s_end];
        int cur = end_state_back ? s_stale : s_end;
        int32_t *P = path + d.path_off;
        for (int t = T - 1; t >= 0; --t) {       // LHMM.py:596-599
            P[t] = cur;
            cur = (t > 0) ? (int)BP[(long long)t * N + cur] : 0;
        }
    }
}

// The last pass's xi, gamma and per-frame posteriors (LHMM.py:394-405,431-445,486-500) from the alpha / beta the recursion
// kernel left in HBM: nothing here is a chain over t, so eight waves share an utterance's frames (wave w: t = w, w + 8, ...);
// each lane keeps an online (max, sum) pair per stored transition of its state, and the waves' pairs are merged in wave order
// (deterministic, independent of the batch).  Round 2 ran this on the recursion kernel's two waves: 0.23 of the 0.52 ms the
// forward-backward of 128 utterances took.
#ifndef PCL_POST_W
#define PCL_POST_W 8
#endif
constexpr int POST_W = PCL_POST_W;
#ifdef PCL_POST_WAVES
#define PCL_POST_ATTR __attribute__((amdgpu_waves_per_eu(PCL_POST_WAVES, PCL_POST_WAVES)))
#else
#define PCL_POST_ATTR
#endif
// NW waves share the utterance's frames (t = w, w + NW, ...); NW = 1: one wave does everything (the fall-back inside hmm_postl_kernel)
template <int NW>
__device__ __forceinline__ void hmm_post_body(const UttDesc *__restrict__ utts, int u, const double *__restrict__ Bt,
                                              const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                              const double *__restrict__ csr_val, const double *__restrict__ alpha,
                                              const double *__restrict__ beta, double *__restrict__ lgam,
                                              double *__restrict__ ksai, double *__restrict__ gamma_out,
                                              const double *__restrict__ logp) {
    __shared__ double ms[3][NW][2][64];
    const UttDesc d = utts[u];
    const int N = d.N, T = d.T;
    const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool act = i < N;
    const double *B = Bt + d.b_off, *A_ = alpha + d.b_off, *Bv = beta + d.b_off;
    double *G = lgam + d.b_off;
    for (long long e = threadIdx.x; e < (long long)N * N; e += 64 * NW) ksai[d.mat_off + e] = -INFINITY;   // LHMM.py:404: ln 0 entries
    int sidx[2] = {0, 0}, nsucc = 0;
    double sval[2] = {-INFINITY, -INFINITY};
    if (act) {
        const int sr0 = row_ptr[d.ptr_off + i] + d.nnz_off, sr1 = row_ptr[d.ptr_off + i + 1] + d.nnz_off;
        nsucc = sr1 - sr0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < sr1 - sr0) {
                sidx[k] = col_idx[sr0 + k];
                sval[k] = csr_val[sr0 + k];
            }
    }
    const double qnew = logp[u];
    double gm = -INFINITY, gs = 0.0, xm[2] = {-INFINITY, -INFINITY}, xs[2] = {0.0, 0.0};
    // (the loads of the wave's next frame are issued before this one is worked on: each iteration is otherwise a chain of an L2
    //  round trip, two wave reductions and three exponentials)
    auto fetch = [&](int t, double &at, double &bt, double (&nx)[2]) {
        at = -INFINITY;
        bt = 0.0;
        nx[0] = nx[1] = 0.0;
        if (act && t < T) {
            at = A_[(long long)t * N + i];
            bt = Bv[(long long)t * N + i];
            if (t < T - 1) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < nsucc) {
                        const long long o = (long long)(t + 1) * N + sidx[k];
                        nx[k] = B[o] + Bv[o];
                    }
            }
        }
    };
    double at_n, bt_n, nx_n[2];
    fetch(w, at_n, bt_n, nx_n);
    for (int t = w; t < T; t += NW) {
        const double at = at_n, bt = bt_n, nx[2] = {nx_n[0], nx_n[1]};
        fetch(t + NW, at_n, bt_n, nx_n);
        const double l = at + bt;
        // sum_value[t] = LSE_i l[i,t] (LHMM.py:488).  Every one of them is ln P(O) up to rounding, so the log-sum-exp is
        // taken with THAT as its shift -- no maximum over the wave, and the logarithm of a sum within 1e-4 of 1 is three
        // terms of its series; anything else (an impossible utterance: -inf) takes the general path
        double norm;
        const double ssum = wave_sum(act ? exp_neg(l - qnew) : 0.0), u1 = ssum - 1.0;
        if (qnew > -INFINITY && fabs(u1) < 1.0e-4) norm = qnew + u1 * (1.0 - u1 * (0.5 - u1 * (1.0 / 3.0)));
        else norm = wave_lse(act ? l : -INFINITY);
        if (act) G[(long long)t * N + i] = l - norm;                     // l[:,t] - sum_value[t] (:486-500)
        if (act && t < T - 1) {
            online_lse(l, gm, gs);                                       // gamma_i over t < T-1 (:442-445)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k < nsucc)                                           // xi_ij (+)= alpha_t(i) + ln a_ij + b_j(o_{t+1}) + beta_{t+1}(j)   (LHMM.py:394-405)
                    online_lse(at + (sval[k] + nx[k]), xm[k], xs[k]);
        }
    }
    // merge the waves' partial log-sum-exps: (m, s) pairs through LDS, in wave order
    ms[0][w][0][i] = gm; ms[0][w][1][i] = gs;
    ms[1][w][0][i] = xm[0]; ms[1][w][1][i] = xs[0];
    ms[2][w][0][i] = xm[1]; ms[2][w][1][i] = xs[1];
    __syncthreads();
    if (w == 0 && act) {
        double outv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double M = -INFINITY;
#pragma unroll
            for (int v = 0; v < NW; ++v) M = fmax(M, ms[c][v][0][i]);
            double r = -INFINITY;
            if (M > -INFINITY) {
                double tsum = 0.0;
#pragma unroll
                for (int v = 0; v < NW; ++v) {
                    const double mv = ms[c][v][0][i];
                    if (mv > -INFINITY) tsum += ms[c][v][1][i] * exp(mv - M);
                }
                r = M + log(tsum);
            }
            outv[c] = r;
        }
        gamma_out[d.vec_off + i] = outv[0];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < nsucc) ksai[d.mat_off + (long long)i * N + sidx[k]] = outv[1 + k];
    }
}

__global__ __launch_bounds__(64 * POST_W) PCL_POST_ATTR void hmm_post_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                                                             const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                                             const double *__restrict__ csr_val, const double *__restrict__ alpha,
                                                             const double *__restrict__ beta, double *__restrict__ lgam,
                                                             double *__restrict__ ksai, double *__restrict__ gamma_out,
                                                             const double *__restrict__ logp) {
    hmm_post_body<POST_W>(utts, blockIdx.x, Bt, row_ptr, col_idx, csr_val, alpha, beta, lgam, ksai, gamma_out, logp);
}

// The scaled linear-domain forward-backward for left-to-right sentence HMMs (its kernels hand the utterances outside their
// exponent range to the two bodies above)
#include "hmm_fb_linear.inc"
#include "hmm_fb_linear_mw.inc"

// dense ragged (N,N) xi -> values of the stored transitions in CSR (row-major) order
__global__ void ksai_gather_kernel(const UttDesc *__restrict__ utts, const int *__restrict__ row_ptr,
                                   const int *__restrict__ col_idx, const double *__restrict__ ksai, double *__restrict__ dst) {
    const UttDesc d = utts[blockIdx.x];
    for (int i = threadIdx.x; i < d.N; i += blockDim.x)
        for (int k = row_ptr[d.ptr_off + i]; k < row_ptr[d.ptr_off + i + 1]; ++k)
            dst[d.nnz_off + k] = ksai[d.mat_off + (long long)i * d.N + col_idx[d.nnz_off + k]];
}

// the shader clock under load: one wave reads the shader-cycle counter and the constant 100 MHz counter, spins, reads them again
__global__ void clock_probe_kernel(unsigned long long spin_ticks, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c0; out[1] = c1; out[2] = r0; out[3] = r1;
}

}  // namespace

// ---- launchers of the scaled linear-domain forward-backward (hmm_fb_linear.inc)
bool pcl_fb_linear_enabled() {
    const char *v = getenv("PCL_FB_LINEAR");            // read per call: the tests switch between the two paths inside one process
    return !(v && atoi(v) == 0);
}

int pcl_launch_fb_linear(pcl_ctx *ctx, pcl_batch *b, int fix_pi, double threshold) {
    if (!b->Bp) {
        TRY(dev_alloc(ctx, &b->Bp, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->alpha_e, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->beta_e, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->fb_kmax, (size_t)PCL_FB_KREC * b->U));
        TRY(dev_alloc(ctx, &b->fb_dump, (size_t)128));                  // where the posterior kernel's lanes beyond N "store"
    }
    hipLaunchKernelGGL(hmm_emis_pack_kernel, dim3(PACK_BLOCKS, b->U), dim3(256), 0, ctx->stream, b->d_utt, b->Bt, b->Bp, b->fb_kmax, b->row_ptr, b->csr_val,
                       b->logpi);
    const int NPc = (b->Nmax + 63) / 64 * 64;
    if (NPc > 64) {
#define LAUNCH_CM(W)                                                                                                                             \
    hipLaunchKernelGGL((hmm_fblm_kernel<W>), dim3(b->U), dim3(128 * W), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx, b->csr_val, \
                       b->logpi, b->alpha, b->alpha_e, b->beta, b->beta_e, b->pi_out, b->logp, b->qtrace, b->npass, fix_pi, threshold, b->fb_dump)
        if (NPc == 128) LAUNCH_CM(2); else if (NPc == 192) LAUNCH_CM(3); else LAUNCH_CM(4);
#undef LAUNCH_CM
        HIPCHK(ctx, hipGetLastError());
        return PCL_OK;
    }
    hipLaunchKernelGGL(hmm_fbl_kernel, dim3(b->U), dim3(128), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx, b->csr_val,
                       b->logpi, b->alpha, b->alpha_e, b->beta, b->beta_e, b->pi_out, b->logp, b->qtrace, b->npass, fix_pi, threshold, b->Bt, b->col_ptr,
                       b->row_idx, b->csc_val, reinterpret_cast<const double2 *>(ctx->d_softplus), b->fb_dump);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_fb_linear_post(pcl_ctx *ctx, pcl_batch *b) {
    const int NPp = (b->Nmax + 63) / 64 * 64;
    if (!b->fb_part_m) {
        TRY(dev_alloc(ctx, &b->fb_part_m, (size_t)b->U * 3 * POSTL_W * NPp));
        TRY(dev_alloc(ctx, &b->fb_part_e, (size_t)b->U * 3 * POSTL_W * NPp));
    }
    if (NPp > 64) {                                                  // more than one wave of states: hmm_fb_linear_mw.inc
#define LAUNCH_PM(W)                                                                                                                              \
    hipLaunchKernelGGL((hmm_postlm_kernel<W>), dim3(POSTL_W, b->U), dim3(64 * W), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx, \
                       b->csr_val, b->alpha, b->alpha_e, b->beta, b->beta_e, b->lgam, b->ksai, b->logp, b->fb_dump, b->fb_part_m, b->fb_part_e)
        if (NPp == 128) LAUNCH_PM(2); else if (NPp == 192) LAUNCH_PM(3); else LAUNCH_PM(4);
#undef LAUNCH_PM
        hipLaunchKernelGGL(hmm_postlm_merge_kernel, dim3(b->U), dim3(NPp), 0, ctx->stream, b->d_utt, b->fb_kmax, b->row_ptr, b->col_idx, b->fb_part_m,
                           b->fb_part_e, b->ksai, b->gamma_out);
        HIPCHK(ctx, hipGetLastError());
        return PCL_OK;
    }
    hipLaunchKernelGGL(hmm_postl_kernel, dim3(POSTL_W, b->U), dim3(64), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx,
                       b->csr_val, b->alpha, b->alpha_e, b->beta, b->beta_e, b->lgam, b->ksai, b->gamma_out, b->logp, b->Bt, b->fb_dump, b->fb_part_m,
                       b->fb_part_e);
    hipLaunchKernelGGL(hmm_postl_merge_kernel, dim3(b->U), dim3(64), 0, ctx->stream, b->d_utt, b->fb_kmax, b->row_ptr, b->col_idx, b->fb_part_m,
                       b->fb_part_e, b->ksai, b->gamma_out);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_fb_to_log(pcl_ctx *ctx, pcl_batch *b, const double *m, const int *e, double *out) {
    hipLaunchKernelGGL(hmm_to_log_kernel, dim3(8, b->U), dim3(256), 0, ctx->stream, b->d_utt, b->fb_kmax, m, e, out);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_clock_probe(pcl_ctx *ctx, int spin_us, unsigned long long *d_out) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ctx->stream_aux, (unsigned long long)spin_us * 100ULL, d_out);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_ksai_gather(pcl_ctx *ctx, pcl_batch *b, double *dst) {
    hipLaunchKernelGGL(ksai_gather_kernel, dim3(b->U), dim3(64), 0, ctx->stream, b->d_utt, b->row_ptr, b->col_idx, b->ksai, dst);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

namespace {
__global__ void one_frame_kernel(const UttDesc *__restrict__ utts, double *__restrict__ lgam) {
    const UttDesc d = utts[blockIdx.x];
    if (d.T != 1) return;
    for (int i = threadIdx.x; i < d.N; i += blockDim.x) lgam[d.b_off + i] = -INFINITY;
}
}  // namespace

int pcl_launch_forward_backward(pcl_ctx *ctx, pcl_batch *b, int fix_pi, double threshold) {
    const int NP = (b->Nmax + 63) / 64 * 64;
    if (NP > 64 * MAXW) PCL_FAIL(ctx, PCL_ERR_INVALID, "HMM with %d states exceeds the %d-state limit", b->Nmax, 64 * MAXW);
    const size_t shm = (size_t)3 * NP * sizeof(double);
    if (!ctx->d_softplus) {                                      // (f, s) of log1p(exp(-d)) at d = k / 32, host libm
        std::vector<double> tab(2 * SP_N);
        for (int k = 0; k < SP_N; ++k) {
            const double d = k / 32.0;
            tab[2 * k] = log1p(exp(-d));
            tab[2 * k + 1] = 1.0 / (1.0 + exp(d));
        }
        TRY(dev_alloc(ctx, &ctx->d_softplus, tab.size()));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_softplus, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    pcl_timer_begin(ctx, "fb");
    static const bool one_wave = getenv("PCL_FB_ONE_WAVE") && atoi(getenv("PCL_FB_ONE_WAVE")) != 0;      // A/B: the round-1 kernel
    b->fb_linear = false;
    if (b->max_indeg <= 2 && b->max_outdeg <= 2 && NP == 64 && !one_wave) {
        // left-to-right sentence HMMs (everything AcousticModel.embedded builds): the scaled linear-domain chain (hmm_fb_linear.inc),
        // whose kernels run the log-domain bodies for the utterances that do not fit its int32 exponents (usually none); any other
        // structure, or PCL_FB_LINEAR=0: the log-domain kernels
        if (b->left_right && pcl_fb_linear_enabled()) {
            TRY(pcl_launch_fb_linear(ctx, b, fix_pi, threshold));
            TRY(pcl_launch_fb_linear_post(ctx, b));
            b->fb_linear = true;
        } else {
            hipLaunchKernelGGL(hmm_fb2_kernel, dim3(b->U), dim3(128), 0, ctx->stream, b->d_utt, b->Bt, b->row_ptr, b->col_idx,
                               b->csr_val, b->col_ptr, b->row_idx, b->csc_val, b->logpi, b->alpha, b->beta, b->pi_out, b->logp, b->qtrace, b->npass,
                               fix_pi, threshold, reinterpret_cast<const double2 *>(ctx->d_softplus));
            hipLaunchKernelGGL(hmm_post_kernel, dim3(b->U), dim3(64 * POST_W), 0, ctx->stream, b->d_utt, b->Bt, b->row_ptr, b->col_idx, b->csr_val,
                               b->alpha, b->beta, b->lgam, b->ksai, b->gamma_out, b->logp);
        }
    } else if (b->max_indeg <= 2 && b->max_outdeg <= 2) {
        // more than 64 states (a label of 21 units has 65): left-to-right HMMs of up to 256 states run the scaled chain spread over
        // several waves (hmm_fb_linear_mw.inc); the log-domain kernel behind it takes the utterances outside the exponent range
        const bool lin = b->left_right && NP <= 256 && pcl_fb_linear_enabled();
        if (lin) {
            TRY(pcl_launch_fb_linear(ctx, b, fix_pi, threshold));
            TRY(pcl_launch_fb_linear_post(ctx, b));
            b->fb_linear = true;
        }
        hipLaunchKernelGGL(hmm_fb_kernel<2>, dim3(b->U), dim3(NP), shm, ctx->stream, b->d_utt, b->Bt, b->row_ptr, b->col_idx,
                           b->csr_val, b->col_ptr, b->row_idx, b->csc_val, b->logpi, b->alpha, b->beta, b->lgam, b->xi_m,
                           b->xi_s, b->ksai, b->gamma_out, b->pi_out, b->logp, b->qtrace, b->npass, fix_pi, threshold, lin ? b->fb_kmax : nullptr);
    } else
        hipLaunchKernelGGL(hmm_fb_kernel<0>, dim3(b->U), dim3(NP), shm, ctx->stream, b->d_utt, b->Bt, b->row_ptr, b->col_idx,
                           b->csr_val, b->col_ptr, b->row_idx, b->csc_val, b->logpi, b->alpha, b->beta, b->lgam, b->xi_m,
                           b->xi_s, b->ksai, b->gamma_out, b->pi_out, b->logp, b->qtrace, b->npass, fix_pi, threshold, (const int *)nullptr);
    // A one-frame utterance: LHMM.baulm_welch raises on it (the sum over t < T - 1 of LHMM.py:424-445 is empty: ValueError, golden G15), so
    // under the reference's workers it adds nothing to any accumulator.  Here: ln P(O), alpha and beta are what one frame gives, the
    // posteriors are ln 0 -- the accumulate passes (GMM statistics and per-unit transitions) then skip the utterance as a whole.
    if (b->has_one_frame) hipLaunchKernelGGL(one_frame_kernel, dim3(b->U), dim3(64), 0, ctx->stream, b->d_utt, b->lgam);
    pcl_timer_end(ctx, "fb");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// ------------------------------------------------------------------------------------------------
// next row f2: what follows forced alignment in training scheme 1 (the loop at AcousticModel.py:758-764 and
// __get_gmmdata, :629-644), per frame: unit[t] = the unit whose HMM the Viterbi path is in at t; runs = maximal
// blocks of equal unit (AcousticModel.discriminate, :937-955: the same unit twice in a row is ONE run); a run of n
// frames is cut into gmm_num slices, the first gmm_num-1 of n / gmm_num frames, the last takes the rest
// (__eq_segment mode 'g', :614-625); k[t] = the slice of frame t.  One wave per utterance: run starts by a forward
// max-scan over boundary flags, run ends by a backward min-scan, both through LDS.
__global__ void hmm_regroup_kernel(const UttDesc *__restrict__ utts, const int32_t *__restrict__ path,
                                   const int32_t *__restrict__ row_unit, int gmm_num, int32_t *__restrict__ frame_unit,
                                   int32_t *__restrict__ frame_k) {
    extern __shared__ int rg[];                       // [Tmax] units, [Tmax] run starts
    const UttDesc d = utts[blockIdx.x];
    const int T = d.T, lane = threadIdx.x;
    int *un = rg, *st = rg + T;
    for (int t = lane; t < T; t += 64) un[t] = row_unit[d.vec_off + path[d.path_off + t]];
    __syncthreads();
    // forward: start of the run of t
    int carry = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int v = -1;
        if (t < T && (t == 0 || un[t] != un[t - 1])) v = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int w = __shfl_up(v, o, 64);
            if (lane >= o) v = max(v, w);
        }
        v = max(v, carry);
        if (t < T) st[t] = v;
        carry = __shfl(v, 63, 64);
    }
    __syncthreads();
    // backward: end (exclusive) of the run of t, then the slice
    int carry_e = T;
    for (int t0 = (T - 1) / 64 * 64; t0 >= 0; t0 -= 64) {
        const int t = t0 + lane;
        int e = 0x7fffffff;
        if (t < T && (t == T - 1 || un[t + 1] != un[t])) e = t + 1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int w = __shfl_down(e, o, 64);
            if (lane + o < 64) e = min(e, w);
        }
        e = min(e, carry_e);
        if (t < T) {
            const int n = e - st[t], chunk = n / gmm_num, pos = t - st[t];
            frame_unit[d.path_off + t] = un[t];
            frame_k[d.path_off + t] = (chunk == 0) ? gmm_num - 1 : min(pos / chunk, gmm_num - 1);
        }
        carry_e = __shfl(e, 0, 64);
    }
}

int pcl_launch_regroup(pcl_ctx *ctx, pcl_batch *b, const int32_t *d_row_unit, int gmm_num, int32_t *d_frame_unit, int32_t *d_frame_k) {
    const size_t shm = (size_t)2 * b->Tmax * sizeof(int);
    if (shm > 64 * 1024) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_regroup: utterances of %d frames exceed the 8192-frame limit", b->Tmax);
    pcl_timer_begin(ctx, "regroup");
    hipLaunchKernelGGL(hmm_regroup_kernel, dim3(b->U), dim3(64), shm, ctx->stream, b->d_utt, b->path, d_row_unit, gmm_num, d_frame_unit,
                       d_frame_k);
    pcl_timer_end(ctx, "regroup");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_viterbi(pcl_ctx *ctx, pcl_batch *b, int end_state_back) {
    const int NP = (b->Nmax + 63) / 64 * 64;
    if (NP > 64 * MAXW) PCL_FAIL(ctx, PCL_ERR_INVALID, "HMM with %d states exceeds the %d-state limit", b->Nmax, 64 * MAXW);
    if (b->Nmax > 65535) PCL_FAIL(ctx, PCL_ERR_INVALID, "too many states for 16-bit back-pointers");
    const size_t shm = (size_t)2 * NP * sizeof(double);
    pcl_timer_begin(ctx, "viterbi");
    hipLaunchKernelGGL(hmm_viterbi_kernel, dim3(b->U), dim3(NP), shm, ctx->stream, b->d_utt, b->Bt, b->col_ptr,
                       b->row_idx, b->csc_val, b->logpi, b->bp, b->path, b->point, end_state_back);
    pcl_timer_end(ctx, "viterbi");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// hmm_fb_linear.hip -- Baum-Welch forward / backward for left-to-right sentence HMMs in a SCALED LINEAR domain (gfx950).
//
// Replaces the same reference code as hmm_dp.hip's hmm_fb2_kernel + hmm_post_kernel:
//   LHMM.__forward_algorithm   StatisticalModel/LHMM.py:335-351
//   LHMM.__backward_algorithm  StatisticalModel/LHMM.py:353-366
//   LHMM.__maximization        StatisticalModel/LHMM.py:426-471   (xi / gamma / pi)
//   LHMM.__expectation         StatisticalModel/LHMM.py:412-422   (Q)
//   LHMM.baulm_welch           StatisticalModel/LHMM.py:526-544   (pass loop, quirk Q6)
//   LHMM.update_acc            StatisticalModel/LHMM.py:486-500   (per-frame posteriors l - sum_value)
//
// Why.  The reference walks the lattice in the log domain: every step of the T-long dependent chain is a two-term
// log-sum-exp, ~35 dependent float64 instructions + a table read even after round 3's work (~680 cycles per step; it was the
// floor of BASELINE config 2 and the kernel furthest from any roof).  The same numbers are products and sums of
//   alpha_t(j) = (alpha_{t-1}(j) a_jj + alpha_{t-1}(j-1) a_{j-1,j}) b_j(o_t)
// so here every quantity is carried as  value = m * 2^e  with m a float64 and e an int32 kept PER LANE (per state): the
// step of the chain is two v_ldexp_f64 (aligning the two terms to the larger exponent), one add and one multiply, the
// exponents run on the integer pipe ahead of the mantissas, and nothing is ever exponentiated or logged inside the chain:
//   - exp(B_t(j)) is split OFF the chain, for all t in parallel, by hmm_emis_pack_kernel into one 64-bit word per (t, j):
//     a 16-bit power of two (k + 32768; 0 = outside the packed range, 65535 = ln 0) and the 48 leading mantissa bits of
//     2^frac in [1, 2) (rounded to nearest: 2^-49 relative per frame, ~3e-14 after 300 frames);
//   - per-lane exponents cannot underflow, so a state 5000 nats below the best one keeps its exact value, as in the reference's
//     log domain (a per-frame common scale would flush it to zero and turn a finite ln alpha into -inf);
//   - zero (ln 0 = -inf: the exit state's emission row, the entry state after t = 0, absent transitions) lives in the
//     EXPONENT: anything below LE_PZ is zero whatever its mantissa, zero factors add LE_ZADD, sums are floored at LE_FLOOR,
//     so no test of a mantissa sits on the chain;
//   - beta does not depend on pi, the only thing that changes between the passes of baulm_welch for an embedded HMM
//     (quirk Q6): the backward chain is walked ONCE per call (the reference recomputes identical numbers every pass), beside
//     the first forward pass; alpha is written to HBM only in the pass that can be the last one (the third with a free pi,
//     the first with a locked one; a call that ends on another pass repeats that forward pass with the stores on -- same
//     inputs, same bits);
//   - ln alpha / ln beta are what PCL_GET_ALPHA / PCL_GET_BETA return: they are produced on demand by
//     pcl_launch_fb_to_log (nothing on the device reads them in the log domain);
//   - xi, gamma and the per-frame posteriors (hmm_postl_kernel, eight waves per utterance, parallel over t) accumulate
//     m * 2^e pairs per lane and take ONE logarithm per (t, j) -- ln gamma_t(j), which the accumulate kernels and
//     PCL_GET_LGAMMA need in the log domain -- instead of three exponentials.
// Rounding: products and sums of float64 mantissas, ~4 roundings per step: |d ln alpha| ~ 1e-13 after 300 frames, below the
// reference's own log-domain rounding (an ulp of |ln alpha| ~ 2e4 is 3.6e-12 per step); golden G6 holds at 1e-10.
//
// Range.  int32 exponents hold as long as  (max|k_B| + max|k_A| + 4) (T + 2) + max|k_pi| < 2^26  and max|k_B| fits the
// packed word (|ln b| < 22000 nats).  hmm_emis_pack_kernel measures the three maxima per utterance; an utterance outside
// the range (frames 1e6 sigma away, a caller-supplied ln A of -1e9) is left to the log-domain kernels of hmm_dp.hip, which
// are launched right behind and skip every utterance this file handled.  PCL_FB_LINEAR=0 sends everything there (A/B).
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

constexpr double LOG2E = 1.4426950408889634074;
constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// ln(m 2^e); zero (e in the pseudo-zero region) -> -inf
__device__ __forceinline__ double le_log(double m, int e) {
    if (e < LE_PZ || !(m > 0.0)) return -INFINITY;
    return fma((double)e, LN2_HI, fma((double)e, LN2_LO, log(m)));
}
// x = ln v  ->  v = m 2^e with m in [1, 2).  x = -inf -> a zero factor.  |x| log2e < 2^27 is the caller's business.
__device__ __forceinline__ void exp_split(double x, double &m, int &e) {
    if (!(x > -INFINITY)) {
        m = 1.0;
        e = LE_ZADD;
        return;
    }
    const double k = rint(x * LOG2E);
    double r = fma(-k, LN2_HI, x);                // one rounding: the product is exact inside the fma
    r = fma(-k, LN2_LO, r);
    double p = 1.0 / 479001600.0;                 // degree-12 Taylor polynomial on |r| <= ln 2 / 2: 1.7e-16 relative
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
    p = fma(p, r, 1.0);                           // in [0.707, 1.415]
    int ki = (int)k;
    if (p < 1.0) {
        p *= 2.0;
        ki -= 1;
    }
    m = p;
    e = ki;
}
// the packed emission word: [63:52] = field >> 4, [51:4] = 48 leading fraction bits of m in [1, 2), [3:0] = field & 15
__device__ __forceinline__ unsigned long long emis_pack(double m, int k) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    unsigned long long frac = (bits & 0xfffffffffffffULL) + 8ULL;      // round the 4 dropped bits to nearest
    if (frac >> 52) {                                                   // 1.111..1 rounds up to 2.0 = 1.0 * 2^(k+1)
        frac = 0;
        k += 1;
    }
    const unsigned long long field = (unsigned long long)(k + 32768);
    return ((field >> 4) << 52) | (frac & 0xffffffffffff0ULL) | (field & 15ULL);
}
__device__ __forceinline__ void emis_unpack(unsigned long long w, double &m, int &k) {
    const unsigned int hi = (unsigned int)(w >> 32), lo = (unsigned int)w;
    const int field = (int)(((hi >> 20) << 4) | (lo & 15u));
    m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), (int)(lo & 0xfffffff0u));
    k = (field == 65535) ? LE_ZADD : field - 32768;
}
constexpr unsigned long long EMIS_ZERO = (0xfffULL << 52) | 15ULL;     // field 65535: ln 0
constexpr double EMIS_MAX_ABS = 22000.0;                               // |ln b| the 16-bit power of two holds

// DPP wave shifts with an explicit value for the lane that has no source (lane 0 of wave_shr, lane 63 of wave_shl)
__device__ __forceinline__ int shr_i(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int shl_i(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
// mantissas: 0 for the lane without a source (bound_ctrl: no move to preset the destination); a lane switched off by EXEC
// counts as "no source" too, which is what the chains rely on for the state after the last one
__device__ __forceinline__ double shr_d(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ double shl_d(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ int floor_add(int a, int b) { return max(a + b, LE_FLOOR); }

// ------------------------------------------------------------------------------------------------
// exp(B) off the chain: Bt (time-major ln b) -> packed words, and the per-utterance exponent maxima the range test needs:
// kmax[3u] emissions, [3u+1] transitions, [3u+2] the caller's ln pi.  grid (8, U): the 8 blocks of an utterance stride over
// its N*T values; block 0 also looks at the utterance's ln A and ln pi.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int kabs(double x) {          // |rint(x log2e)| saturated; -inf (a zero) does not count
    if (x == -INFINITY) return 0;
    const double a = fabs(x) * LOG2E;
    return (a < 1073741824.0) ? (int)a + 1 : (1 << 30);  // (NaN and +inf saturate too)
}
__global__ __launch_bounds__(256) void hmm_emis_pack_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ Bt,
                                                           unsigned long long *__restrict__ Bp, int *__restrict__ kmax,
                                                           const int *__restrict__ row_ptr, const double *__restrict__ csr_val,
                                                           const double *__restrict__ logpi) {
    __shared__ int red[4];
    const UttDesc d = utts[blockIdx.y];
    const long long n = (long long)d.N * d.T;
    int km = 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += 8 * 256) {
        const double x = Bt[d.b_off + e];
        unsigned long long w;
        if (x == -INFINITY) w = EMIS_ZERO;
        else if (!(fabs(x) < EMIS_MAX_ABS)) {
            w = 0;                                                    // outside the packed range: the utterance takes the log-domain kernels
            km = 1 << 30;
        } else {
            double m;
            int k;
            exp_split(x, m, k);
            w = emis_pack(m, k);
            km = max(km, abs(k) + 1);
        }
        Bp[d.b_off + e] = w;
    }
    km = wave_max_i(km);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = km;
    __syncthreads();
    if (threadIdx.x == 0) {
        km = max(max(red[0], red[1]), max(red[2], red[3]));
        if (km > 0) atomicMax(&kmax[3 * blockIdx.y], km);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        int ka = 0, kp = 0;
        const int nz = row_ptr[d.ptr_off + d.N];
        for (int k = threadIdx.x; k < nz; k += 64) ka = max(ka, kabs(csr_val[d.nnz_off + k]));
        for (int k = threadIdx.x; k < d.N; k += 64) kp = max(kp, kabs(logpi[d.vec_off + k]));
        ka = wave_max_i(ka);
        kp = wave_max_i(kp);
        if (threadIdx.x == 0) {
            kmax[3 * blockIdx.y + 1] = ka;
            kmax[3 * blockIdx.y + 2] = kp;
        }
    }
}

constexpr int FBL_BLK = 4;      // frames of packed emissions in flight ahead of the recursion = steps between renormalisations

// One walk along the chain: `nsteps` steps starting at frame t0 (forward: t0, t0 + 1, ...; backward: t0, t0 - 1, ...).
// State in / out: the lane's two published values (z for itself, p for its neighbour) with their exponents.
//   forward  (FWD):  neighbour = lane before (wave_shr); stored value of a step = alpha_t = s b_t       (STORE only)
//   backward (!FWD): neighbour = lane after  (wave_shl); stored value of a step = beta_t  = s           (always stored)
// (as, ap) = the transitions the published values travel along.  Bl = packed emissions + the lane's column (any valid
// column), bstride = N.  Called with EXEC = the lanes of the utterance's states (one divergent branch around the whole walk):
// the loop body itself has NO branch and every load and store in it is unconditional, so the compiler counts its s_waitcnt
// exactly -- with a predicated store or load inside the loop it falls back to vmcnt(0) in every block, which waits for the
// block's own prefetch and for the stores' acknowledgements (~800 cycles) -- and the row addresses stay scalar.
// Returns the last step's (s, e0).
template <bool FWD, bool STORE>
__device__ __forceinline__ void le_chain(const unsigned long long *__restrict__ Bl, long long bstride, int nsteps, int t0, int T, double as_m,
                                         int as_e, double ap_m, int ap_e, double &z, int &ez, double &p, int &ep, double *__restrict__ sm,
                                         int *__restrict__ se, long long sstride, double &s_out, int &e_out) {
    auto frame = [&](int step) {
        const int t = FWD ? t0 + step : t0 - step;
        return min(max(t, 0), T - 1);
    };
    double s_last = s_out;
    int e_last = e_out;
    auto one = [&](unsigned long long wq, int t) {
        double cb;
        int kb;
        emis_unpack(wq, cb, kb);
        const double cz = cb * as_m, cp = cb * ap_m;
        const int kz = max(kb + as_e, LE_FLOOR), kp = max(kb + ap_e, LE_FLOOR);
        const double pn = FWD ? shr_d(p) : shl_d(p);
        const int epn = FWD ? shr_i(ep, LE_FLOOR) : shl_i(ep, LE_FLOOR);
        const int e0 = max(ez, epn);
        const double s = ldexp(z, ez - e0) + ldexp(pn, epn - e0);
        if (FWD) {
            if (STORE) {
                sm[(long long)t * sstride] = s * cb;
                se[(long long)t * sstride] = floor_add(e0, kb);
            }
        } else {
            sm[(long long)t * sstride] = s;
            se[(long long)t * sstride] = e0;
        }
        s_last = s;
        e_last = e0;
        z = s * cz;
        p = s * cp;
        ez = floor_add(e0, kz);
        ep = floor_add(e0, kp);
    };
    auto renorm = [&]() {                      // one common power of two for the lane's two published values
        const int x = __builtin_amdgcn_frexp_exp(z + p);
        z = ldexp(z, -x);
        p = ldexp(p, -x);
        ez = floor_add(ez, x);
        ep = floor_add(ep, x);
    };
    // two register sets of packed emissions, used alternately (a copy "next -> current" at the end of a block is hoisted by
    // the compiler to right behind the loads, with a wait for them: the prefetch would be waited for inside its own block)
    unsigned long long qa[FBL_BLK], qb[FBL_BLK];
    auto load = [&](unsigned long long (&q)[FBL_BLK], int blk) {
#pragma unroll
        for (int k = 0; k < FBL_BLK; ++k) q[k] = Bl[(long long)frame(blk * FBL_BLK + k) * bstride];
        // the loads stay HERE, ahead of the block's stores: the memory counter retires in order, and loads queued behind the
        // stores could only be waited for together with the stores' acknowledgements
        __builtin_amdgcn_sched_barrier(0);
    };
    auto block = [&](unsigned long long (&q)[FBL_BLK], int blk) {
#pragma unroll
        for (int k = 0; k < FBL_BLK; ++k) one(q[k], FWD ? t0 + blk * FBL_BLK + k : t0 - blk * FBL_BLK - k);
        renorm();
    };
    auto tail = [&](unsigned long long (&q)[FBL_BLK], int blk, int rem) {      // rem < FBL_BLK steps whose frames are in q already
#pragma unroll
        for (int k = 0; k < FBL_BLK - 1; ++k)
            if (k < rem) one(q[k], FWD ? t0 + blk * FBL_BLK + k : t0 - blk * FBL_BLK - k);
        if (rem > 0) renorm();
    };
    const int nblk = nsteps / FBL_BLK, rem = nsteps - nblk * FBL_BLK;
    load(qa, 0);
    int b = 0;
    for (; b + 2 <= nblk; b += 2) {
        load(qb, b + 1);
        block(qa, b);
        load(qa, b + 2);
        block(qb, b + 1);
    }
    if (b < nblk) {
        load(qb, b + 1);
        block(qa, b);
        tail(qb, b + 1, rem);
    } else {
        tail(qa, b, rem);
    }
    s_out = s_last;
    e_out = e_last;
}

// ------------------------------------------------------------------------------------------------
// The pass loop.  128 threads: wave 0 walks alpha (once per pass), wave 1 walks beta (once per call).  Lane i <-> state i.
// A lane PUBLISHES, after every step, its value already multiplied by the transition it will travel along:
//   forward:  z = alpha_t(i) a_ii  (for itself)       p = alpha_t(i) a_{i,i+1}  (for lane i + 1, read through wave_shr)
//   backward: z = w_t(i) a_ii      (for itself)       p = w_t(i) a_{i-1,i}      (for lane i - 1, read through wave_shl)
// with w_t(j) = b_j(o_t) beta_t(j); the multiplications by b_j(o_{t+1}) a.. of the NEXT step are folded into two constants
// per frame that do not depend on the chain.  The loop-carried chain of a step is  mul -> (DPP) -> ldexp -> add.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void hmm_fbl_kernel(const UttDesc *__restrict__ utts, const unsigned long long *__restrict__ Bp,
                                                     const int *__restrict__ kmax, const int *__restrict__ row_ptr,
                                                     const int *__restrict__ col_idx, const double *__restrict__ csr_val,
                                                     const double *__restrict__ logpi_in, double *__restrict__ alpha,
                                                     int *__restrict__ alpha_e, double *__restrict__ beta, int *__restrict__ beta_e,
                                                     double *__restrict__ pi_out, double *__restrict__ logp, double *__restrict__ qtrace,
                                                     int32_t *__restrict__ npass_out, int fix_pi, double threshold) {
    const UttDesc d = utts[blockIdx.x];
    const int N = d.N, T = d.T;
    if (!pcl_fb_linear_ok(kmax, blockIdx.x, T)) return;              // left to hmm_fb2_kernel (block-uniform: no barrier is skipped by part of the block)
    __builtin_amdgcn_s_setprio(3);
    __shared__ double a0s[64], b0s[64], lpi[64];
    __shared__ double s_q;
    const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool act = i < N;
    const unsigned long long *Bl = Bp + d.b_off + i;
    double *Am = alpha + d.b_off + i, *Bm = beta + d.b_off + i;
    int *Ae = alpha_e + d.b_off + i, *Be = beta_e + d.b_off + i;
    const long long sstride = N;
    // the lane's two stored transitions: to itself and to the next state (absent = a zero factor)
    double as_m = 1.0, ao_m = 1.0;
    int as_e = LE_ZADD, ao_e = LE_ZADD;
    if (act) {
        const int sr0 = row_ptr[d.ptr_off + i] + d.nnz_off, sr1 = row_ptr[d.ptr_off + i + 1] + d.nnz_off;
        for (int k = sr0; k < sr1; ++k) {
            const int j = col_idx[k];
            if (j == i) exp_split(csr_val[k], as_m, as_e);
            else if (j == i + 1) exp_split(csr_val[k], ao_m, ao_e);
        }
    }
    // the transition INTO this lane from the one before it (= that lane's a_{i-1,i}): what the backward chain publishes with
    const double an_m = __hiloint2double(shr_i(__double2hiint(ao_m), 0x3ff00000), shr_i(__double2loint(ao_m), 0));
    const int an_e = shr_i(ao_e, LE_ZADD);
    if (w == 0) lpi[i] = act ? logpi_in[d.vec_off + i] : -INFINITY;
    __syncthreads();
    double q = -INFINITY;
    int npass = 0;
    bool beta_done = false;
    for (;;) {
        // alpha goes to HBM only in a pass that can be the last: the first with a locked pi (quirk Q6), the third with a free one
        bool store = (fix_pi != 0) || npass >= 2;
        bool final_pass = false, q6 = false;
        double qnew = 0.0;
        for (;;) {
            if (w == 0) {
                // ---------------------------------------------------------------- forward (LHMM.py:335-351)
                double am = 0.0;                                          // alpha_{T-1}(i)
                int ae = LE_FLOOR;
                a0s[i] = -INFINITY;
                if (act) {                                                // EXEC = the utterance's states for the whole walk
                    double m, pm, bm;
                    int e, pe, bk;
                    exp_split(lpi[i], pm, pe);
                    emis_unpack(Bl[0], bm, bk);
                    m = pm * bm;
                    e = floor_add(pe, bk);
                    {
                        const int x = __builtin_amdgcn_frexp_exp(m);
                        m = __builtin_amdgcn_frexp_mant(m);
                        e = floor_add(e, x);
                    }
                    if (store) {
                        Am[0] = m;
                        Ae[0] = e;
                    }
                    a0s[i] = le_log(m, e);
                    double z = m * as_m, p = m * ao_m;
                    int ez = floor_add(e, as_e), ep = floor_add(e, ao_e);
                    double s_l = 0.0;
                    int e_l = LE_FLOOR;
                    if (store) le_chain<true, true>(Bl, N, T - 1, 1, T, as_m, as_e, ao_m, ao_e, z, ez, p, ep, Am, Ae, sstride, s_l, e_l);
                    else le_chain<true, false>(Bl, N, T - 1, 1, T, as_m, as_e, ao_m, ao_e, z, ez, p, ep, Am, Ae, sstride, s_l, e_l);
                    am = m;
                    ae = e;
                    if (T > 1) {
                        double cb;
                        int kb;
                        emis_unpack(Bl[(long long)(T - 1) * N], cb, kb);
                        am = s_l * cb;
                        ae = floor_add(e_l, kb);
                    }
                }
                // Q = LSE_i alpha_{T-1}(i) (LHMM.py:412-422, datasize == 1 on this path); util.log_sum_exp returns the maximum itself when it is -inf (Q4)
                const int eM = wave_max_i(act ? ae : LE_FLOOR);
                const double S = wave_sum_d(act ? ldexp(am, max(ae - eM, -4000)) : 0.0);
                if (i == 0) s_q = (eM < LE_PZ || !(S > 0.0)) ? -INFINITY : fma((double)eM, LN2_HI, fma((double)eM, LN2_LO, log(S)));
            } else if (!beta_done) {
                // ---------------------------------------------------------------- backward (LHMM.py:353-366); beta_{T-1} = 1 (quirk Q8)
                b0s[i] = -INFINITY;
                if (act) {
                    double bm;
                    int bk;
                    Bm[(long long)(T - 1) * sstride] = 1.0;
                    Be[(long long)(T - 1) * sstride] = 0;
                    emis_unpack(Bl[(long long)(T - 1) * N], bm, bk);
                    // w_{T-1}(i) = b_i(o_{T-1}) * 1
                    double z = bm * as_m, p = bm * an_m;
                    int ez = floor_add(bk, as_e), ep = floor_add(bk, an_e);
                    double s_l = 1.0;
                    int e_l = 0;
                    le_chain<false, true>(Bl, N, T - 1, T - 2, T, as_m, as_e, an_m, an_e, z, ez, p, ep, Bm, Be, sstride, s_l, e_l);
                    b0s[i] = le_log(s_l, e_l);                           // ln beta_0(i) (T == 1: ln 1 = 0)
                }
            }
            beta_done = true;
            __syncthreads();
            qnew = s_q;
            const bool natural = !(qnew - q > threshold) || (npass + 1 >= PCL_MAX_PASS);   // LHMM.py:539
            // quirk Q6: with pi locked the next pass would reproduce this one bit for bit and then stop
            q6 = fix_pi && !natural && threshold >= 0.0 && npass + 1 < PCL_MAX_PASS;
            final_pass = natural || q6;
            if (!final_pass || store) break;
            store = true;                                                // the pass turned out to be the last: once more, with alpha going to HBM
            __syncthreads();
        }
        if (threadIdx.x == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
        ++npass;
        if (q6) {
            if (threadIdx.x == 0) qtrace[(long long)blockIdx.x * PCL_MAX_PASS + npass] = qnew;
            ++npass;
        }
        // ---------------------------------------------------------------- pi (LHMM.py:447-452,470-471), wave 0
        if (w == 0) {
            if (!fix_pi) {
                const double p0 = act ? a0s[i] + b0s[i] : -INFINITY;
                double mx = p0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                const double n0 = isinf(mx) ? mx : mx + log(wave_sum_d(exp(p0 - mx)));
                const double pv = exp(p0 - n0);                  // the reference stores pi = exp(.) and takes np.log of it again
                lpi[i] = act ? log(pv) : -INFINITY;
                if (final_pass && act) pi_out[d.vec_off + i] = pv;
            } else if (final_pass && act) {
                pi_out[d.vec_off + i] = exp(lpi[i]);
            }
        }
        if (final_pass) {
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

// one more term into a lane's running sum  S = sm 2^se
__device__ __forceinline__ void le_accumulate(double g, int ge, double &sm, int &se) {
    const int e1 = max(se, ge);
    sm = ldexp(sm, max(se - e1, -4000)) + ldexp(g, max(ge - e1, -4000));
    se = e1;
}

// ------------------------------------------------------------------------------------------------
// xi, gamma and the per-frame posteriors of the last pass (LHMM.py:394-405,431-445,486-500) from the (m, e) alpha / beta the
// chain kernel left: eight waves per utterance, wave w takes frames w, w + 8, ...; per lane three running sums (gamma_i,
// xi_ii, xi_{i,i+1}) as m 2^e pairs, merged over the waves in wave order (deterministic, independent of the batch).
//   ln gamma_t(i) - sum_value[t]:  sum_value[t] = LSE_i (alpha + beta) is ln P(O) up to rounding, so with
//   ssum = sum_i gamma / 2^eQ and c0 = P(O) / 2^eQ:  sum_value[t] = ln P(O) + ln(ssum / c0), the logarithm of a number within
//   1e-4 of 1 (three terms of its series; anything else takes log()).
// ------------------------------------------------------------------------------------------------
constexpr int POSTL_W = 8;
__global__ __launch_bounds__(64 * POSTL_W) void hmm_postl_kernel(const UttDesc *__restrict__ utts, const unsigned long long *__restrict__ Bp,
                                                                const int *__restrict__ kmax, const int *__restrict__ row_ptr,
                                                                const int *__restrict__ col_idx, const double *__restrict__ csr_val,
                                                                const double *__restrict__ alpha, const int *__restrict__ alpha_e,
                                                                const double *__restrict__ beta, const int *__restrict__ beta_e,
                                                                double *__restrict__ lgam, double *__restrict__ ksai,
                                                                double *__restrict__ gamma_out, const double *__restrict__ logp) {
    const UttDesc d = utts[blockIdx.x];
    const int N = d.N, T = d.T;
    if (!pcl_fb_linear_ok(kmax, blockIdx.x, T)) return;
    __shared__ double msm[3][POSTL_W][64];
    __shared__ int mse[3][POSTL_W][64];
    const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool act = i < N;
    const unsigned long long *B = Bp + d.b_off;
    const double *Am = alpha + d.b_off, *Bm = beta + d.b_off;
    const int *Ae = alpha_e + d.b_off, *Be = beta_e + d.b_off;
    double *G = lgam + d.b_off;
    for (long long e = threadIdx.x; e < (long long)N * N; e += 64 * POSTL_W) ksai[d.mat_off + e] = -INFINITY;   // LHMM.py:404: ln 0 entries
    double as_m = 1.0, ao_m = 1.0;
    int as_e = LE_ZADD, ao_e = LE_ZADD;
    bool has_s = false, has_o = false;
    if (act) {
        const int sr0 = row_ptr[d.ptr_off + i] + d.nnz_off, sr1 = row_ptr[d.ptr_off + i + 1] + d.nnz_off;
        for (int k = sr0; k < sr1; ++k) {
            const int j = col_idx[k];
            if (j == i) { exp_split(csr_val[k], as_m, as_e); has_s = true; }
            else if (j == i + 1) { exp_split(csr_val[k], ao_m, ao_e); has_o = true; }
        }
    }
    const double qnew = logp[blockIdx.x];
    const bool dead = !(qnew > -INFINITY);                           // P(O) = 0: every l is -inf, l - sum_value is NaN as in the reference
    const double kq = rint(qnew * LOG2E);
    const int eQ = dead ? 0 : (int)kq;
    const double lc0 = dead ? 0.0 : fma(-kq, LN2_LO, fma(-kq, LN2_HI, qnew));   // ln c0, c0 = P(O) / 2^eQ in [0.7, 1.42]
    const double rc0 = dead ? 1.0 : exp(-lc0);
    double sm[3] = {0.0, 0.0, 0.0};
    int se[3] = {LE_FLOOR, LE_FLOOR, LE_FLOOR};
    struct Row { double am, bm, bm1; int ae, be, be1; unsigned long long w1; };
    auto fetch = [&](int t, Row &r) {
        r.am = r.bm = r.bm1 = 0.0;
        r.ae = r.be = r.be1 = LE_FLOOR;
        r.w1 = EMIS_ZERO;
        if (act && t < T) {
            const long long o = (long long)t * N + i;
            r.am = Am[o]; r.ae = Ae[o]; r.bm = Bm[o]; r.be = Be[o];
            if (t < T - 1) { r.bm1 = Bm[o + N]; r.be1 = Be[o + N]; r.w1 = B[o + N]; }
        }
    };
    Row nx;
    fetch(w, nx);
    for (int t = w; t < T; t += POSTL_W) {
        const Row r = nx;
        fetch(t + POSTL_W, nx);
        const double g = r.am * r.bm;
        const int ge = floor_add(r.ae, r.be);
        const double ssum = wave_sum_d(act ? ldexp(g, max(ge - eQ, -4000)) : 0.0) * rc0, u1 = ssum - 1.0;
        double corr;                                                     // sum_value[t] - ln P(O)
        if (fabs(u1) < 1.0e-4) corr = u1 * (1.0 - u1 * (0.5 - u1 * (1.0 / 3.0)));
        else corr = log(ssum);
        if (act) {
            double lg;
            if (dead) lg = NAN;
            else if (ge < LE_PZ || !(g > 0.0)) lg = -INFINITY;
            else lg = (fma((double)ge, LN2_HI, fma((double)ge, LN2_LO, log(g))) - qnew) - corr;
            G[(long long)t * N + i] = lg;                                // l[:,t] - sum_value[t] (:486-500)
        }
        if (t < T - 1) {
            double bm1;
            int bk1;
            emis_unpack(r.w1, bm1, bk1);
            const double wm = r.bm1 * bm1;                               // w_{t+1}(i) = b_i(o_{t+1}) beta_{t+1}(i)
            const int we = floor_add(r.be1, bk1);
            const double wn = shl_d(wm);
            const int wen = shl_i(we, LE_FLOOR);
            if (act) {
                le_accumulate(g, ge, sm[0], se[0]);                      // gamma_i over t < T-1 (:442-445)
                // xi_ij (+)= alpha_t(i) a_ij b_j(o_{t+1}) beta_{t+1}(j)   (LHMM.py:394-405)
                le_accumulate(r.am * as_m * wm, floor_add(floor_add(r.ae, as_e), we), sm[1], se[1]);
                le_accumulate(r.am * ao_m * wn, floor_add(floor_add(r.ae, ao_e), wen), sm[2], se[2]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        msm[c][w][i] = sm[c];
        mse[c][w][i] = se[c];
    }
    __syncthreads();
    if (w == 0 && act) {
        double outv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int E = LE_FLOOR;
#pragma unroll
            for (int v = 0; v < POSTL_W; ++v) E = max(E, mse[c][v][i]);
            double tsum = 0.0;
#pragma unroll
            for (int v = 0; v < POSTL_W; ++v) tsum += ldexp(msm[c][v][i], max(mse[c][v][i] - E, -4000));
            outv[c] = le_log(tsum, E);
        }
        gamma_out[d.vec_off + i] = outv[0];
        if (has_s) ksai[d.mat_off + (long long)i * N + i] = outv[1];
        if (has_o) ksai[d.mat_off + (long long)i * N + i + 1] = outv[2];
    }
}

// (m, e) -> ln, for PCL_GET_ALPHA / PCL_GET_BETA; utterances the log-domain kernels handled are copied
__global__ __launch_bounds__(256) void hmm_to_log_kernel(const UttDesc *__restrict__ utts, const int *__restrict__ kmax,
                                                        const double *__restrict__ m, const int *__restrict__ e, double *__restrict__ out) {
    const UttDesc d = utts[blockIdx.y];
    const long long n = (long long)d.N * d.T;
    const bool lin = pcl_fb_linear_ok(kmax, blockIdx.y, d.T);
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += 8 * 256)
        out[d.b_off + k] = lin ? le_log(m[d.b_off + k], e[d.b_off + k]) : m[d.b_off + k];
}

}  // namespace

bool pcl_fb_linear_enabled() {
    const char *v = getenv("PCL_FB_LINEAR");            // read per call: the tests switch between the two paths inside one process
    return !(v && atoi(v) == 0);
}

int pcl_launch_fb_linear(pcl_ctx *ctx, pcl_batch *b, int fix_pi, double threshold) {
    if (!b->Bp) {
        TRY(dev_alloc(ctx, &b->Bp, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->alpha_e, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->beta_e, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->fb_kmax, (size_t)3 * b->U));
    }
    HIPCHK(ctx, hipMemsetAsync(b->fb_kmax, 0, (size_t)3 * b->U * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(hmm_emis_pack_kernel, dim3(8, b->U), dim3(256), 0, ctx->stream, b->d_utt, b->Bt, b->Bp, b->fb_kmax, b->row_ptr, b->csr_val,
                       b->logpi);
    hipLaunchKernelGGL(hmm_fbl_kernel, dim3(b->U), dim3(128), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx, b->csr_val,
                       b->logpi, b->alpha, b->alpha_e, b->beta, b->beta_e, b->pi_out, b->logp, b->qtrace, b->npass, fix_pi, threshold);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_fb_linear_post(pcl_ctx *ctx, pcl_batch *b) {
    hipLaunchKernelGGL(hmm_postl_kernel, dim3(b->U), dim3(64 * POSTL_W), 0, ctx->stream, b->d_utt, b->Bp, b->fb_kmax, b->row_ptr, b->col_idx,
                       b->csr_val, b->alpha, b->alpha_e, b->beta, b->beta_e, b->lgam, b->ksai, b->gamma_out, b->logp);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_fb_to_log(pcl_ctx *ctx, pcl_batch *b, const double *m, const int *e, double *out) {
    hipLaunchKernelGGL(hmm_to_log_kernel, dim3(8, b->U), dim3(256), 0, ctx->stream, b->d_utt, b->fb_kmax, m, e, out);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

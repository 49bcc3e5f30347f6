// model_derive.hip -- device-side model preparation and the GMM M-step (gfx950).
//
//   derive: from the float64 master copy (mean, var, weight) build every layout the kernels read:
//           the VALU scoring rows [s_d c_d ... k2] (f32 + f64), the per-state expansion centres, the f32-input
//           MFMA layout [m-tile][KS4][64][4] of gmm_score_mfma.hip, the two-piece f16 layout [m-tile][2][KS8][64][8]
//           of gmm_score_split.hip / gmm_accumulate_f16.hip with its power-of-two feature scales and K0, and the f32 means.
//   mstep:  Clustering.GMM.update_param (StatisticalModel/Clustering.py:682-693) for every state at once
//           from the resident linear statistics:
//               w = acc / alpha_acc          (= exp(acc - alpha_acc) of the log-domain reference)
//               mu = mean_acc / acc - bias
//               var = max(cov_acc / acc, c_covariance)
//           so that an EM iteration (E-step -> RCCL all-reduce -> M-step -> next E-step) never leaves the GPU.
// Both are HBM-bound streaming passes over J*M*D elements (C4: 240 M elements, a few ms).
#include <algorithm>

#include "pcl_internal.h"

namespace {

constexpr double LOG2E = 1.4426950408889634074, LOG_2PI = 1.8378770664093454836;

__device__ __forceinline__ int ordered_bits(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

// One workgroup per state, one visit: what the layouts need to know about a state as a whole before any of them can be
// written -- its expansion centre c_j[d] = (float) mean_m mu[j,m,d]; the exact power-of-two scale of every feature of the
// split-f16 layout (gmm_score_split.hip): fscale[j][h][d] = 2^e, e = floor(log2 max_m |coef_m,h,d|) (less a shift for states of very unequal variances, below), coef = -log2e/(2 var)
// (h = 0) or log2e (mu - c)/var (h = 1), so that the scaled coefficients fill [1, 2) and the frame features carry the range;
// and K0_j = max_m k'_m (log2 units) of the centred expansion, which the folded-constant layout keeps k'_m relative to
// (k'_m - K0_j in two f16 pieces, K0_j added back in f64).  Two phases over the state's 1.3 MB (the second finds them in L2):
//   (A) centres: thread = (row group r, dimension d), blockDim = D x R threads read R whole rows per step, contiguously;
//   (B) 8 lanes per mixture, 64 mixtures per step, the per-mixture sums in derive_kernel's order (so k'_m has its bits).
// (Round 1 did this with three launches of 39..128 busy threads per state and a write-free pass of derive_kernel: 5.0 ms.)
constexpr int PRE_T = 512;
__global__ __launch_bounds__(PRE_T, 4) void state_prepass_kernel(const double *__restrict__ mean64, const double *__restrict__ var64,
                                                             const double *__restrict__ w64, int M, int Mpad, int D, int Dhost, int KS8,
                                                             int flags, float *__restrict__ centers, float *__restrict__ fscale,
                                                             double *__restrict__ kzero, int j0, float cond_split,
                                                             unsigned char *__restrict__ bad, int *__restrict__ bad_idx, int *__restrict__ nbad,
                                                             int *__restrict__ n_on_pipe, int *__restrict__ good_idx, int *__restrict__ npt) {
    __shared__ double part[64 * 40];
    __shared__ float cen[64];
    __shared__ unsigned long long fbits[2][64];
    __shared__ unsigned int fminb[64];                           // smallest biased exponent of a feature's quadratic coefficient over the on-pipe mixtures
    __shared__ int kbits;
    const int j = j0 + blockIdx.x, tid = threadIdx.x;            // (j0: a state range re-derived on its own, pcl_launch_derive_range)
    const double *mu = mean64 + (size_t)j * Mpad * D, *vr = var64 + (size_t)j * Mpad * D;
    // ---- (A)
    const int R = min(PRE_T / D, 40), d = tid % D, r = tid / D;
    if (r < R) {
        double s = 0.0;
        for (int m = r; m < M; m += R) s += mu[(size_t)m * D + d];
        part[r * 64 + d] = s;
    }
    if (tid < 128) fbits[tid >> 6][tid & 63] = 0ull;
    if (tid < 64) fminb[tid] = 0x7ffu;
    if (tid == 0) kbits = (int)0x80808080;                       // below every real value
    __syncthreads();
    if (tid < D) {
        double s = 0.0;
        for (int q = 0; q < R; ++q) s += part[q * 64 + tid];
        const float c = (float)(s / M);
        centers[(size_t)j * D + tid] = c;
        cen[tid] = c;
    }
    __syncthreads();
    // ---- (B)
    const int ml = tid >> 3, sub = tid & 7;
    double mx0[8], mx1[8];
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 emin[4];                                                 // (exponents only, two per register: eight doubles more cost the kernel a wave per SIMD and 0.7 ms)
#pragma unroll
    for (int k = 0; k < 8; ++k) mx0[k] = mx1[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) emin[k] = us2{0x7ff, 0x7ff};
    float kmax = -INFINITY;
    bool any = false;
    for (int m0 = 0; m0 < M; m0 += PRE_T / 8) {
        const int m = m0 + ml;
        const bool real_m = m < M;
        double sumvar = 0.0, sumlog = 0.0, kq = 0.0;
        double t0[8], t1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dd = sub + 8 * k;
            t0[k] = t1[k] = 0.0;
            if (dd < D && dd < Dhost && real_m) {
                const double v = vr[(size_t)m * D + dd], dm = mu[(size_t)m * D + dd] - (double)cen[dd];
                const double hr = 0.5 / v;                       // ONE float64 division per element (three cost 2 ms per re-derive)
                sumvar += v;
                if (flags & PCL_MODEL_LOGDET) sumlog += log(v);
                kq += dm * dm * hr;
                t0[k] = LOG2E * hr;
                t1[k] = fabs(2.0 * LOG2E * dm * hr);
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            sumvar += __shfl_xor(sumvar, o, 64);
            sumlog += __shfl_xor(sumlog, o, 64);
            kq += __shfl_xor(kq, o, 64);
        }
        // a mixture whose own cancelling term is beyond the expansion's range leaves the matrix-pipe layouts (pcl_internal.h, split
        // states): it must not set the state's feature scales or K0 either, or the mixtures that stay would lose their bits to it
        const bool off_pipe = real_m && (float)(LOG2E * kq) > cond_split;
        if (sub == 0 && real_m) bad[(size_t)j * Mpad + m] = off_pipe ? 1 : 0;
        if (!off_pipe) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mx0[k] = fmax(mx0[k], t0[k]);
                mx1[k] = fmax(mx1[k], t1[k]);
            }
            if (real_m) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned short e0 = t0[2 * k] > 0.0 ? (unsigned short)((__double_as_longlong(t0[2 * k]) >> 52) & 0x7ff) : (unsigned short)0x7ff;
                    const unsigned short e1 = t0[2 * k + 1] > 0.0 ? (unsigned short)((__double_as_longlong(t0[2 * k + 1]) >> 52) & 0x7ff) : (unsigned short)0x7ff;
                    emin[k] = __builtin_elementwise_min(emin[k], us2{e0, e1});
                }
            }
        }
        if (sub == 0 && real_m && !off_pipe) {
            // util.py:29 (quirk Q1): -D/2 ln 2pi - 1/2 sum(var); textbook log-determinant only on request
            const double tail = (flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;
            const double k2 = LOG2E * (log(w64[(size_t)j * Mpad + m]) - 0.5 * Dhost * LOG_2PI - 0.5 * tail);
            if (k2 > -INFINITY) {
                const float kp = (float)(k2 - LOG2E * kq);
                kmax = any ? fmaxf(kmax, kp) : kp;
                any = true;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dd = sub + 8 * k;                              // (non-negative doubles order like their bit patterns)
        if (dd < D) {
            atomicMax(&fbits[0][dd], (unsigned long long)__double_as_longlong(mx0[k]));
            atomicMax(&fbits[1][dd], (unsigned long long)__double_as_longlong(mx1[k]));
            atomicMin(&fminb[dd], (unsigned int)emin[k >> 1][k & 1]);
        }
    }
    if (any) atomicMax(&kbits, ordered_bits(kmax));
    __syncthreads();
    for (int t = tid; t < 2 * KS8 * 8; t += PRE_T) {
        const int h = t / (KS8 * 8), dd = t % (KS8 * 8);
        const double mx = (dd < Dhost && dd < 64) ? __longlong_as_double((long long)fbits[h][dd]) : 0.0;
        int ex = 1;
        if (mx > 0.0 && mx < 1e300) (void)frexp(mx, &ex);     // mx = f 2^ex, f in [0.5, 1)
        // A state whose on-pipe mixtures differ widely in variance: with the LARGEST coefficient of a feature in [1, 2) the wide mixtures'
        // coefficients sink into f16's subnormals (second piece below 2^-14: a' = 2^-13 keeps 12 bits, measured 1.7e-4 nats on a state with
        // variances over four decades, tests/test_gpu_fuzz_estep.py).  f16 has as much room above 2 as below 1, so the scale goes to the
        // middle of the feature's coefficient range (minus one octave: ratios up to 8 keep the round 1-5 scale and bits): a' in
        // [2^-L, 2^L], the term's error ~ 2^(L-25) (term + 1) instead of 2^(2L-25) term; L <= 7 (variance ratios up to 2^14).
        {
            const double mxa = (dd < Dhost && dd < 64) ? __longlong_as_double((long long)fbits[0][dd]) : 0.0;
            const unsigned int emn = (dd < Dhost && dd < 64) ? fminb[dd] : 0x7ffu;
            if (mxa > 0.0 && mxa < 1e300 && emn > 0u && emn < 0x7ffu) {
                int ea;
                (void)frexp(mxa, &ea);
                const int eb = (int)emn - 1022;                  // frexp's convention: value = f 2^e, f in [0.5, 1)
                if (eb <= ea) ex -= min(max((ea - eb) / 2 - 1, 0), 7);
            }
        }
        ex = min(max(ex - 1, -60), 60);
        fscale[((size_t)j * 2 + h) * (KS8 * 8) + dd] = (float)ldexp(1.0, ex);
    }
    if (tid == 0) {
        const int i = kbits;
        const float f = __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
        kzero[j] = (i == (int)0x80808080 || !(f > -3.0e38f)) ? 0.0 : (double)f;      // no real mixture at all: anything
    }
    // the state's off-pipe mixtures as an ascending list (the flags were written by this workgroup before the barriers above)
    {
        __shared__ int cnt[PRE_T];
        const int per = (M + PRE_T - 1) / PRE_T, lo = min(tid * per, M), hi = min(lo + per, M);
        int c = 0;
        for (int m = lo; m < hi; ++m) c += bad[(size_t)j * Mpad + m];
        // exclusive scan of the 512 counts: inside each wave by shuffles, the 8 wave totals through LDS (a serial loop of thread 0 over
        // the 512 entries was 14 us per state: +0.17 ms on the re-derive)
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if ((tid & 63) >= o) inc += up;
        }
        if ((tid & 63) == 63) cnt[tid >> 6] = inc;
        __syncthreads();
        int before = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) before += cnt[wv];
        if (tid == PRE_T - 1) {
            nbad[j] = before + inc;
            n_on_pipe[j] = M - (before + inc);                   // (gmm_score_split.hip: a state without on-pipe mixtures writes -inf and raises no flag)
            npt[j] = (M - (before + inc) + 31) / 32;             // 32-mixture tiles of the matrix-pipe layout in use (round 6: a split state's on-pipe mixtures are compacted to the front)
        }
        int pos = before + inc - c, gpos = lo - pos;             // (the complement list: on-pipe mixtures in ascending order)
        for (int m = lo; m < hi; ++m) {
            if (bad[(size_t)j * Mpad + m]) bad_idx[(size_t)j * Mpad + pos++] = m;
            else good_idx[(size_t)j * Mpad + gpos++] = m;
        }
    }
}

// one workgroup per (state, 32-mixture tile): the tile's mean/var rows are staged in LDS with coalesced
// loads and every output layout is written with contiguous 8/16-byte stores
__global__ __launch_bounds__(256) void derive_kernel(const double *__restrict__ mean64, const double *__restrict__ var64,
                                                     const double *__restrict__ w64, const float *__restrict__ centers, int M,
                                                     int Mpad, int Mpad32, int D, int Dhost, int row, int flags,
                                                     float *__restrict__ params32, double *__restrict__ params64,
                                                     float *__restrict__ mean32, float *__restrict__ pm32,
                                                     uint4 *__restrict__ pm16f,
                                                     const double *__restrict__ kzero,
                                                     const float *__restrict__ fscale, float *__restrict__ cond, int what, int j0,
                                                     const unsigned char *__restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int nmt = Mpad32 / 32, KS = D + 1, KS4 = (KS + 3) / 4;
    const int j = j0 + blockIdx.x / nmt, mt = blockIdx.x % nmt, m0 = mt * 32;
    double *mu = sh, *vr = sh + 32 * D, *k2s = vr + 32 * D, *kqs = k2s + 32;
    float *cen = reinterpret_cast<float *>(kqs + 32);
    float *fa = cen + D, *fb = fa + 32 * D;       // the f32 coefficients of the expansion: a = -log2e/(2 var), b = log2e (mu - c)/var
    __shared__ unsigned char offp[32];            // mixtures of this tile that are off the matrix pipe (split states): absent from pm32 / pm16f
    const int tid = threadIdx.x;
    if (tid < 32) offp[tid] = (m0 + tid < M) ? bad[(size_t)j * Mpad + m0 + tid] : 0;
    for (int e = tid; e < 32 * D; e += 256) {
        const int m = m0 + e / D;
        const bool ok = m < M && (e % D) < Dhost;
        mu[e] = ok ? mean64[((size_t)j * Mpad + m0) * D + e] : 0.0;
        vr[e] = ok ? var64[((size_t)j * Mpad + m0) * D + e] : 1.0;
    }
    for (int d = tid; d < D; d += 256) cen[d] = centers[(size_t)j * D + d];
    __syncthreads();
    {
        // 8 lanes per mixture: per-dimension terms, then a 3-step shuffle reduction
        const int ml = tid >> 3, sub = tid & 7, m = m0 + ml;
        const bool real_m = m < M;
        double sumvar = 0.0, sumlog = 0.0, kq = 0.0;
        for (int dd = sub; dd < D; dd += 8) {
            float a = 0.f, b = 0.f;
            if (dd < Dhost) {
                const double v = vr[ml * D + dd], dm = mu[ml * D + dd] - (double)cen[dd];
                const double hr = 0.5 / v;                       // (one division per element, as in the prepass: the same k'_m)
                sumvar += v;
                if (flags & PCL_MODEL_LOGDET) sumlog += log(v);
                kq += dm * dm * hr;
                if (real_m) {
                    a = (float)(-LOG2E * hr);
                    b = (float)(2.0 * LOG2E * dm * hr);
                }
            }
            fa[ml * D + dd] = a;
            fb[ml * D + dd] = b;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            sumvar += __shfl_xor(sumvar, o, 64);
            sumlog += __shfl_xor(sumlog, o, 64);
            kq += __shfl_xor(kq, o, 64);
        }
        if (sub == 0) {
            double k2 = -INFINITY;
            if (real_m) {
                // util.py:29 (quirk Q1): -D/2 ln 2pi - 1/2 sum(var); textbook log-determinant only on request
                const double tail = (flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;
                k2 = LOG2E * (log(w64[(size_t)j * Mpad + m]) - 0.5 * Dhost * LOG_2PI - 0.5 * tail);
            }
            k2s[ml] = k2;
            kqs[ml] = LOG2E * kq;
            // conditioning of the centred expansion: the largest cancelling term of this state (non-negative floats
            // order like their bit patterns, so an integer atomicMax works)
            if (real_m && (what & PCL_LAYOUT_COND)) {
                // (over ALL mixtures: what the state's worst mixture looks like; with split states the off-pipe ones are counted
                // in nbad and the host decides from that, pcl_state_uses_valu)
                atomicMax(reinterpret_cast<unsigned int *>(cond + j), __float_as_uint((float)(LOG2E * kq)));
                // the 16x16x32 kernel keeps the constant in f16 pieces: a real k' (of a mixture that stays on the pipe) beyond their
                // range sends the state to the direct-form kernels as well
                if (!offp[ml] && (what & PCL_LAYOUT_PM16F) && k2 > -INFINITY && fabs(k2 - LOG2E * kq) > 5.0e4)
                    atomicMax(reinterpret_cast<unsigned int *>(cond + j), __float_as_uint(1.0e30f));
            }
        }
    }
    __syncthreads();
    // VALU scoring rows [s_d c_d ... k2 pad] and the f32 means (only mixtures inside the Mpad grid)
    const bool w32 = what & PCL_LAYOUT_P32, wr64 = what & PCL_LAYOUT_P64;
    if (w32 || wr64)
    for (int e = tid; e < 32 * D; e += 256) {
        const int ml = e / D, d = e - ml * D, m = m0 + ml;
        if (m >= Mpad) continue;
        const bool ok = m < M && d < Dhost;
        const size_t o = ((size_t)j * Mpad + m) * row + 2 * d;
        if (wr64) {                                              // float64 parity rows (derived on first use): exact float64 arithmetic
            const double s = ok ? sqrt(LOG2E / (2.0 * vr[e])) : 0.0, c = ok ? -mu[e] * s : 0.0;
            *reinterpret_cast<double2 *>(params64 + o) = make_double2(s, c);
            if (w32) *reinterpret_cast<float2 *>(params32 + o) = make_float2((float)s, (float)c);
        } else {                                                 // s = sqrt(-a) from the f32 coefficient already there: no float64 sqrt / division
            const float s = ok ? sqrtf(-fa[e]) : 0.f;
            *reinterpret_cast<float2 *>(params32 + o) = make_float2(s, ok ? (float)(-mu[e] * (double)s) : 0.f);
        }
        if (w32) mean32[((size_t)j * Mpad + m) * D + d] = ok ? (float)mu[e] : 0.f;
    }
    if (w32 || wr64)
    for (int e = tid; e < 32 * (row - 2 * D); e += 256) {
        const int ml = e / (row - 2 * D), k = 2 * D + e % (row - 2 * D), m = m0 + ml;
        if (m >= Mpad) continue;
        const double v = (k == 2 * D) ? k2s[ml] : 0.0;
        if (wr64) params64[((size_t)j * Mpad + m) * row + k] = v;
        if (w32) params32[((size_t)j * Mpad + m) * row + k] = (float)v;
    }
    // MFMA layout [KS4][64 lanes][4]: lane = half * 32 + mixture, element = k-step 4q + e
    float4 *pt = reinterpret_cast<float4 *>(pm32) + ((size_t)j * nmt + mt) * (KS4 * 64);
    if (what & PCL_LAYOUT_PM32)
    for (int e = tid; e < KS4 * 64; e += 256) {
        const int q = e >> 6, ln = e & 63, half = ln >> 5, cl = ln & 31;
        const bool off = offp[cl] != 0;
        const bool real_m = (m0 + cl) < M && !off;
        float v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int s = 4 * q + x;
            float val = 0.f;
            if (s < D) {
                val = off ? 0.f : (half ? fb[cl * D + s] : fa[cl * D + s]);
            } else if (s == D) {
                // the constant pair: k' on the low half-wave, 1 in the spare slot (multiplied by 0 in plain
                // scoring, by -ref / cf in the kernels that use the slot)
                val = half ? 1.f : (real_m ? (float)(k2s[cl] - kqs[cl]) : -INFINITY);
            }
            v[x] = val;
        }
        pt[e] = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int KS8f = (D + 7) / 8;
    // folded-constant f16 layout (variant 7): [piece 2][KS8f][64 lanes][8 f16] as above, and in the spare slot d = D:
    // a1: [k1 | 0], a2: [k2 | 1] with k1 + k2 = k'_m - K0_j in f16 pieces (log zero = -6e4)
    if (what & PCL_LAYOUT_PM16F) {
        uint4 *pf = pm16f + ((size_t)j * nmt + mt) * (2 * KS8f * 64);
        const double k0 = kzero[j];
        float *isc = reinterpret_cast<float *>(mu);              // (the float64 tiles are not read any more) 1 / scale, exact: powers of two
        __syncthreads();
        for (int t = tid; t < 2 * KS8f * 8; t += 256) isc[t] = 1.0f / fscale[(size_t)j * 2 * (KS8f * 8) + t];
        __syncthreads();
        for (int e = tid; e < 2 * KS8f * 64; e += 256) {
            const int p = (e >> 6) / KS8f, s = (e >> 6) % KS8f, ln = e & 63, half = ln >> 5, cl = ln & 31;
            const bool off = offp[cl] != 0;
            const bool real_m = (m0 + cl) < M && !off;
            unsigned short h[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int dd = 8 * s + x;
                float val = 0.f;
                bool is_const = false;
                if (dd < D) {
                    val = off ? 0.f : (half ? fb[cl * D + dd] : fa[cl * D + dd]) * isc[half * (KS8f * 8) + dd];
                } else if (dd == D) {
                    is_const = true;
                    if (half == 0) {
                        const double kp = real_m ? k2s[cl] - kqs[cl] - k0 : -INFINITY;
                        val = (kp > -5.0e4) ? (float)kp : -6.0e4f;
                    } else {
                        val = 1.f;
                    }
                }
                const _Float16 h1 = (_Float16)val;
                _Float16 hp;
                if (!is_const) hp = p ? (_Float16)(val - (float)h1) : h1;
                else if (half == 0) hp = p ? ((val <= -6.0e4f) ? (_Float16)0.f : (_Float16)(val - (float)h1)) : h1;    // k1 | k2
                else hp = p ? (_Float16)1.f : (_Float16)0.f;                                                       // a1: 0, a2: 1
                h[x] = __builtin_bit_cast(unsigned short, hp);
            }
            pf[e] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
        }
    }
}

// Clustering.GMM.update_param, one thread per (state, mixture, dim)
__global__ void mstep_kernel(const double *__restrict__ st_acc, const double *__restrict__ st_alpha,
                             const double *__restrict__ st_mean, const double *__restrict__ st_cov, int J, int M, int Mpad,
                             int D, int Dhost, double bias, double floor_var, int j_lo, int j_hi,
                             double *__restrict__ mean64, double *__restrict__ var64, double *__restrict__ w64) {
    const long long first = (long long)j_lo * Mpad * D, total = (long long)j_hi * Mpad * D;   // a rank re-estimates the states it owns
    for (long long gid = first + blockIdx.x * (long long)blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(gid % D);
        const long long jm = gid / D;
        const int m = (int)(jm % Mpad), j = (int)(jm / Mpad);
        if (m >= M || d >= Dhost) continue;
        const double a = st_acc[jm], al = st_alpha[j];
        // A state no frame reached (alpha_acc = 0) keeps its model; a mixture of a seen state whose occupancy is exactly
        // 0 -- every gamma_t(j,m) flushed to zero in the f32 accumulate, or pruned at 2^-150 -- gets weight 0 (= acc /
        // alpha_acc) and keeps its mean and variance: 0/0 would make it NaN, one NaN mixture turns the state's whole
        // log-sum-exp NaN and an all-reduce spreads it to every rank.  (The reference's log-domain accumulators stay
        // finite for such a mixture and give it a vanishing weight, Clustering.py:685.)
        if (!(al > 0.0)) continue;
        if (d == 0) w64[jm] = a / al;                            // Clustering.py:685
        if (!(a > 0.0)) continue;
        mean64[gid] = st_mean[gid] / a - bias;                   // :686
        double c = st_cov[gid] / a;                              // :688
        if (!(c >= floor_var)) c = floor_var;                    // :689-692 (also catches a NaN from upstream)
        var64[gid] = c;
    }
}

// gather the padded device master copy into the caller's dense (J,M,D) arrays
__global__ void pack_kernel(const double *__restrict__ src, int J, int M, int Mpad, int D, int Dhost, double *__restrict__ dst) {
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (gid >= (long long)J * M * Dhost) return;
    const int d = (int)(gid % Dhost);
    const long long jm = gid / Dhost;
    const int j = (int)(jm / M), m = (int)(jm % M);
    dst[gid] = src[((size_t)j * Mpad + m) * D + d];
}

// element-type conversion of the frame matrix: d64 -> f32 (dst32) or f32 -> d64 (dst64)
__global__ void cast_kernel(const double *__restrict__ src64, float *__restrict__ f32, double *__restrict__ dst64, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (dst64) dst64[i] = (double)f32[i];
        else f32[i] = (float)src64[i];
    }
}

}  // namespace

int pcl_launch_cast(pcl_ctx *ctx, const double *src64, float *f32, double *dst64, size_t n) {
    hipLaunchKernelGGL(cast_kernel, dim3(2048), dim3(256), 0, ctx->stream, src64, f32, dst64, n);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// layouts every launch of the active scoring variant reads; the others (f64 parity mode) are derived on first use
// The direct-form f32 rows (params32, mean32: 3 GB at C4, a third of what a re-derive wrote) are read only when WHOLE states are
// scored / accumulated by the direct-form kernels: variant 1, a feature dimension without a matrix-pipe instance, or states whose
// conditioning leaves the centred expansion's range -- they are derived on first use (pcl_ensure_layouts); the fix-up launches of
// the matrix-pipe path make their rows from the master copy (gmm_score.hip MasterModel).  PCL_EAGER_P32=1: as before (A/B).
static int eager_layouts(const pcl_ctx *ctx) {
    int what = PCL_LAYOUT_COND;
    static const bool eager32 = getenv("PCL_EAGER_P32") && atoi(getenv("PCL_EAGER_P32")) != 0;
    if (ctx->score_variant == 1 || !pcl_score_mfma_supported(ctx->D) || eager32) what |= PCL_LAYOUT_P32;
    if (ctx->score_variant == 3) what |= PCL_LAYOUT_PM32;
    if (ctx->score_variant == 7) what |= PCL_LAYOUT_PM16F;
    return what;
}

static int alloc_for(pcl_ctx *ctx, int what) {                 // the buffers of the lazily derived layouts
    const size_t np = (size_t)ctx->J * ctx->Mpad * ctx->row, nm = (size_t)ctx->J * ctx->Mpad * ctx->D;
    if ((what & PCL_LAYOUT_P32) && !ctx->params32) {
        TRY(dev_alloc(ctx, &ctx->params32, np));
        TRY(dev_alloc(ctx, &ctx->mean32, nm));
    }
    if ((what & PCL_LAYOUT_P64) && !ctx->params64) TRY(dev_alloc(ctx, &ctx->params64, np));
    return PCL_OK;
}

static int launch_derive_kernel(pcl_ctx *ctx, int what, int j_lo, int j_hi) {
    TRY(alloc_for(ctx, what));
    const size_t shm = (size_t)(2 * 32 * ctx->D + 64) * sizeof(double) + (size_t)(ctx->D + 2 * 32 * ctx->D) * sizeof(float);
    if (j_hi <= j_lo) return PCL_OK;
    hipLaunchKernelGGL(derive_kernel, dim3((unsigned)((j_hi - j_lo) * (ctx->Mpad32 / 32))), dim3(256), shm, ctx->stream, ctx->mean64, ctx->var64,
                       ctx->w64, ctx->centers32, ctx->M, ctx->Mpad, ctx->Mpad32, ctx->D, ctx->Dhost, ctx->row, ctx->model_flags,
                       ctx->params32, ctx->params64, ctx->mean32, ctx->pm32, reinterpret_cast<uint4 *>(ctx->pm16f),
                       ctx->kzero, ctx->fscale, ctx->d_cond, what, j_lo, ctx->d_bad);
    HIPCHK(ctx, hipGetLastError());
    // split states: their on-pipe mixtures compacted to the front of the state's tiles (gmm_score_coarse.hip), so that the matrix-pipe
    // kernels walk ceil(on-pipe / 32) tiles instead of all of them
    if (what & PCL_LAYOUT_PM16F) TRY(pcl_launch_compact_main(ctx, j_lo, j_hi));
    return PCL_OK;
}

// The eager layouts of the states [j_lo, j_hi) from the master copy, on ctx->stream, nothing waited for: the pipelined
// exchange (pcl_comm.hip) re-derives a state range as soon as its new parameters are there; pcl_derive_finish closes.
int pcl_launch_derive_range(pcl_ctx *ctx, int j_lo, int j_hi) {
    if (j_hi <= j_lo) return PCL_OK;
    const int KS8f = (ctx->D + 7) / 8;
    hipLaunchKernelGGL(state_prepass_kernel, dim3(j_hi - j_lo), dim3(PRE_T), 0, ctx->stream, ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Mpad,
                       ctx->D, ctx->Dhost, KS8f, ctx->model_flags, ctx->centers32, ctx->fscale, ctx->kzero, j_lo, pcl_split_threshold(ctx), ctx->d_bad,
                       ctx->d_bad_idx, ctx->d_nbad, ctx->d_non, ctx->d_good_idx, ctx->d_npt);
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cond + j_lo, 0, (size_t)(j_hi - j_lo) * sizeof(float), ctx->stream));
    return launch_derive_kernel(ctx, eager_layouts(ctx), j_lo, j_hi);
}

int pcl_derive_finish(pcl_ctx *ctx) {
    ctx->layouts_valid = eager_layouts(ctx);
    ctx->cond.resize(ctx->J);
    HIPCHK(ctx, hipMemcpyAsync(ctx->cond.data(), ctx->d_cond, (size_t)ctx->J * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ctx->nbad.resize(ctx->J);
    HIPCHK(ctx, hipMemcpyAsync(ctx->nbad.data(), ctx->d_nbad, (size_t)ctx->J * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ++ctx->model_gen;
    return PCL_OK;
}

int pcl_launch_derive(pcl_ctx *ctx) {
    const int KS8f = (ctx->D + 7) / 8;
    hipLaunchKernelGGL(state_prepass_kernel, dim3(ctx->J), dim3(PRE_T), 0, ctx->stream, ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Mpad,
                       ctx->D, ctx->Dhost, KS8f, ctx->model_flags, ctx->centers32, ctx->fscale, ctx->kzero, 0, pcl_split_threshold(ctx), ctx->d_bad,
                       ctx->d_bad_idx, ctx->d_nbad, ctx->d_non, ctx->d_good_idx, ctx->d_npt);
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cond, 0, (size_t)ctx->J * sizeof(float), ctx->stream));
    const int what = eager_layouts(ctx);
    const int rc = launch_derive_kernel(ctx, what, 0, ctx->J);
    if (rc != PCL_OK) return rc;
    ctx->layouts_valid = what;
    // the per-state conditioning decides which kernel scores a state: bring it to the host (J floats)
    ctx->cond.resize(ctx->J);
    HIPCHK(ctx, hipMemcpyAsync(ctx->cond.data(), ctx->d_cond, (size_t)ctx->J * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ctx->nbad.resize(ctx->J);
    HIPCHK(ctx, hipMemcpyAsync(ctx->nbad.data(), ctx->d_nbad, (size_t)ctx->J * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ++ctx->model_gen;
    return PCL_OK;
}

// derive whatever of `need` the current model does not have yet (same master copy, same centres: no new generation)
int pcl_ensure_layouts(pcl_ctx *ctx, int need) {
    const int missing = need & ~ctx->layouts_valid;
    if (!missing) return PCL_OK;
    const int rc = launch_derive_kernel(ctx, missing, 0, ctx->J);
    if (rc != PCL_OK) return rc;
    ctx->layouts_valid |= missing;
    return PCL_OK;
}

// GMM.update_param for the states [j_lo, j_hi) only (the master copy; the caller re-derives the layouts)
int pcl_launch_mstep_range(pcl_ctx *ctx, double floor_var, int j_lo, int j_hi) {
    if (j_hi <= j_lo) return PCL_OK;
    const long long work = (long long)(j_hi - j_lo) * ctx->Mpad * ctx->D;
    hipLaunchKernelGGL(mstep_kernel, dim3((unsigned)std::min<long long>(4096, (work + 255) / 256)), dim3(256), 0, ctx->stream, ctx->st_acc, ctx->st_alpha, ctx->st_mean,
                       ctx->st_cov, ctx->J, ctx->M, ctx->Mpad, ctx->D, ctx->Dhost, 100.0, floor_var, j_lo, j_hi, ctx->mean64,
                       ctx->var64, ctx->w64);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_mstep(pcl_ctx *ctx, double floor_var) {
    pcl_timer_begin(ctx, "mstep");
    int r = pcl_launch_mstep_range(ctx, floor_var, 0, ctx->J);
    if (r == PCL_OK) r = pcl_launch_derive(ctx);
    pcl_timer_end(ctx, "mstep");
    HIPCHK(ctx, hipGetLastError());
    return r;
}

int pcl_launch_pack(pcl_ctx *ctx, const double *src, int inner, double *dst) {
    // inner = D for (J,M,D) arrays, 1 for (J,M)
    const long long n = (long long)ctx->J * ctx->M * (inner == 1 ? 1 : ctx->Dhost);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src, ctx->J, ctx->M, ctx->Mpad,
                       inner == 1 ? 1 : ctx->D, inner == 1 ? 1 : ctx->Dhost, dst);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// gmm_accumulate.hip -- E-step sufficient statistics of the GMM states for gfx950 (MI355X).
//
// Replaces the reference's HOT LOOP 5:
//   Clustering.GMM.update_acc   StatisticalModel/Clustering.py:653-680
// called from LHMM.update_acc   StatisticalModel/LHMM.py:497-505 with
//   l_value = ln gamma_t(j) = (alpha+beta)[j,t] - LSE_i (alpha+beta)[i,t],  b_value = ln b_j(o_t).
// The reference materialises record (M,T) = ln w_m N_m(o_t) per state and adds (l - b) to get
// ln gamma_t(j,m), then log-sum-exps over t.  Here the component values are RECOMPUTED (the record
// would be 491 KB per frame at M = 2048) and the sums are kept in the linear domain:
//   acc[j,m]        = sum_t g            g = gamma_t(j,m) = exp(ln w_m N_m(o_t) + l_t - b_t)
//   mean_acc[j,m,d] = sum_t g (o_td + bias)
//   cov_acc[j,m,d]  = sum_t g (o_td - mu_jmd)^2
//   alpha_acc[j]    = sum_t gamma_t(j)
//
// Mapping (MI355X-first):
//   1. compaction.  In a left-right sentence HMM almost every (frame, state) pair has a posterior
//      that underflows: gamma_t(j,m) <= gamma_t(j), so when ln gamma_t(j) < UNDERFLOW every g is
//      exactly 0 in the kernel's arithmetic and the frame can be dropped without changing a bit.
//      Three small kernels (count per segment -> exclusive scan -> ordered fill) build, per state,
//      the list of surviving frames in a fixed order, so sums are reproducible run to run.
//   2. accumulate.  lanes = mixtures: a lane owns one mixture of one state and keeps its 2D+1
//      running sums and its 2D+1 scoring parameters in VGPRs for the whole pass over the state's
//      frame list -- there is no cross-lane reduction and no atomic.  Frames are staged through LDS
//      in chunks and read back as broadcasts.  Per (frame, mixture, dim): y = x s + c; q += y^2
//      (2 FMA), then z = g y; S1 += z; S2 += z y (3 ops).  Centred, scaled moments avoid the
//      cancellation of raw moments: (o - mu) = y / s, (o - mu)^2 = y^2 / s^2.
#include <stdlib.h>

#include <algorithm>

#include "pcl_internal.h"

namespace {

constexpr int WG = 256;
constexpr int FC = 32;  // frames per LDS chunk

// one wave per segment: number of frames whose posterior survives
__global__ void acc_count_kernel(const ScoreSeg *__restrict__ segs, int n_segs, const double *__restrict__ lgam,
                                 double thr, int *__restrict__ cnt) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n_segs) return;
    const int lane = threadIdx.x & 63;
    const ScoreSeg sg = segs[k];
    int c = 0;
    for (int t = lane; t < sg.len; t += 64) c += lgam[sg.out0 + (long long)t * sg.out_stride] >= thr;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) cnt[k] = c;
}

// single-workgroup exclusive scan over the (state-sorted) segments; off[n] = total
__global__ void acc_scan_kernel(const int *__restrict__ cnt, int n, long long *__restrict__ off) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (n + nt - 1) / nt;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < nt; ++i) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        off[n] = run;
    }
    __syncthreads();
    long long run = part[tid];
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += cnt[i];
    }
}

// one wave per segment: ordered compaction of the surviving frames
__global__ void acc_fill_kernel(const ScoreSeg *__restrict__ segs, int n_segs, const double *__restrict__ lgam,
                                const double *__restrict__ Bt, double thr, const long long *__restrict__ off,
                                ActiveFrame *__restrict__ list) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n_segs) return;
    const int lane = threadIdx.x & 63;
    const ScoreSeg sg = segs[k];
    long long pos = off[k];
    for (int t0 = 0; t0 < sg.len; t0 += 64) {
        const int t = t0 + lane;
        double lg = -INFINITY, lb = 0.0;
        if (t < sg.len) {
            lg = lgam[sg.out0 + (long long)t * sg.out_stride];
            lb = Bt[sg.out0 + (long long)t * sg.out_stride];
        }
        const bool keep = (t < sg.len) && (lg >= thr);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            ActiveFrame a;
            a.frame = sg.frame0 + t;
            a.coef = lg - lb;
            a.lg = exp(lg);
            list[pos + rank] = a;
        }
        pos += __popcll(mask);
    }
}

// The same two steps with the posteriors read the way they lie: ln gamma and ln b are time-major per utterance (element (t, row) at
// b_off + t N + row), so a wave per SEGMENT (one row, all t) takes one 8-byte value from each 64-byte line it touches and every line is
// fetched by 8 different waves (acc_count 0.30 + acc_fill 0.67 ms per pass at the C4 shard).  Here a wave owns 8 ROWS of one utterance:
// lane = 8 tt + r reads row r of frame 8 i + tt, eight lanes share a line; the per-row counts and ranks come from the ballot restricted to
// the row's lanes (0x0101..01 << r), lower lanes = earlier frames, so the lists are the ones the segment kernels build, entry for entry.
constexpr int ROWS_WPB = 4;   // waves (groups of 8 rows) per block
__global__ void acc_count_rows_kernel(const UttDesc *__restrict__ utt, const int *__restrict__ seg_of_row, const double *__restrict__ lgam,
                                      double thr, int *__restrict__ cnt) {
    const UttDesc d = utt[blockIdx.y];
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * ROWS_WPB + (threadIdx.x >> 6)) * 8;
    if (row0 >= d.N) return;
    const int r = lane & 7, tt = lane >> 3, row = row0 + r;
    const int seg = row < d.N ? seg_of_row[d.vec_off + row] : -1;
    int c = 0;
    if (seg >= 0)
        for (int t = tt; t < d.T; t += 8) c += lgam[d.b_off + (long long)t * d.N + row] >= thr;
    c += __shfl_xor(c, 8, 64);
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    if (tt == 0 && seg >= 0) cnt[seg] = c;
}

__global__ void acc_fill_rows_kernel(const UttDesc *__restrict__ utt, const int *__restrict__ seg_of_row, const double *__restrict__ lgam,
                                     const double *__restrict__ Bt, double thr, const long long *__restrict__ off, ActiveFrame *__restrict__ list) {
    const UttDesc d = utt[blockIdx.y];
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * ROWS_WPB + (threadIdx.x >> 6)) * 8;
    if (row0 >= d.N) return;
    const int r = lane & 7, tt = lane >> 3, row = row0 + r;
    const int seg = row < d.N ? seg_of_row[d.vec_off + row] : -1;
    long long pos = seg >= 0 ? off[seg] : 0;
    const unsigned long long rowmask = 0x0101010101010101ull << r, below = (1ull << lane) - 1ull;
    for (int t0 = 0; t0 < d.T; t0 += 8) {                       // (the trip count is uniform over the wave: every lane takes part in the ballot)
        const int t = t0 + tt;
        const bool in = seg >= 0 && t < d.T;
        double lg = -INFINITY, lb = 0.0;
        if (in) {
            const long long i = d.b_off + (long long)t * d.N + row;
            lg = lgam[i];
            lb = Bt[i];
        }
        const bool keep = in && lg >= thr;
        const unsigned long long mine = __ballot(keep) & rowmask;
        if (keep) {
            ActiveFrame a;
            a.frame = d.frame0 + t;
            a.coef = lg - lb;
            a.lg = exp(lg);
            list[pos + __popcll(mine & below)] = a;
        }
        pos += __popcll(mine);
    }
}

template <typename real>
struct Fast;
template <>
struct Fast<float> {
    static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};
template <>
struct Fast<double> {
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
};

// MASTER = true (the masked fix-up of the matrix-pipe pass): a lane makes its mixture's row [s_d c_d ... k2] and its mean from the
// float64 master copy with derive_kernel's arithmetic, so params32 / mean32 need not exist for it (see gmm_score.hip, MasterModel).
// Split states (pcl_internal.h): the masked fix-up leaves the off-pipe mixtures (`bad`) out, as the matrix-pipe pass did; the SUBSET launch
// (bad_idx != NULL) gives a lane the idx-th off-pipe mixture of its state, walks ALL the state's frames and leaves alpha_acc alone.
struct AccMaster {
    const double *mean64, *var64, *w64;
    int M, Dhost, flags;
    const unsigned char *bad;
    const int *bad_idx, *nbad;
};

// grid = (mixture slices, states with work).  A lane owns mixture `m` of state `j`.
template <int D, typename real, int MINW, bool MASTER = false>
__global__ __launch_bounds__(WG, MINW) void gmm_accumulate_kernel(
    const real *__restrict__ frames, const real *__restrict__ params, const real *__restrict__ means, int Mpad,
    const int *__restrict__ work_states, const int *__restrict__ seg_lo, const int *__restrict__ seg_hi,
    const long long *__restrict__ off, const ActiveFrame *__restrict__ list, double bias, double *__restrict__ st_acc,
    double *__restrict__ st_alpha, double *__restrict__ st_mean, double *__restrict__ st_cov,
    const int *__restrict__ tile_off, const unsigned int *__restrict__ tile_mask, const int *__restrict__ state_flag, AccMaster mm) {
    // tile_mask != NULL (fix-up of the f16 producer / consumer path, gmm_accumulate_f16.hip): only the frames whose bit is set
    // in their 32-frame tile's mask are accumulated -- the ones that path took out because a scaled feature left the f16
    // range -- and alpha_acc is left alone (the consumer summed gamma_t(j) of every frame).
    constexpr int ROW = (2 * D + 1 + 3) / 4 * 4;
    constexpr int XS = (D + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) real xs[FC * XS];
    __shared__ real cf[FC];
    __shared__ double red[WG / 64];

    // subset launch: grid = (states, slices) -- with the slice in x most slices are empty (a state has ~100 off-pipe mixtures of the 1024
    // allowed) and the workgroups that do have work sit on block indices 0, 4, 8, ...: two of the eight XCDs (measured: 8.7 ms against 2.2)
    const bool subset = MASTER && mm.bad_idx != nullptr;
    const int w = subset ? blockIdx.x : blockIdx.y, slice = subset ? blockIdx.y : blockIdx.x;
    if (state_flag && !state_flag[w]) return;                            // fix-up mode: nothing of this state was left out
    const int j = work_states[w];
    const long long beg = off[seg_lo[w]], end = off[seg_hi[w]];
    if (beg == end) return;
    int m = slice * WG + threadIdx.x;
    bool live = m < Mpad;
    if (MASTER && subset) {
        if (slice * WG >= mm.nbad[j]) return;                             // (uniform: no lane of this slice has a mixture)
        live = m < mm.nbad[j];
        m = live ? mm.bad_idx[(size_t)j * Mpad + m] : 0;
    }
    const real *p = MASTER ? nullptr : params + ((size_t)j * Mpad + (live ? m : 0)) * ROW;
    const size_t jm_ld = (size_t)j * Mpad + (live ? m : 0);

    // f32: the 2D scoring parameters stay in VGPRs for the whole pass.  f64 (parity mode) re-reads them
    // (L1-resident, 632 B per lane): with them resident the kernel needs > 256 VGPRs per lane and hipcc's
    // AGPR spill code for 64-bit values returned doubles with damaged low words (2^-20 relative errors).
    constexpr bool PREG = sizeof(real) == 4;
    real s[PREG ? D : 1], c[PREG ? D : 1], S1[D], S2[D];
    static_assert(!MASTER || PREG, "the on-the-fly rows are the f32 fix-up's");
    real k2;
    if (MASTER) {
        constexpr double L2E = 1.4426950408889634074, LOG_2PI = 1.8378770664093454836;
        const bool real_m = live && m < mm.M && (subset || !(mm.bad && mm.bad[jm_ld]));
        double tail = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            real sv = 0, cv = 0;
            if (real_m && d < mm.Dhost) {
                const double v = mm.var64[jm_ld * D + d], mu = mm.mean64[jm_ld * D + d];
                const float a = (float)(-L2E * (0.5 / v));
                const float sf = sqrtf(-a);
                sv = (real)sf;
                cv = (real)(float)(-mu * (double)sf);
                tail += (mm.flags & PCL_MODEL_LOGDET) ? log(v) : v;
            }
            s[PREG ? d : 0] = sv;
            c[PREG ? d : 0] = cv;
            S1[d] = 0;
            S2[d] = 0;
        }
        k2 = real_m ? (real)(float)(L2E * (log(mm.w64[jm_ld]) - 0.5 * mm.Dhost * LOG_2PI - 0.5 * tail)) : (real)-INFINITY;
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (PREG) {
                s[d] = p[2 * d];
                c[d] = p[2 * d + 1];
            }
            S1[d] = 0;
            S2[d] = 0;
        }
        k2 = p[2 * D];
    }
    real S0 = 0;
    double galpha = 0.0;  // slice 0 only: sum_t gamma_t(j)
    constexpr double LOG2E = 1.4426950408889634074;

    for (long long f0 = beg; f0 < end; f0 += FC) {
        const int nf = (int)min((long long)FC, end - f0);
        unsigned int fmask = 0xffffffffu;
        if (tile_mask) {
            fmask = tile_mask[tile_off[w] + (int)((f0 - beg) / FC)];
            if (fmask == 0u) continue;                                   // (uniform over the workgroup)
        }
        __syncthreads();
        // stage the chunk: FC x D features (rows gathered by index) and the per-frame coefficient
        for (int e = threadIdx.x; e < nf * D; e += WG) {
            const int f = e / D, d = e - f * D;
            xs[f * XS + d] = frames[list[f0 + f].frame * D + d];
        }
        if ((int)threadIdx.x < nf) {
            const ActiveFrame a = list[f0 + threadIdx.x];
            cf[threadIdx.x] = ((fmask >> threadIdx.x) & 1u) ? (real)(a.coef * LOG2E) : (real)-INFINITY;
            if (slice == 0 && !tile_mask && !subset) galpha += a.lg;
        }
        __syncthreads();
        for (int f = 0; f < nf; ++f) {
            const real *x = &xs[f * XS];
            real y[D];
            real q = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                y[d] = PREG ? Fast<real>::fma(x[d], s[d], c[d]) : Fast<real>::fma(x[d], p[2 * d], p[2 * d + 1]);
                q = Fast<real>::fma(y[d], y[d], q);
            }
            const real g = Fast<real>::exp2((k2 - q) + cf[f]);   // gamma_t(j,m)  (Clustering.py:660-661)
            S0 += g;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const real z = g * y[d];
                S1[d] += z;
                S2[d] = Fast<real>::fma(z, y[d], S2[d]);
            }
        }
    }
    // (round 6) a mixture that received nothing from this batch -- every posterior exactly 0 in f32: S0 is a sum of non-negative terms --
    // adds +0 to 2 D + 1 float64 sums: skipped, the same bits.  On the models EM leaves most mixtures own a handful of frames of the whole
    // corpus and see none of them in most batches, and this flush (79 lane-strided float64 read-modify-writes per mixture, 7.7 GB per pass at
    // C4) was most of the kernel: 14.1 ms per batch whatever the arithmetic did (profiles/r06_accumulate_direct.txt).
    if (live && S0 != (real)0) {
        const size_t jm = (size_t)j * Mpad + m;
        const double a0 = (double)S0;
        st_acc[jm] += a0;                                                   // Clustering.py:665
        const real *mu = MASTER ? nullptr : means + jm * D;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double sd = (double)(PREG ? s[d] : p[2 * d]);
            if (sd > 0.0) {
                // sum g (o + bias) = sum g (o - mu) + (mu + bias) sum g    (Clustering.py:669-672)
                const double mud = MASTER ? (double)(float)mm.mean64[jm * D + d] : (double)mu[d];      // (mean32 holds the f32 rounding)
                st_mean[jm * D + d] += (double)S1[d] / sd + (mud + bias) * a0;
                st_cov[jm * D + d] += (double)S2[d] / (sd * sd);            // Clustering.py:674-678
            }
        }
    }
    if (slice == 0 && !tile_mask && !subset) {
        // deterministic block sum of the per-thread partial posteriors
        double v = galpha;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < WG / 64; ++k) t += red[k];
            st_alpha[j] += t;                                               // Clustering.py:667
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the accumulate pass (f32).  Both halves of the E-step statistics are dense contractions:
//   (1) v[f,m] + cf[f]  = Xe[f,:] . P[:,m]        (the scoring GEMM of gmm_score_mfma.hip, K = 2D+2, with the
//                                                  per-frame coefficient cf riding in the spare K slot)
//   (2) S[m,:]         += sum_f g[f,m] Xe[f,:]     g = exp2(v + cf)  -- raw moments about the state centre:
//                                                  columns [x'^2_d, x'_d | 1] give S2, S1, S0
// Orientation: MFMA (1) is computed as D1[frame rows][mixture cols], so a lane holds, for ITS mixture
// column, 16 frame rows per register set; register r of D1, used directly as the A operand of MFMA (2),
// supplies the k-pair (frame row(r), frame row(r)+4) with the mixture on the lane -- no data movement
// between the two products.  The frame tile Xe (32 frames x 80 features) is staged once in LDS by the
// workgroup and read by MFMA (1) frame-major and by MFMA (2) feature-major (stride 97: conflict free).
// A wave owns one 32-mixture tile of one state: its 40 parameter registers and its 48 moment
// accumulators stay resident while the workgroup walks the state's frame list.
// Flush: cov = S2 - 2 d S1 + d^2 S0, mean = S1 + (c + bias) S0 with d = mu - c, in float64.
// ------------------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4x __attribute__((ext_vector_type(4)));
#ifndef PCL_ACC_T16
#define PCL_ACC_T16 1      // 1: 16x16x4 MFMA tiles (80 moment columns), 0: 32x32x2 (96 with padding)
#endif
constexpr int AW = 8;        // waves (32-mixture tiles) per workgroup
constexpr int XSTR = PCL_ACC_T16 ? 98 : 97;   // LDS row stride of the frame tile (floats): odd multiples avoid bank conflicts of the frame-major reads

template <int D, bool T16>
__global__ __launch_bounds__(AW * 64, 2) void gmm_accumulate_mfma_kernel(
    const float *__restrict__ frames, const float *__restrict__ pm, const float *__restrict__ centers,
    const double *__restrict__ means64, int M, int Mpad, int n_mtiles, int n_states, const int *__restrict__ work_states,
    const int *__restrict__ seg_lo, const int *__restrict__ seg_hi, const long long *__restrict__ off,
    const ActiveFrame *__restrict__ list, double bias, double *__restrict__ st_acc, double *__restrict__ st_alpha,
    double *__restrict__ st_mean, double *__restrict__ st_cov) {
    constexpr int KS = D + 1, KS4 = (KS + 3) / 4;
    constexpr int NCT = (2 * D + 1 + 31) / 32;   // 32-column tiles covering the 2D feature columns + the constant
    __shared__ __attribute__((aligned(16))) float xe[2][32 * XSTR];

    // XCD-aware mapping: the 8 slices of one state sit on block indices with equal residue mod 8
    const int nslice = (n_mtiles + AW - 1) / AW;
    const int b = blockIdx.x;
    int w, slice;
    if (nslice == 8) {
        w = (b & 7) + 8 * (b >> 6);
        slice = (b >> 3) & 7;
    } else {
        w = b / nslice;
        slice = b - w * nslice;
    }
    if (w >= n_states) return;
    const int j = work_states[w];
    const long long beg = off[seg_lo[w]], end = off[seg_hi[w]];
    if (beg == end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int mt = slice * AW + wave;
    const bool live = mt < n_mtiles;
    const float *cen = centers + (size_t)j * D;

    // parameters of this wave's m-tile, from the scoring layout [m-tile][KS4][64][4] (spare slot = 1 carries cf)
    //   T16 = false: 32x32x2 products, the layout is used as it is (lane = half * 32 + mixture)
    //   T16 = true : 16x16x4 products (no padding columns in the moment product: 5 x 16 = 80 instead of 3 x 32 = 96):
    //                lane l needs P[kappa = 4 s + (l >> 4)][mixture ms * 16 + (l & 15)], kappa = 2 * pair + half
    constexpr int K16 = (2 * D + 2 + 3) / 4;          // k-steps of 4
    constexpr int NCT16 = (2 * D + 1 + 15) / 16;      // 16-column tiles of the moment product
    float pb[T16 ? 2 * K16 : KS4 * 4];
    if constexpr (T16) {
        const float *base = pm + ((size_t)j * n_mtiles + (live ? mt : 0)) * (KS4 * 64 * 4);
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int s = 0; s < K16; ++s) {
                const int kappa = 4 * s + (lane >> 4), pair = kappa >> 1, hf = kappa & 1, cl = ms * 16 + (lane & 15);
                pb[ms * K16 + s] = (pair < KS4 * 4) ? base[((size_t)(pair >> 2) * 64 + hf * 32 + cl) * 4 + (pair & 3)] : 0.f;
            }
    } else {
        const float4 *pa = reinterpret_cast<const float4 *>(pm) + ((size_t)j * n_mtiles + (live ? mt : 0)) * (KS4 * 64) + lane;
#pragma unroll
        for (int q = 0; q < KS4; ++q) {
            const float4 t = pa[q * 64];
            pb[4 * q] = t.x; pb[4 * q + 1] = t.y; pb[4 * q + 2] = t.z; pb[4 * q + 3] = t.w;
        }
    }
    f16v S[T16 ? 1 : NCT];
    f4x S16[T16 ? 2 : 1][T16 ? NCT16 : 1];
    if constexpr (T16) {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ct = 0; ct < NCT16; ++ct) S16[ms][ct] = f4x{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[ct][r] = 0.f;
    }
    double galpha = 0.0;
    constexpr double LOG2E = 1.4426950408889634074;

    // Staging is split (issue early / write late): the gather loads of tile i+1 are issued into registers
    // BEFORE the MFMAs of tile i and written to the other LDS buffer AFTER them, so their latency (two
    // dependent loads: list entry -> frame row) is covered by 88 MFMAs instead of being exposed per tile.
    constexpr int NE = (32 * D + AW * 64 - 1) / (AW * 64);   // feature elements per thread per tile
    float xv[NE];
    long long fidx[NE];                                      // frame rows of the tile after next (one more stage ahead)
    float cfv = -INFINITY, cfn = -INFINITY;
    double lgv = 0.0, lgn = 0.0;
    auto stage_index = [&](long long f0) {                   // level 1 of the gather: list entries
        const int nf = (f0 < end) ? (int)min(32LL, end - f0) : 0;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = threadIdx.x + k * AW * 64;
            const int f = e / D;
            fidx[k] = (e < 32 * D && f < nf) ? list[f0 + f].frame : -1;
        }
        cfn = -INFINITY;                                     // padding frame: g = exp2(-inf) = 0
        lgn = 0.0;
        if ((int)threadIdx.x < nf) {
            const ActiveFrame a = list[f0 + threadIdx.x];
            cfn = (float)(a.coef * LOG2E);
            lgn = a.lg;
        }
    };
    auto stage_load = [&]() {                                // level 2: frame rows, from the indices loaded a tile ago
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = threadIdx.x + k * AW * 64;
            const int d = e % D;
            xv[k] = (fidx[k] >= 0) ? frames[fidx[k] * D + d] - cen[d] : 0.f;
        }
        cfv = cfn;
        lgv = lgn;
    };
    auto stage_store = [&](int buf) {
        float *x = xe[buf];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = threadIdx.x + k * AW * 64;
            const int f = e / D, d = e - f * D;
            if (e < 32 * D) {
                x[f * XSTR + 2 * d] = xv[k] * xv[k];
                x[f * XSTR + 2 * d + 1] = xv[k];
            }
        }
        if (threadIdx.x < 32) {
            const int f = threadIdx.x;
            if (slice == 0) galpha += lgv;
            x[f * XSTR + 2 * D] = 1.f;
            x[f * XSTR + 2 * D + 1] = cfv;
            for (int k = 2 * D + 2; k < XSTR; ++k) x[f * XSTR + k] = 0.f;
        }
    };

    stage_index(beg);
    stage_load();
    stage_store(0);
    stage_index(beg + 32);
    __syncthreads();
    int buf = 0;
    for (long long f0 = beg; f0 < end; f0 += 32) {
        const bool more = f0 + 32 < end;
        if (more) {
            stage_load();                                // tile i+1: frame rows in flight during the MFMAs below
            stage_index(f0 + 64);                        // tile i+2: list entries
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the loads in front of the matrix work
        if (live) {
            const float *x = xe[buf];
            if constexpr (T16) {
                const int l15 = lane & 15, kk = lane >> 4;
                // (1) D1[ft][ms][frame 4 kk + reg][mixture l15] = Xe . P, four independent 16x16 chains
                f4x d1[2][2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms) d1[ft][ms] = f4x{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < K16; ++s) {
                    const float a0 = x[l15 * XSTR + 4 * s + kk], a1 = x[(16 + l15) * XSTR + 4 * s + kk];
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms) {
                        d1[0][ms] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, pb[ms * K16 + s], d1[0][ms], 0, 0, 0);
                        d1[1][ms] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, pb[ms * K16 + s], d1[1][ms], 0, 0, 0);
                    }
                }
                // posteriors gamma_t(j,m)   (Clustering.py:660-661)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d1[ft][ms][r] = __builtin_amdgcn_exp2f(d1[ft][ms][r]);
                // (2) S[ms][ct][mixture][feature] += g^T . Xe ; register r of d1 = frames 4 kk + r (k = kk)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fr = ft * 16 + 4 * kk + r;
#pragma unroll
                        for (int ct = 0; ct < NCT16; ++ct) {
                            const float bx = x[fr * XSTR + ct * 16 + l15];
                            S16[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[ft][0][r], bx, S16[0][ct], 0, 0, 0);
                            S16[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[ft][1][r], bx, S16[1][ct], 0, 0, 0);
                        }
                    }
            } else {
                // (1) D1[frame][mixture] = Xe . P   (log2 domain, + cf)
                f16v d1;
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[col * XSTR + 2 * s + half], pb[s], d1, 0, 0, 0);
                // posteriors gamma_t(j,m)   (Clustering.py:660-661)
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = __builtin_amdgcn_exp2f(d1[r]);
                // (2) S[mixture][feature] += g^T . Xe ; register r of d1 = frames (row(r), row(r)+4)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fr = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        S[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[r], x[fr * XSTR + ct * 32 + col], S[ct], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_store(buf ^ 1);                  // nobody reads buf^1 until the barrier below
        __syncthreads();
        buf ^= 1;
    }

    // ---- flush: lane = feature column, register = mixture row; cov = S2 - 2 d S1 + d^2 S0, mean = S1 + (c + bias) S0
    auto flush_one = [&](int m, int cidx, float s0, float s1, float s2v) {
        if (m < M && !(cidx & 1) && cidx < 2 * D) {
            const int d = cidx >> 1;
            const size_t o = ((size_t)j * Mpad + m) * D + d;
            const double c = (double)cen[d], dl = means64[o] - c;
            const double S0 = (double)s0, S1 = (double)s1, S2 = (double)s2v;
            st_mean[o] += S1 + (c + bias) * S0;                            // Clustering.py:669-672
            st_cov[o] += fmax(S2 - 2.0 * dl * S1 + dl * dl * S0, 0.0);     // Clustering.py:674-678 (never negative: gmm_accumulate_f16.hip)
        }
        if (m < M && cidx == 2 * D) st_acc[(size_t)j * Mpad + m] += (double)s2v;   // Clustering.py:665
    };
    if (live) {
        if constexpr (T16) {
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 32 + ms * 16 + 4 * (lane >> 4) + r;
                    const float s0 = __shfl(S16[ms][(2 * D) >> 4][r], (lane & 48) + ((2 * D) & 15), 64);
#pragma unroll
                    for (int ct = 0; ct < NCT16; ++ct) {
                        const float s1 = __shfl_xor(S16[ms][ct][r], 1, 64);
                        flush_one(m, ct * 16 + (lane & 15), s0, s1, S16[ms][ct][r]);
                    }
                }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float s0 = __shfl(S[(2 * D) >> 5][r], (lane & 32) + ((2 * D) & 31), 64);   // column 2D = the constant feature
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const float s1 = __shfl_xor(S[ct][r], 1, 64);                      // odd neighbour: x' column of the same d
                    flush_one(m, ct * 32 + col, s0, s1, S[ct][r]);
                }
            }
        }
    }
    if (slice == 0) {
        double v = (threadIdx.x < 32) ? galpha : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) st_alpha[j] += v;                                    // Clustering.py:667
    }
}

template <int D, typename real, int MINW>
void launch_acc_t(pcl_ctx *ctx, pcl_batch *b, const real *frames, const real *params, const real *means, int first, int count,
                  const int *tile_off = nullptr, const unsigned int *tile_mask = nullptr, const int *state_flag = nullptr) {
    if (count == 0) return;
    dim3 grid((ctx->Mpad + WG - 1) / WG, (unsigned)count);
    if (tile_mask && sizeof(real) == 4) {                              // the masked fix-up: rows from the master copy
        const AccMaster mm{ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Dhost, ctx->model_flags, ctx->d_bad, nullptr, nullptr};
        hipLaunchKernelGGL((gmm_accumulate_kernel<D, float, MINW, true>), grid, dim3(WG), 0, ctx->stream, (const float *)frames, (const float *)nullptr,
                           (const float *)nullptr, ctx->Mpad, b->ctx->acc.d_work_states + first, b->ctx->acc.d_seg_lo + first, b->ctx->acc.d_seg_hi + first, b->ctx->acc.acc_off, b->ctx->acc.acc_list, 100.0,
                           ctx->st_acc, ctx->st_alpha, ctx->st_mean, ctx->st_cov, tile_off, tile_mask, state_flag, mm);
        return;
    }
    hipLaunchKernelGGL((gmm_accumulate_kernel<D, real, MINW>), grid, dim3(WG), 0, ctx->stream, frames, params, means,
                       ctx->Mpad, b->ctx->acc.d_work_states + first, b->ctx->acc.d_seg_lo + first, b->ctx->acc.d_seg_hi + first, b->ctx->acc.acc_off, b->ctx->acc.acc_list, 100.0, ctx->st_acc,
                       ctx->st_alpha, ctx->st_mean, ctx->st_cov, tile_off, tile_mask, state_flag, AccMaster{});
}

// split states among [first, first + count) of the accumulate order (split_flag marks them): their off-pipe mixtures over all their frames
template <int D, int MINW>
void launch_acc_subset_t(pcl_ctx *ctx, pcl_batch *b, int first, int count, const int *split_flag) {
    int most = 1;                                                         // slices for the state with the most off-pipe mixtures in this range
    for (int k = first; k < first + count; ++k)
        if (b->acc_split[k]) most = std::max(most, ctx->nbad[b->acc_ws[k]]);
    dim3 grid((unsigned)count, (most + WG - 1) / WG);                     // (states, slices): see the kernel
    const AccMaster mm{ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Dhost, ctx->model_flags, ctx->d_bad, ctx->d_bad_idx, ctx->d_nbad};
    hipLaunchKernelGGL((gmm_accumulate_kernel<D, float, MINW, true>), grid, dim3(WG), 0, ctx->stream, ctx->frames32, (const float *)nullptr,
                       (const float *)nullptr, ctx->Mpad, b->ctx->acc.d_work_states + first, b->ctx->acc.d_seg_lo + first, b->ctx->acc.d_seg_hi + first, b->ctx->acc.acc_off, b->ctx->acc.acc_list, 100.0,
                       ctx->st_acc, ctx->st_alpha, ctx->st_mean, ctx->st_cov, (const int *)nullptr, (const unsigned int *)nullptr, split_flag + first, mm);
}
void launch_acc_subset(pcl_ctx *ctx, pcl_batch *b, int first, int count, const int *split_flag) {
    if (count == 0) return;
    switch (ctx->D) {
        case 13: launch_acc_subset_t<13, 2>(ctx, b, first, count, split_flag); break;
        case 26: launch_acc_subset_t<26, 2>(ctx, b, first, count, split_flag); break;
        case 39: launch_acc_subset_t<39, 2>(ctx, b, first, count, split_flag); break;
        case 47: launch_acc_subset_t<47, 2>(ctx, b, first, count, split_flag); break;
        default: break;
    }
}

// f32, states [first, first + count) of the accumulate order (optionally only the frames a tile mask marks)
void launch_acc_f32(pcl_ctx *ctx, pcl_batch *b, int first, int count, const int *tile_off = nullptr, const unsigned int *tile_mask = nullptr,
                    const int *state_flag = nullptr) {
    switch (ctx->D) {
#define CASE32(DD) case DD: launch_acc_t<DD, float, 2>(ctx, b, ctx->frames32, ctx->params32, ctx->mean32, first, count, tile_off, tile_mask, state_flag); break;
        CASE32(13) CASE32(26) CASE32(39) CASE32(47)
#undef CASE32
#define CASE32W(DD) case DD: launch_acc_t<DD, float, 1>(ctx, b, ctx->frames32, ctx->params32, ctx->mean32, first, count, tile_off, tile_mask, state_flag); break;
        CASE32W(48) CASE32W(64)
#undef CASE32W
        default: break;
    }
}

bool device_dim_supported(int D) { return D == 13 || D == 26 || D == 39 || D == 47 || D == 48 || D == 64; }

}  // namespace

void pcl_accumulate_release(pcl_ctx *ctx) {
    struct { pcl_ctx *ctx; } bb{ctx}, *b = &bb;              // (the body below names the members through b->ctx->acc)
    dev_free(b->ctx->acc.acc_cnt);
    dev_free(b->ctx->acc.acc_off);
    dev_free(b->ctx->acc.acc_list);
    dev_free(b->ctx->acc.d_work_states);
    dev_free(b->ctx->acc.d_seg_lo);
    dev_free(b->ctx->acc.d_seg_hi);
    dev_free(b->ctx->acc.d_split_flag);
    for (int k = 0; k < 2; ++k) {
        dev_free(b->ctx->acc.acc16_images[k]);
        dev_free(b->ctx->acc.acc16_tile_off[k]);
        dev_free(b->ctx->acc.acc16_tile_mask[k]);
        dev_free(b->ctx->acc.acc16_state_flag[k]);
        if (b->ctx->acc.acc16_ev_prod[k]) (void)hipEventDestroy(b->ctx->acc.acc16_ev_prod[k]);
        if (b->ctx->acc.acc16_ev_cons[k]) (void)hipEventDestroy(b->ctx->acc.acc16_ev_cons[k]);
        b->ctx->acc.acc16_images[k] = nullptr; b->ctx->acc.acc16_tile_off[k] = nullptr; b->ctx->acc.acc16_tile_mask[k] = nullptr; b->ctx->acc.acc16_state_flag[k] = nullptr;
        b->ctx->acc.acc16_ev_prod[k] = b->ctx->acc.acc16_ev_cons[k] = nullptr;
    }
    if (b->ctx->acc.acc16_ev_start) (void)hipEventDestroy(b->ctx->acc.acc16_ev_start);
    b->ctx->acc.acc16_ev_start = nullptr;
    b->ctx->acc.acc16_cap_tiles = b->ctx->acc.acc16_cap_states = 0;
    b->ctx->acc.acc_cnt = nullptr; b->ctx->acc.acc_off = nullptr; b->ctx->acc.acc_list = nullptr;
    b->ctx->acc.d_work_states = b->ctx->acc.d_seg_lo = b->ctx->acc.d_seg_hi = b->ctx->acc.d_split_flag = nullptr;
    b->ctx->acc.acc_cap_list = b->ctx->acc.acc_cap_segs = b->ctx->acc.acc_cap_states = 0;
}

int pcl_launch_accumulate(pcl_ctx *ctx, pcl_batch *b, int precision) {
    if (b->n_segs == 0) return PCL_OK;
    size_t cap = 0;
    for (size_t k = 0; k < b->work_states.size(); ++k) {
        const ScoreSeg &last = b->segs[b->state_seg_hi[k] - 1];
        cap += (size_t)last.vstart + last.len;
    }
    if (b->ctx->acc.acc_cap_segs < (size_t)b->n_segs + 1) {
        dev_free(b->ctx->acc.acc_cnt);
        dev_free(b->ctx->acc.acc_off);
        b->ctx->acc.acc_cnt = nullptr; b->ctx->acc.acc_off = nullptr;
        TRY(dev_alloc(ctx, &b->ctx->acc.acc_cnt, (size_t)(((size_t)b->n_segs + 1))));
        TRY(dev_alloc(ctx, &b->ctx->acc.acc_off, (size_t)b->n_segs + 1));
        b->ctx->acc.acc_cap_segs = (size_t)b->n_segs + 1;
    }
    if (b->ctx->acc.acc_cap_list < cap) {
        dev_free(b->ctx->acc.acc_list);
        b->ctx->acc.acc_list = nullptr;
        TRY(dev_alloc(ctx, &b->ctx->acc.acc_list, (size_t)(cap)));
        b->ctx->acc.acc_cap_list = cap;
    }
    const size_t ns = b->work_states.size();
    if (b->ctx->acc.acc_cap_states < ns) {
        dev_free(b->ctx->acc.d_work_states);
        dev_free(b->ctx->acc.d_seg_lo);
        dev_free(b->ctx->acc.d_seg_hi);
        dev_free(b->ctx->acc.d_split_flag);
        b->ctx->acc.d_work_states = b->ctx->acc.d_seg_lo = b->ctx->acc.d_seg_hi = b->ctx->acc.d_split_flag = nullptr;
        TRY(dev_alloc(ctx, &b->ctx->acc.d_split_flag, (size_t)(ns)));
        TRY(dev_alloc(ctx, &b->ctx->acc.d_work_states, (size_t)(ns)));
        TRY(dev_alloc(ctx, &b->ctx->acc.d_seg_lo, (size_t)(ns)));
        TRY(dev_alloc(ctx, &b->ctx->acc.d_seg_hi, (size_t)(ns)));
        b->ctx->acc.acc_cap_states = ns;
    }
    // MFMA mode: well-conditioned states first (MFMA kernel), then the ill-conditioned ones (direct-form VALU kernel)
    const int D = ctx->D;
    const bool mfma = precision == PCL_F32 && ctx->score_variant >= 3 && (D == 47 || D == 39 || D == 26 || D == 13);
    b->acc_ws.clear(); b->acc_lo.clear(); b->acc_hi.clear();
    int n_good = 0, n_split = 0;
    b->acc_split.clear();
    for (int pass = 0; pass < 2; ++pass)
        for (size_t k = 0; k < ns; ++k) {
            const bool bad = mfma && pcl_state_acc_uses_valu(ctx, b->work_states[k]);
            if (bad != (pass == 1)) continue;
            b->acc_ws.push_back(b->work_states[k]);
            b->acc_lo.push_back(b->state_seg_lo[k]);
            b->acc_hi.push_back(b->state_seg_hi[k]);
            const int sp = (mfma && pass == 0 && pcl_state_acc_is_split(ctx, b->work_states[k])) ? 1 : 0;
            b->acc_split.push_back(sp);
            n_split += sp;
            n_good += pass == 0;
        }
    const int n_bad = (int)ns - n_good;
    // the staging vectors live in the batch: the copies below are asynchronous from pageable memory only until the
    // call returns on this runtime, but keeping them alive costs nothing
    HIPCHK(ctx, hipMemcpyAsync(b->ctx->acc.d_work_states, b->acc_ws.data(), ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->ctx->acc.d_seg_lo, b->acc_lo.data(), ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->ctx->acc.d_seg_hi, b->acc_hi.data(), ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (n_split) HIPCHK(ctx, hipMemcpyAsync(b->ctx->acc.d_split_flag, b->acc_split.data(), ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));

    // a frame survives unless every gamma_t(j,m) <= gamma_t(j) underflows to exactly 0 in the
    // kernel's arithmetic (f32: 2^-149, f64: 2^-1074)
    const double LN2 = 0.693147180559945309417232121458;
    const double thr = std::max(precision == PCL_F64 ? -1076.0 : -150.0, ctx->acc_prune_log2) * LN2;
    pcl_timer_begin(ctx, "accumulate");
    const int wpb = 4;
    dim3 gseg((b->n_segs + wpb - 1) / wpb);
    static const bool rows_on = !(getenv("PCL_ACC_ROWS") && atoi(getenv("PCL_ACC_ROWS")) == 0);      // 0: a wave per segment (rounds 1-3), A/B
    const bool by_rows = rows_on && b->U <= 65535 && b->d_seg_of_row;                                // (grid.y = utterances)
    const dim3 grows((b->max_N + 8 * ROWS_WPB - 1) / (8 * ROWS_WPB), (unsigned)b->U);
    if (by_rows) hipLaunchKernelGGL(acc_count_rows_kernel, grows, dim3(64 * ROWS_WPB), 0, ctx->stream, b->d_utt, b->d_seg_of_row, b->lgam, thr, b->ctx->acc.acc_cnt);
    else hipLaunchKernelGGL(acc_count_kernel, gseg, dim3(64 * wpb), 0, ctx->stream, b->d_segs, b->n_segs, b->lgam, thr, b->ctx->acc.acc_cnt);
    hipLaunchKernelGGL(acc_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, b->ctx->acc.acc_cnt, b->n_segs, b->ctx->acc.acc_off);
    if (by_rows) hipLaunchKernelGGL(acc_fill_rows_kernel, grows, dim3(64 * ROWS_WPB), 0, ctx->stream, b->d_utt, b->d_seg_of_row, b->lgam, b->Bt, thr,
                                    b->ctx->acc.acc_off, b->ctx->acc.acc_list);
    else hipLaunchKernelGGL(acc_fill_kernel, gseg, dim3(64 * wpb), 0, ctx->stream, b->d_segs, b->n_segs, b->lgam, b->Bt, thr,
                            b->ctx->acc.acc_off, b->ctx->acc.acc_list);
    if (mfma && n_good > 0 && ctx->score_variant == 7) {
        // producer / consumer on the f16 + bf16 matrix pipes (gmm_accumulate_f16.hip), in groups of states whose tile
        // images fit the image buffer (worst case: every frame of the state survives)
        const size_t budget = (size_t)(getenv("PCL_ACC_IMAGE_MB") ? std::max(1L, atol(getenv("PCL_ACC_IMAGE_MB"))) : 2048) << 20;   // per buffer set, two sets; read per call (tests lower it to force many groups)
        const size_t ib = pcl_acc16_image_bytes(D);
        // group by the ACTUAL tile counts: the scan result comes back (n_segs + 1 offsets, one short copy behind the scan
        // kernel) -- on peaked posteriors a tenth of the frames survive, and sizing the groups for the worst case made
        // eight half-empty launches, each with its parameter prologue and statistics flush, out of one
        std::vector<long long> off_h((size_t)b->n_segs + 1);
        HIPCHK(ctx, hipMemcpyAsync(off_h.data(), b->ctx->acc.acc_off, off_h.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int> wtiles(n_good);
        size_t worst = 0, biggest = 0;
        for (int k = 0; k < n_good; ++k) {
            wtiles[k] = (int)((off_h[b->acc_hi[k]] - off_h[b->acc_lo[k]] + 31) / 32);
            worst += wtiles[k];
            biggest = std::max(biggest, (size_t)wtiles[k]);
        }
        if (getenv("PCL_DEBUG_ACC")) {                       // distribution of the active-frame lists over the states (diagnostic)
            std::vector<int> sorted(wtiles);
            std::sort(sorted.begin(), sorted.end());
            if (!sorted.empty())
                fprintf(stderr, "[pcl] accumulate: %d states, active 32-frame tiles per state: min %d median %d p90 %d p99 %d max %d, total %zu\n", n_good,
                        sorted.front(), sorted[sorted.size() / 2], sorted[sorted.size() * 9 / 10], sorted[sorted.size() * 99 / 100], sorted.back(), worst);
        }
        const size_t by_budget = std::max<size_t>(budget / ib, 1);
        size_t cap_tiles = std::max(biggest, std::min(worst, by_budget));
        if (b->ctx->acc.acc16_cap_tiles >= cap_tiles) cap_tiles = b->ctx->acc.acc16_cap_tiles;                 // never shrink: the counts move from call to call
        else cap_tiles = std::max(cap_tiles, std::min(cap_tiles + cap_tiles / 4, std::max(by_budget, biggest)));   // grow with headroom
        if (b->ctx->acc.acc16_cap_tiles < cap_tiles) {
            for (int k = 0; k < 2; ++k) {
                dev_free(b->ctx->acc.acc16_images[k]);
                dev_free(b->ctx->acc.acc16_tile_mask[k]);
                b->ctx->acc.acc16_images[k] = nullptr; b->ctx->acc.acc16_tile_mask[k] = nullptr;
                b->ctx->acc.acc16_images[k] = pcl_pool_alloc(ctx->device, cap_tiles * ib);
                if (!b->ctx->acc.acc16_images[k]) PCL_FAIL(ctx, PCL_ERR_NOMEM, "device memory: %zu bytes of tile images", cap_tiles * ib);
                TRY(dev_alloc(ctx, &b->ctx->acc.acc16_tile_mask[k], cap_tiles));
            }
            b->ctx->acc.acc16_cap_tiles = cap_tiles;
        }
        if (b->ctx->acc.acc16_cap_states < (size_t)n_good + 1) {
            for (int k = 0; k < 2; ++k) {
                dev_free(b->ctx->acc.acc16_tile_off[k]);
                dev_free(b->ctx->acc.acc16_state_flag[k]);
                b->ctx->acc.acc16_tile_off[k] = b->ctx->acc.acc16_state_flag[k] = nullptr;
                TRY(dev_alloc(ctx, &b->ctx->acc.acc16_tile_off[k], (size_t)(((size_t)n_good + 1))));
                TRY(dev_alloc(ctx, &b->ctx->acc.acc16_state_flag[k], (size_t)(((size_t)n_good + 1))));
            }
            b->ctx->acc.acc16_cap_states = (size_t)n_good + 1;
        }
        if (!b->ctx->acc.acc16_ev_start) {
            HIPCHK(ctx, hipEventCreateWithFlags(&b->ctx->acc.acc16_ev_start, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) {
                HIPCHK(ctx, hipEventCreateWithFlags(&b->ctx->acc.acc16_ev_prod[k], hipEventDisableTiming));
                HIPCHK(ctx, hipEventCreateWithFlags(&b->ctx->acc.acc16_ev_cons[k], hipEventDisableTiming));
            }
        }
        // groups of states: (first, count, worst-case tiles)
        // the FIRST group is small: its producer is the only one nothing runs beside (1.25 ms of a flat pass when the groups are equal)
        static const int first_div = getenv("PCL_ACC_FIRST_DIV") ? atoi(getenv("PCL_ACC_FIRST_DIV")) : 12;
        std::vector<int> gfirst, gcount, gtiles;
        for (int first = 0; first < n_good;) {
            size_t t = 0;
            int last = first;
            const size_t cap_g = (first == 0 && first_div > 1) ? std::max(biggest, std::min(cap_tiles, worst / (size_t)first_div)) : cap_tiles;
            while (last < n_good && t + wtiles[last] <= cap_g) t += wtiles[last++];
            gfirst.push_back(first); gcount.push_back(last - first); gtiles.push_back((int)t);
            first = last;
        }
        // the producer of group g + 1 runs on the auxiliary stream beside the consumer of group g (it is HBM-write
        // bound, the consumer matrix-pipe bound, and a consumer workgroup leaves registers for one small wave per SIMD)
        static const bool overlap = !(getenv("PCL_ACC_OVERLAP") && atoi(getenv("PCL_ACC_OVERLAP")) == 0);
        hipStream_t ps = overlap ? ctx->stream_aux : ctx->stream;
        HIPCHK(ctx, hipEventRecord(b->ctx->acc.acc16_ev_start, ctx->stream));            // the active-frame lists are complete
        if (overlap) HIPCHK(ctx, hipStreamWaitEvent(ps, b->ctx->acc.acc16_ev_start, 0));
        const int G = (int)gfirst.size();
        bool ascending = true;
        for (int k = 1; k < n_good; ++k) ascending = ascending && b->acc_ws[k - 1] < b->acc_ws[k];
        auto produce = [&](int g) -> int {
            const int buf = g & 1;
            if (overlap && g >= 2) HIPCHK(ctx, hipStreamWaitEvent(ps, b->ctx->acc.acc16_ev_cons[buf], 0));      // the buffer set is free again
            const int rc = pcl_launch_acc16_produce(ctx, b, gfirst[g], gcount[g], gtiles[g], buf, ps);
            if (rc != PCL_OK) return rc;
            if (overlap) HIPCHK(ctx, hipEventRecord(b->ctx->acc.acc16_ev_prod[buf], ps));
            return PCL_OK;
        };
        if (G > 0) { const int rc = produce(0); if (rc != PCL_OK) return rc; }
        for (int g = 0; g < G; ++g) {
            const int buf = g & 1;
            if (overlap && g + 1 < G) { const int rc = produce(g + 1); if (rc != PCL_OK) return rc; }
            if (overlap) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b->ctx->acc.acc16_ev_prod[buf], 0));
            pcl_timer_begin(ctx, "acc_consume");                                 // one entry per state group (timing mode only)
            int rc = pcl_launch_acc16_consume(ctx, b, gfirst[g], gcount[g], buf, ctx->stats_fresh, ctx->stream);
            pcl_timer_end(ctx, "acc_consume");
            if (rc != PCL_OK) return rc;
            launch_acc_f32(ctx, b, gfirst[g], gcount[g], b->ctx->acc.acc16_tile_off[buf], b->ctx->acc.acc16_tile_mask[buf], b->ctx->acc.acc16_state_flag[buf]);   // the frames the images left out
            if (n_split) {                                                                                    // the mixtures the pipe left out (split states)
                pcl_timer_begin(ctx, "acc_subset");
                launch_acc_subset(ctx, b, gfirst[g], gcount[g], b->ctx->acc.d_split_flag);
                pcl_timer_end(ctx, "acc_subset");
            }
            if (overlap) HIPCHK(ctx, hipEventRecord(b->ctx->acc.acc16_ev_cons[buf], ctx->stream));
            if (!overlap && g + 1 < G) { rc = produce(g + 1); if (rc != PCL_OK) return rc; }
            // a pipelined exchange is open (pcl_batch_accumulate_exchange): every state below the next group's first one has its
            // final statistics behind what is queued now -- its chunks may leave (states come in ascending order; states of the
            // direct-form kernel, if any, are accumulated at the end, so nothing is final before that)
            if (ctx->pipe_active && ascending && n_bad == 0) TRY(pcl_pipe_progress(ctx, g + 1 < G ? b->acc_ws[gfirst[g + 1]] : ctx->J));
        }
    } else if (mfma && n_good > 0) {
        const int nmt = ctx->Mpad32 / 32, nslice = (nmt + AW - 1) / AW, ns = n_good;
        const int nblocks = (nslice == 8) ? ((ns + 7) / 8) * 64 : ns * nslice;
#define LAUNCH_MFMA(DD)                                                                                                   \
    hipLaunchKernelGGL((gmm_accumulate_mfma_kernel<DD, PCL_ACC_T16 != 0>), dim3(nblocks), dim3(AW * 64), 0, ctx->stream, ctx->frames32, ctx->pm32, \
                       ctx->centers32, ctx->mean64, ctx->M, ctx->Mpad, nmt, ns, b->ctx->acc.d_work_states, b->ctx->acc.d_seg_lo, b->ctx->acc.d_seg_hi,   \
                       b->ctx->acc.acc_off, b->ctx->acc.acc_list, 100.0, ctx->st_acc, ctx->st_alpha, ctx->st_mean, ctx->st_cov)
        if (D == 47) LAUNCH_MFMA(47); else if (D == 39) LAUNCH_MFMA(39); else if (D == 26) LAUNCH_MFMA(26); else LAUNCH_MFMA(13);
#undef LAUNCH_MFMA
        if (n_split) launch_acc_subset(ctx, b, 0, n_good, b->ctx->acc.d_split_flag);
    }
    if (precision == PCL_F32) {
        const int first = mfma ? n_good : 0, count = mfma ? n_bad : (int)ns;
        if (count > 0) TRY(pcl_ensure_layouts(ctx, PCL_LAYOUT_P32));   // whole states on the direct-form kernel: its layouts, derived on first use
        if (device_dim_supported(D)) launch_acc_f32(ctx, b, first, count);
        else PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f32 accumulate kernel for padded D=%d", D);
    } else {
        switch (D) {
#define CASE64(DD) case DD: launch_acc_t<DD, double, 1>(ctx, b, ctx->frames64, ctx->params64, ctx->mean64, 0, (int)ns); break;
            CASE64(13) CASE64(26) CASE64(39) CASE64(47) CASE64(48) CASE64(64)
#undef CASE64
            default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f64 accumulate kernel for padded D=%d", D);
        }
    }
    pcl_timer_end(ctx, "accumulate");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// gmm_accumulate_f16.hip -- E-step sufficient statistics, producer / consumer form (gfx950), the default f32 path.
//
// Same reference rows as gmm_accumulate.hip (A13: Clustering.GMM.update_acc, StatisticalModel/Clustering.py:653-680,
// called from LHMM.update_acc, StatisticalModel/LHMM.py:497-505): two chained contractions per 32-frame x 32-mixture
// tile of one state's list of surviving frames,
//   (1) D1[f][m] = Xe[f,:] . P[:,m]            the scoring GEMM (K = 2D + 2): log2 of w_m N_m(o_f), relative to K0_j
//       g[f][m]  = exp2(D1 + cf_f)             cf_f = log2e (ln gamma_f(j) - ln b_j(o_f)) + K0_j: gamma_f(j,m)
//   (2) S[m][:] += sum_f g[f][m] [x'^2_d, x'_d | 1]      raw moments S2, S1, S0 about the state centre
// What changed against the round-1 kernel (48 ms on the bench shard, matrix pipe 43 % busy):
//   * product (1) runs in the scoring kernel's two-piece f16 scheme (gmm_score_split.hip, variant 7: operands scaled by
//     exact powers of two per (state, feature), x = h1 + h2 carries 22 bits, products a2x1 + a1x2 + a1x1, the constant
//     k'_m - K0_j folded into the spare K slot): 15 MFMAs instead of 30.  cf is added on the VALU in f32 (it is a
//     per-frame scalar of magnitude ~100: two f16 pieces would not hold it).  Product (2) is unchanged: posteriors in
//     two bf16 pieces, features in three, five cross terms (30 MFMAs) -- the statistics are the exact moments of the
//     frames under posteriors perturbed by < 2^-16 relative, and bf16 keeps the range of a rarely responsible mixture.
//   * the operand images of a tile are built ONCE by a producer kernel (gather of the 32 frame rows, centring, scaling,
//     splitting, both fragment layouts) and written to HBM in LDS-image order; the 8 workgroups that own the 8 x 256
//     mixtures of the state stream them in by LDS-DMA (global_load_lds, no VGPRs, no VALU).  Round 1 staged every tile
//     in each of the 8 workgroups with ~105 VALU instructions per thread per tile.
//   * the consumer is software pipelined across tiles: while the VALU turns D1 of tile t into posteriors, the matrix
//     pipe already runs product (1) of tile t+1, then product (2) of tile t; three LDS slots, one barrier per tile.
//   * a frame whose scaled feature leaves the f16 range (|x - c| beyond ~300 sigma of the tightest mixture) is taken
//     out of the image (g = 0) and marked in its tile's mask; the direct-form VALU kernel adds exactly those frames
//     afterwards (gmm_accumulate.hip, masked mode), so no input sees a clamped posterior.
// States whose centred expansion is ill conditioned (pcl_model_conditioning) never come here.
#include <algorithm>

#include "pcl_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

#ifndef PCL_ACC16_MT
#define PCL_ACC16_MT 1
#endif
#ifndef PCL_ACC16_AW
#define PCL_ACC16_AW (PCL_ACC16_MT == 1 ? 8 : 4)
#endif
constexpr int MT = PCL_ACC16_MT;       // 32-mixture tiles per consumer wave
constexpr int AW = PCL_ACC16_AW;       // waves per consumer workgroup
constexpr float FMAXH = 6.0e4f;        // what an f16 piece may hold (gmm_score_split.hip)
constexpr double LOG2E = 1.4426950408889634074;

// LDS-DMA of 16 B per lane (1 KiB per wave) as an asm statement: hipcc does not count an asm memory operation, so it does
// not drain it (s_waitcnt vmcnt(0)) in front of the next ds_read the way it does for __builtin_amdgcn_global_load_lds --
// which would expose the whole issue -> landed latency in every tile.  Completion is counted by hand (vmcnt) below.
// M0 carries the wave-uniform LDS byte address and is restored (cdna_hip_programming.md, inline-asm notes).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned int lds_dst) {
    unsigned int keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);           // wave-uniform by construction; tell the compiler
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned int lds_addr(const void *p) {
    return (unsigned int)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}

template <int D>
struct Img {                           // image of one 32-frame tile: 1-KiB blocks (64 lanes x 16 B), LDS-DMA order
    static constexpr int KS = (D + 7) / 8;                 // K-steps of 16 of product (1) (spare slot at d = D)
    static constexpr int NCT = (2 * D + 1 + 31) / 32;      // 32-column tiles of product (2)
    static constexpr int B1 = 0;                           // [piece 2][KS]: frame-major f16 fragments (lane = side * 32 + frame)
    static constexpr int B2 = 2 * KS;                      // [piece 3][NCT][k-step 2]: feature-major bf16 fragments
    static constexpr int BM = B2 + 3 * NCT * 2;            // misc: cf in register order [2][16] f32 @0, gamma_f(j) [32] f64 @256
    static constexpr int NB = BM + 1;
    static_assert(D % 8 != 0, "the folded constant needs a spare K slot");
};

// ---------------------------------------------------------------------------------------------------------------
// tile bookkeeping: tile_off[w] = first tile of state w of this launch's state range (exclusive scan of ceil(n/32))
__global__ void acc16_tiles_kernel(const int *__restrict__ seg_lo, const int *__restrict__ seg_hi, const long long *__restrict__ off,
                                   int n_states, int *__restrict__ tile_off, int *__restrict__ state_flag) {
    __shared__ int part[1024];
    for (int w = threadIdx.x; w < n_states; w += blockDim.x) state_flag[w] = 0;     // set by the producer: a frame of the state left the f16 range
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (n_states + nt - 1) / nt;
    const int lo = min(tid * per, n_states), hi = min(lo + per, n_states);
    int s = 0;
    for (int w = lo; w < hi; ++w) s += (int)((off[seg_hi[w]] - off[seg_lo[w]] + 31) / 32);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nt; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        tile_off[n_states] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int w = lo; w < hi; ++w) {
        tile_off[w] = run;
        run += (int)((off[seg_hi[w]] - off[seg_lo[w]] + 31) / 32);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// producer: one workgroup per tile
template <int D>
__global__ __launch_bounds__(256) void acc16_producer_kernel(
    const float *__restrict__ frames, const float *__restrict__ centers, const float *__restrict__ fscale,
    const double *__restrict__ kzero, int n_states, const int *__restrict__ work_states, const int *__restrict__ seg_lo,
    const int *__restrict__ seg_hi, const long long *__restrict__ off, const ActiveFrame *__restrict__ list, const int *__restrict__ tile_off, int tile_base,
    uint4 *__restrict__ images, unsigned int *__restrict__ tile_mask, int *__restrict__ state_flag) {
    using I = Img<D>;
    constexpr int KS = I::KS, NCT = I::NCT, XS = KS * 8 + 1;     // row stride of the staged tile (odd: conflict-free column reads)
    __shared__ float xs[32 * XS];
    __shared__ float cfs[32];
    __shared__ double lgs[32];
    __shared__ int s_state;
    __shared__ long long s_f0;
    __shared__ unsigned int s_mask;
    const int tid = threadIdx.x;
    const int n_tiles = tile_off[n_states];
    // the grid is sized for the chip, not for the (host-unknown) number of tiles: an empty workgroup is not free
    for (int tile = tile_base + blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                                             // the previous tile's LDS is no longer read
    if (tid == 0) {
        int lo = 0, hi = n_states - 1;                           // last state with tile_off[w] <= tile
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tile_off[mid] <= tile) lo = mid; else hi = mid - 1;
        }
        s_state = lo;
        s_f0 = off[seg_lo[lo]] + 32LL * (tile - tile_off[lo]);
        s_mask = 0u;
    }
    __syncthreads();
    const int w = s_state, j = work_states[w];
    const long long f0 = s_f0;
    const int nf = (int)min(32LL, off[seg_hi[w]] - f0);          // only the state's last tile is not full
    const float *cen = centers + (size_t)j * D;
    // stage the centred rows: lanes run along d (coalesced row reads)
    for (int e = tid; e < 32 * KS * 8; e += 256) {
        const int f = e / (KS * 8), d = e - f * (KS * 8);
        float v = 0.f;
        if (f < nf && d < D) v = frames[list[f0 + f].frame * D + d] - cen[d];
        xs[f * XS + d] = v;
    }
    if (tid < 32) {
        double cf = -INFINITY, lg = 0.0;
        if (tid < nf) {
            const ActiveFrame a = list[f0 + tid];
            cf = a.coef * LOG2E + kzero[j];
            lg = a.lg;
        }
        cfs[tid] = (float)cf;                                    // -inf: padding frame, g = 0
        lgs[tid] = lg;
    }
    __syncthreads();
    uint4 *img = images + (size_t)tile * (I::NB * 64);
    // ---- (1) frame-major f16 fragments: item = (k-step s, lane = side * 32 + frame): 8 features of one frame, both pieces
    const float *fs = fscale + (size_t)j * 2 * (KS * 8);
    unsigned int ovf = 0u;
    for (int it = tid; it < KS * 64; it += 256) {
        const int s = it >> 6, ln = it & 63, side = ln >> 5, f = ln & 31;
        unsigned short h1[8], h2[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int d = 8 * s + x;
            float val = 0.f;
            if (d < D) {
                const float xc = xs[f * XS + d];
                val = (side ? xc : xc * xc) * fs[side * (KS * 8) + d];
                if (__builtin_fabsf(val) > FMAXH) ovf |= 1u << f;
                val = __builtin_fminf(__builtin_fmaxf(val, -FMAXH), FMAXH);
            } else if (d == D) {
                val = side ? 0.f : 1.f;                          // x1: [1 | 0]  (a1: [k1 | 0], a2: [k2 | 1]: the sum is k1 + k2)
            }
            const _Float16 a = (_Float16)val;
            const _Float16 b = (d == D) ? (_Float16)0.f : (_Float16)(val - (float)a);
            h1[x] = __builtin_bit_cast(unsigned short, a);
            h2[x] = __builtin_bit_cast(unsigned short, b);
        }
        img[(I::B1 + 0 * KS + s) * 64 + ln] = make_uint4(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16), h1[4] | ((unsigned)h1[5] << 16), h1[6] | ((unsigned)h1[7] << 16));
        img[(I::B1 + 1 * KS + s) * 64 + ln] = make_uint4(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16), h2[4] | ((unsigned)h2[5] << 16), h2[6] | ((unsigned)h2[7] << 16));
    }
    if (ovf) atomicOr(&s_mask, ovf);
    // ---- (2) feature-major bf16 fragments: item = (column tile ct, k-step sp, lane = h * 32 + c): 8 frames of one column,
    //      element x <-> frame 16 sp + 8 (x >> 2) + 4 h + (x & 3) (the order product (1)'s accumulator registers come in)
    for (int it = tid; it < NCT * 2 * 64; it += 256) {
        const int ct = it / 128, sp = (it >> 6) & 1, ln = it & 63, h = ln >> 5, c = ln & 31;
        const int c2 = ct * 32 + c, d = c2 >> 1, side = c2 & 1;
        unsigned short p[3][8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int f = 16 * sp + 8 * (x >> 2) + 4 * h + (x & 3);
            float val = 0.f;
            if (c2 < 2 * D) {
                const float xc = xs[f * XS + d];
                val = side ? xc : xc * xc;
            } else if (c2 == 2 * D) {
                val = 1.f;                                       // S0
            }
            float r = val;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const __bf16 b = (__bf16)r;
                p[q][x] = __builtin_bit_cast(unsigned short, b);
                r -= (float)b;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            img[(I::B2 + (q * NCT + ct) * 2 + sp) * 64 + ln] =
                make_uint4(p[q][0] | ((unsigned)p[q][1] << 16), p[q][2] | ((unsigned)p[q][3] << 16), p[q][4] | ((unsigned)p[q][5] << 16), p[q][6] | ((unsigned)p[q][7] << 16));
    }
    __syncthreads();
    // ---- misc block: cf in the register order of product (1)'s accumulator (lane half h, register r <-> frame
    //      (r & 3) + 8 (r >> 2) + 4 h), frames taken out of the image get -inf; gamma_f(j) for alpha_acc
    const unsigned int mask = s_mask;
    if (tid < 64) {
        unsigned int w4[4] = {0u, 0u, 0u, 0u};
        if (tid < 8) {                                           // 32 floats = 8 chunks
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int q = tid * 4 + x, h = q >> 4, r = q & 15, f = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = ((mask >> f) & 1u) ? -INFINITY : cfs[f];
                w4[x] = __float_as_uint(v);
            }
        } else if (tid >= 16 && tid < 32) {                      // 32 doubles = 16 chunks, from byte 256
            const int f = (tid - 16) * 2;
            const unsigned long long a = __double_as_longlong(lgs[f]), b = __double_as_longlong(lgs[f + 1]);
            w4[0] = (unsigned int)a; w4[1] = (unsigned int)(a >> 32);
            w4[2] = (unsigned int)b; w4[3] = (unsigned int)(b >> 32);
        }
        img[I::BM * 64 + tid] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    if (tid == 0) {
        tile_mask[tile] = mask;
        if (mask) state_flag[w] = 1;                             // (same value from every writer)
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// consumer: AW waves x MT m-tiles = AW MT x 32 mixtures of one state per workgroup; the slices of a state sit on block
// indices with equal residue mod 8 (one XCD's L2).  MT = 2 at one wave per SIMD: every fragment read from LDS feeds two
// MFMAs, product (1) runs two independent chains and product (2) six.
// FRESH: the statistics are all zero (first pass after pcl_stats_zero) and every (state, mixture) belongs to exactly one wave:
// the flush stores instead of read-modify-writing 7.7 GB of float64.
template <int D, bool FRESH>
__global__ __launch_bounds__(AW * 64, MT == 1 ? 2 : 1) void acc16_consumer_kernel(
    const uint4 *__restrict__ images, const uint4 *__restrict__ pm16f, const float *__restrict__ centers,
    const double *__restrict__ means64, int M, int Mpad, int n_mtiles, int n_states, const int *__restrict__ work_states,
    const int *__restrict__ tile_off, int tile_base, double bias, double *__restrict__ st_acc, double *__restrict__ st_alpha,
    double *__restrict__ st_mean, double *__restrict__ st_cov) {
    using I = Img<D>;
    constexpr int KS = I::KS, NCT = I::NCT, NB = I::NB;
    constexpr int NSLOT = 5;                                     // tile t in slot t % 5: t .. t + 2 being read, t + 3 and t + 4 landing
    __shared__ __attribute__((aligned(16))) uint4 slot[NSLOT][NB * 64];

    const int nslice = (n_mtiles + AW * MT - 1) / (AW * MT);
    const int b = blockIdx.x;
    const int w = (b & 7) + 8 * (b / (8 * nslice));              // block b runs on XCD b % 8: all slices of a state on one XCD
    const int slice = (b >> 3) % nslice;
    if (w >= n_states) return;
    const int t0 = tile_off[w] - tile_base, t1 = tile_off[w + 1] - tile_base;
    if (t0 == t1) return;
    const int j = work_states[w];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int mt0 = (slice * AW + wave) * MT;                    // this wave's m-tiles: mt0 .. mt0 + MT - 1
    const bool live = mt0 < n_mtiles;                            // (a tile past the end repeats the last one and is not flushed)

    // parameters of this wave's m-tiles: the scoring layout of variant 7 as it is (B operand of product (1))
    h8v pf[MT][2][KS];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int mt = min(mt0 + i, n_mtiles - 1);
        const uint4 *pq = pm16f + ((size_t)j * n_mtiles + (live ? mt : 0)) * (2 * KS * 64) + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < KS; ++s) pf[i][p][s] = __builtin_bit_cast(h8v, pq[(p * KS + s) * 64]);
    }
    f16v S[MT][NCT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[i][ct][r] = 0.f;
    double galpha = 0.0;

    auto dma = [&](int t) {                                      // tile t -> slot t % NSLOT; the NB blocks are dealt to the waves
        const uint4 *src = images + (size_t)t * (NB * 64);
        const unsigned int dst = __builtin_amdgcn_readfirstlane(lds_addr(&slot[t % NSLOT][0]));
        for (int p = wave; p < NB; p += AW) glds16(src + p * 64 + lane, dst + (unsigned int)p * 1024u);
    };
    // (the frame fragments travel from one iteration to the next as plain 128-bit integers: carried as half vectors the
    //  compiler splits them into 16-bit halves at the loop edge and re-packs them with v_perm_b32, 80 VALU ops per tile)
    auto load1 = [&](int t, uint4 (&a1)[KS], uint4 (&a2)[KS]) {
#ifdef PCL_ACC16_DIAG_NOLDS
        const uint4 fake = make_uint4(t, lane, t ^ lane, 0x3c003c00u);
#pragma unroll
        for (int s = 0; s < KS; ++s) { a1[s] = fake; a2[s] = fake; }
        return;
#endif
        const uint4 *x1 = &slot[t % NSLOT][I::B1 * 64];
#pragma unroll
        for (int s = 0; s < KS; ++s) a2[s] = x1[(1 * KS + s) * 64 + lane];
#pragma unroll
        for (int s = 0; s < KS; ++s) a1[s] = x1[(0 * KS + s) * 64 + lane];
    };
    // The chains start from the frames' coefficients cf (ln gamma - ln b in log2 units, -inf for a frame that is not in the
    // image): the posterior's exponent comes out of the matrix pipe complete, no VALU add per value; d comes in holding cf
    // (load_cf writes the very registers the chain accumulates in) and goes out as D1.  Small cross terms first.
    auto mfma1 = [&](const uint4 (&a1)[KS], const uint4 (&a2)[KS], f16v (&d)[MT]) {
#ifdef PCL_ACC16_DIAG_NOP1
        d[0][0] += (float)a1[0].x + (float)a2[KS - 1].w;
        return;
#endif
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a2[s]), pf[i][0][s], d[i], 0, 0, 0);   // x2 a1
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a1[s]), pf[i][1][s], d[i], 0, 0, 0);   // x1 a2
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a1[s]), pf[i][0][s], d[i], 0, 0, 0);   // x1 a1
    };
    auto load2 = [&](int t, int sp, bf8v (&bq)[3][NCT]) {
#ifdef PCL_ACC16_DIAG_NOLDS
        const uint4 fake2 = make_uint4(t, lane, sp, 0x3f803f80u);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bq[p][ct] = __builtin_bit_cast(bf8v, fake2);
        return;
#endif
        const uint4 *x2 = &slot[t % NSLOT][I::B2 * 64];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bq[p][ct] = __builtin_bit_cast(bf8v, x2[((p * NCT + ct) * 2 + sp) * 64 + lane]);
    };
    auto mfma2 = [&](const bf8v (&g1)[MT], const bf8v (&g2)[MT], const bf8v (&bq)[3][NCT]) {
        // the MT x NCT accumulators are independent, issued round robin; small cross terms first
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < MT; ++i) S[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[i], bq[2][ct], S[i][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < MT; ++i) S[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[i], bq[1][ct], S[i][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < MT; ++i) S[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[i], bq[0][ct], S[i][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < MT; ++i) S[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[i], bq[1][ct], S[i][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < MT; ++i) S[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[i], bq[0][ct], S[i][ct], 0, 0, 0);
    };

    // Software pipeline, per wave, one basic block per tile t (the waves of a workgroup meet at ONE barrier per tile):
    //   region X   product (1) of tile t + 1  ||  VALU: D1 of tile t -> posteriors  ||  reads: product (2) fragments of tile t
    //   region Y   product (2) of tile t      ||  reads: cf and product (1) fragments of tile t + 2
    // LDS-DMA: tile t + 4 is issued at the top of tile t and waited for (counted vmcnt, leaving the newest tile in flight)
    // at the end of tile t + 1, two barriers before its first read: issued -> landed takes microseconds when every CU streams.
    constexpr int MYB_HI = (NB + AW - 1) / AW, MYB_LO = NB / AW;  // blocks per tile this wave issues: waves < NB % AW one more
    const bool more_blocks = wave < NB % AW;
    for (int k = 0; k < 4; ++k)
        if (t0 + k < t1) dma(t0 + k);
    // this wave's blocks have landed -- and, as a BUILTIN the compiler's wait-count pass sees, its own parameter loads too:
    // otherwise it re-waits for them (vmcnt(0)) at their first use inside the loop, every iteration, and drains the DMA
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();                                             // everyone's have
    // Two accumulator sets take turns (the tile loop is unrolled by two): while the posteriors of tile t are read out of one,
    // the chains of tile t + 1 run in the other, which was loaded with that tile's cf a phase earlier -- no register copies.
    f16v dA[MT], dB[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dA[i][r] = dB[i][r] = 0.f;
    uint4 a1[KS], a2[KS];
    auto load_cf = [&](int t, f16v (&c)[MT]) {                   // cf of tile t in the register order of D1 (lane half h, register r)
        const float4 *cfp = reinterpret_cast<const float4 *>(&slot[t % NSLOT][I::BM * 64]) + half * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = cfp[q];
#pragma unroll
            for (int i = 0; i < MT; ++i) { c[i][4 * q] = v.x; c[i][4 * q + 1] = v.y; c[i][4 * q + 2] = v.z; c[i][4 * q + 3] = v.w; }
        }
    };
    if (live) {
        load1(t0, a1, a2);
        load_cf(t0, dA);
        mfma1(a1, a2, dA);
        load_cf(t0 + 1, dB);                                     // (past the last tile: whatever the slot holds, nobody reads the product)
        load1(t0 + 1, a1, a2);
    }
#ifdef PCL_ACC16_STAMPS          // diagnostic build: where a wave's time goes (s_memtime at points where no LDS read is pending)
    unsigned long long st_busy = 0, st_vm = 0, st_bar = 0, st_prev = __builtin_amdgcn_s_memtime();
    unsigned int st_n = 0;
#endif
    auto tile_step = [&](int t, f16v (&d1)[MT], f16v (&dn)[MT]) __attribute__((always_inline)) {
#ifdef PCL_ACC16_DIAG_NODMA
        const bool ahead = false;
#else
        const bool ahead = t + 4 < t1;
#endif
        if (ahead) dma(t + 4);                                   // its slot held tile t - 1: everyone left it at the last barrier
        const uint4 *cur = &slot[t % NSLOT][0];
        if (slice == 0 && wave == 0 && lane < 32) galpha += reinterpret_cast<const double *>(cur + I::BM * 64)[32 + lane];   // byte 256: gamma_f(j)
        if (live) {
            bf8v bq0[3][NCT], bq1[3][NCT];
            bf8v g1a[MT], g1b[MT], g2a[MT], g2b[MT];
            // ---- region X
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                // posteriors gamma_t(j,m) (Clustering.py:660-661) of tile t in two bf16 pieces = the A fragments of product (2)
                typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
                unsigned int u1[8], u2[8];
#ifdef PCL_ACC16_DIAG_NOV
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    u1[r] = __float_as_uint(d1[i][2 * r]);
                    u2[r] = __float_as_uint(d1[i][2 * r + 1]);
                }
#else
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float ga = __builtin_amdgcn_exp2f(d1[i][2 * r]), gb = __builtin_amdgcn_exp2f(d1[i][2 * r + 1]);
                    const bf2v c = bf2v{(__bf16)ga, (__bf16)gb};                   // v_cvt_pk_bf16_f32
                    u1[r] = __builtin_bit_cast(unsigned int, c);
                    const bf2v e = bf2v{(__bf16)(ga - __uint_as_float(u1[r] << 16)), (__bf16)(gb - __uint_as_float(u1[r] & 0xffff0000u))};
                    u2[r] = __builtin_bit_cast(unsigned int, e);
                }
#endif
                g1a[i] = __builtin_bit_cast(bf8v, make_uint4(u1[0], u1[1], u1[2], u1[3]));
                g1b[i] = __builtin_bit_cast(bf8v, make_uint4(u1[4], u1[5], u1[6], u1[7]));
                g2a[i] = __builtin_bit_cast(bf8v, make_uint4(u2[0], u2[1], u2[2], u2[3]));
                g2b[i] = __builtin_bit_cast(bf8v, make_uint4(u2[4], u2[5], u2[6], u2[7]));
            }
            mfma1(a1, a2, dn);                                   // product (1) of tile t + 1, from its frames' coefficients
            load2(t, 0, bq0);                                    // (asked for at the top of the tile and pinned there with a sched_barrier,
            load2(t, 1, bq1);                                    //  the 18 reads make the first MFMA wait for all of them: 43.7 vs 41.9 ms)
            // ---- region Y
#ifndef PCL_ACC16_DIAG_NOP2      // (timing diagnostics: wrong results)
            mfma2(g1a, g2a, bq0);                                // product (2) of tile t: S[mixture][feature] += g^T . Xe
            mfma2(g1b, g2b, bq1);
#else
            S[0][0][0] += (float)g1a[0][0] + (float)g2b[MT - 1][1] + (float)bq0[0][0][0] + (float)bq1[2][NCT - 1][3];
#endif
            load_cf(t + 2, d1);                                  // (the posteriors of tile t have been read out of d1)
            load1(t + 2, a1, a2);
#ifdef PCL_ACC16_IGLP
            __builtin_amdgcn_iglp_opt(PCL_ACC16_IGLP);
#endif
#ifndef PCL_ACC16_NOSGB           // pin the fragment reads between the MFMAs (left alone the scheduler issues product (2)'s 6 NCT reads in
            if (MT == 1) {               // one burst right in front of its first MFMA): a third of them behind every third of product (1)'s
                                         // chain, the next tile's 4 + 2 KS reads spread through product (2).  41.9 vs 42.4 ms
                constexpr int NX3 = KS, RX3 = 2 * NCT, NY = 10 * NCT, RY = 4 + 2 * KS, PER = NY / RY > 0 ? NY / RY : 1;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NX3, 0);     // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, RX3, 0);     // DS read
                }
#pragma unroll
                for (int i = 0; i < RY; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (NY - PER * RY > 0) __builtin_amdgcn_sched_group_barrier(0x008, NY - PER * RY > 0 ? NY - PER * RY : 1, 0);
            }
#endif
        }
#ifdef PCL_ACC16_STAMPS
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        // this wave's blocks of tile t + 3 (issued two tiles ago) have landed; those of tile t + 4 stay in flight
        static_assert(MYB_HI <= 8 && MYB_LO >= 1, "counted waits below");
        {
            const int keep = !ahead ? 0 : (more_blocks ? MYB_HI : MYB_LO);       // wave-uniform
            switch (keep) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            }
        }
#ifdef PCL_ACC16_STAMPS
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();                            // (raw: __syncthreads would drain the DMA in flight) tile t + 3 is there for everyone; slot t % NSLOT is free
#ifdef PCL_ACC16_STAMPS
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
        st_busy += st_a - st_prev; st_vm += st_b - st_a; st_bar += st_c - st_b; st_prev = st_c; ++st_n;
#endif
    };
    for (int t = t0; t < t1; t += 2) {
        tile_step(t, dA, dB);
        if (t + 1 < t1) tile_step(t + 1, dB, dA);
    }
#ifdef PCL_ACC16_STAMPS
    if ((blockIdx.x == 40 || blockIdx.x == 1000) && lane == 0 && st_n)
        printf("block %d wave %d (live %d) tiles %u: work %llu  dma-wait %llu  barrier %llu  (s_memtime ticks per tile)\n", blockIdx.x, wave, (int)live, st_n,
               st_busy / st_n, st_vm / st_n, st_bar / st_n);
#endif

    // ---- flush: lane = feature column, register = mixture row; cov = S2 - 2 d S1 + d^2 S0, mean = S1 + (c + bias) S0
    const float *cen = centers + (size_t)j * D;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int mt = mt0 + i;
        if (mt >= n_mtiles) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float s0 = __shfl(S[i][(2 * D) >> 5][r], (lane & 32) + ((2 * D) & 31), 64);   // column 2D = the constant feature
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float s1 = __shfl_xor(S[i][ct][r], 1, 64);                           // odd neighbour: x' column of the same d
                const int cidx = ct * 32 + col;
                if (m < M && !(cidx & 1) && cidx < 2 * D) {
                    const int d = cidx >> 1;
                    const size_t o = ((size_t)j * Mpad + m) * D + d;
                    const double c = (double)cen[d], dl = means64[o] - c;
                    const double S0 = (double)s0, S1 = (double)s1, S2 = (double)S[i][ct][r];
                    const double vm = S1 + (c + bias) * S0;                        // Clustering.py:669-672
                    const double vc = S2 - 2.0 * dl * S1 + dl * dl * S0;           // Clustering.py:674-678
                    if (FRESH) {
                        st_mean[o] = vm;
                        st_cov[o] = vc;
                    } else {
                        st_mean[o] += vm;
                        st_cov[o] += vc;
                    }
                }
                if (m < M && cidx == 2 * D) {                                      // Clustering.py:665
                    if (FRESH) st_acc[(size_t)j * Mpad + m] = (double)S[i][ct][r];
                    else st_acc[(size_t)j * Mpad + m] += (double)S[i][ct][r];
                }
            }
        }
    }
    if (slice == 0 && wave == 0) {
        double v = (lane < 32) ? galpha : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) st_alpha[j] += v;                                           // Clustering.py:667
    }
}

}  // namespace

size_t pcl_acc16_image_bytes(int D) {
    switch (D) {
        case 39: return (size_t)Img<39>::NB * 1024;
        case 26: return (size_t)Img<26>::NB * 1024;
        case 13: return (size_t)Img<13>::NB * 1024;
        default: return 0;
    }
}

// states [first, first + ns) of the batch's accumulate order, whose tiles fit one image buffer: the tile bookkeeping and
// the producer (into buffer set `buf`) ...
int pcl_launch_acc16_produce(pcl_ctx *ctx, pcl_batch *b, int first, int ns, int max_tiles, int buf, hipStream_t stream) {
    if (ns == 0) return PCL_OK;
    const int *ws = b->d_work_states + first, *lo = b->d_seg_lo + first, *hi = b->d_seg_hi + first;
    hipLaunchKernelGGL(acc16_tiles_kernel, dim3(1), dim3(1024), 0, stream, lo, hi, b->acc_off, ns, b->acc16_tile_off[buf], b->acc16_state_flag[buf]);
    const int pgrid = std::min(max_tiles, std::max(ctx->cus, 1) * 16);
#define PRODUCE16(DD)                                                                                                         \
    hipLaunchKernelGGL((acc16_producer_kernel<DD>), dim3(pgrid), dim3(256), 0, stream, ctx->frames32, ctx->centers32, ctx->fscale, \
                       ctx->kzero, ns, ws, lo, hi, b->acc_off, b->acc_list, b->acc16_tile_off[buf], 0, reinterpret_cast<uint4 *>(b->acc16_images[buf]),  \
                       b->acc16_tile_mask[buf], b->acc16_state_flag[buf])
    switch (ctx->D) {
        case 39: PRODUCE16(39); break;
        case 26: PRODUCE16(26); break;
        case 13: PRODUCE16(13); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f16 accumulate kernel for D=%d", ctx->D);
    }
#undef PRODUCE16
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// ... and the consumer of the same group
int pcl_launch_acc16_consume(pcl_ctx *ctx, pcl_batch *b, int first, int ns, int buf, bool fresh, hipStream_t stream) {
    if (ns == 0) return PCL_OK;
    const int nmt = ctx->Mpad32 / 32, nslice = (nmt + AW * MT - 1) / (AW * MT);
    const int nblocks = ((ns + 7) / 8) * 8 * nslice;
    const int *ws = b->d_work_states + first;
#define CONSUME16F(DD, FR)                                                                                                    \
    hipLaunchKernelGGL((acc16_consumer_kernel<DD, FR>), dim3(nblocks), dim3(AW * 64), 0, stream,                              \
                       reinterpret_cast<const uint4 *>(b->acc16_images[buf]), reinterpret_cast<const uint4 *>(ctx->pm16f), ctx->centers32, \
                       ctx->mean64, ctx->M, ctx->Mpad, nmt, ns, ws, b->acc16_tile_off[buf], 0, 100.0, ctx->st_acc, ctx->st_alpha,      \
                       ctx->st_mean, ctx->st_cov)
#define CONSUME16(DD)                     \
    do {                                  \
        if (fresh) CONSUME16F(DD, true);  \
        else CONSUME16F(DD, false);       \
    } while (0)
    switch (ctx->D) {
        case 39: CONSUME16(39); break;
        case 26: CONSUME16(26); break;
        case 13: CONSUME16(13); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f16 accumulate kernel for D=%d", ctx->D);
    }
#undef CONSUME16
#undef CONSUME16F
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

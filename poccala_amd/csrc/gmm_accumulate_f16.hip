// gmm_accumulate_f16.hip -- E-step sufficient statistics, producer / consumer form (gfx950), the default f32 path.
//
// Same reference rows as gmm_accumulate.hip (A13: Clustering.GMM.update_acc, StatisticalModel/Clustering.py:653-680,
// called from LHMM.update_acc, StatisticalModel/LHMM.py:497-505): two chained contractions per 32-frame x 32-mixture
// tile of one state's list of surviving frames,
//   (1) D1[f][m] = Xe[f,:] . P[:,m]            the scoring GEMM (K = 2D + 2): log2 of w_m N_m(o_f), relative to K0_j
//       g[f][m]  = exp2(D1 + cf_f)             cf_f = log2e (ln gamma_f(j) - ln b_j(o_f)) + K0_j: gamma_f(j,m)
//   (2) S^T[:][m] += sum_f [x'^2_d, x'_d | 1]^T g[f][m]      raw moments S2, S1, S0 about the state centre
// Round 3: BOTH products run in the scoring kernel's two-piece f16 scheme on ONE operand image (round 2: product (2) in
// bf16, posteriors in two pieces and features in three, five cross terms = 30 MFMAs, and a second, feature-major copy of
// the tile in the image):
//   * product (1) as before (gmm_score_split.hip, variant 7: operands scaled by exact powers of two per (state, feature),
//     x = h1 + h2 carries 22 bits, products a2x1 + a1x2 + a1x1, the constant k'_m - K0_j folded into the spare K slot):
//     15 MFMAs, cf riding in as the C operand of the chain.
//   * product (2) = X^T . g with the SAME scaled f16 x2 frame operand, read back TRANSPOSED from the same LDS image
//     (ds_read_b64_tr_b16: a 16-lane group fetches 4 frames x 16 features and receives them feature-major), and the
//     posteriors in two f16 pieces: g1 X1 + g1 X2 + g2 X1 = 18 MFMAs (30 before).  f16 has 5 exponent bits where bf16 had
//     8, so a rarely responsible mixture (gamma ~ 1e-30) would vanish; every mixture therefore carries its own running
//     power-of-two scale E_m (online-softmax style): posteriors enter as exp2(D1 - E_m), a tile whose largest value would
//     leave (2^-24, 2^15] raises E_m and rescales that mixture's sums by the exact power of two (a wave-uniform branch
//     that is taken for the first tile and rarely afterwards); what falls below 2^-38 of the mixture's own largest
//     posterior so far is dropped, far below f32 resolution of the sums.  The sums are S^T (feature rows x mixture
//     columns): the mixture stays on the lane from product (1)'s accumulator through product (2) to the flush, so the
//     scale is a per-lane register and nothing crosses lanes.  Flush: S x 2^E_m / feature scale, in float64.
//   * the image of a 32-frame tile is 11 KiB (29 before): [2 pieces][5 k-steps] x 1 KiB in product (1)'s fragment order
//     with the 64-byte units (4 frames) of each side rotated by (side + 2 (s & 1)) -- row reads (ds_read_b128) and
//     transposed reads are both bank-conflict free (tools/ubench_trread.hip checks the maps with exact integers) -- plus
//     one block of per-frame coefficients.  The producer writes 2.6 x fewer bytes, the 8 slice workgroups of a state
//     stream 2.6 x fewer through LDS-DMA, and 8 tiles fit in flight.
//   * unchanged: tile images built once per state by the producer; the consumer software-pipelined across tiles
//     (posteriors(t) on the VALU beside product (1) of tile t+1, then product (2)(t) beside the reads for t+2);
//     a frame whose scaled feature leaves the f16 range is taken out of the image (g = 0), marked in its tile's mask and
//     added by the direct-form VALU kernel afterwards (gmm_accumulate.hip, masked mode).
// States whose centred expansion is ill conditioned (pcl_model_conditioning) never come here.
#include <algorithm>

#include "pcl_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

#ifndef PCL_ACC16_AW
#define PCL_ACC16_AW 4
#endif
#ifndef PCL_ACC16_NSLOT
#define PCL_ACC16_NSLOT 6
#endif
constexpr int AW = PCL_ACC16_AW;       // waves per consumer workgroup, one 32-mixture tile each
constexpr float FMAXH = 6.0e4f;        // what an f16 piece may hold (gmm_score_split.hip)
constexpr double LOG2E = 1.4426950408889634074;
constexpr float E_INIT = -1048576.f;   // "no posterior seen yet": any real tile raises it (an integer, so E stays one)
constexpr float G_TOP = 15.f;          // a tile whose largest log2 posterior exceeds E + G_TOP rescales (f16 max = 2^16)
constexpr float G_SET = 12.f;          // ... to put that largest value into (2^11, 2^12]

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
    static constexpr int NCT = (KS + 1) / 2;               // 32-row tiles of S^T: two k-steps' 16 columns (8 x'^2 | 8 x') each
    static constexpr int B1 = 0;                           // [piece 2][KS]: the scaled f16 x2 frame operand, see unit_of()
    static constexpr int BM = 2 * KS;                      // misc: cf in register order [2][16] f32 @0, gamma_f(j) [32] f64 @256
    static constexpr int NB = BM + 1;
    static_assert(D % 8 != 0, "the folded constant needs a spare K slot");
};
// Inside the 1-KiB block of k-step s: side (0: x'^2, 1: x') x 512 B, in it eight 64-byte units of 4 frames x 16 B (8 f16
// features of one frame), unit of frame group fg = frame >> 2 at position (fg + side + 2 (s & 1)) & 7.  A ds_read_b128 of
// the block (lane = side * 32 + frame) covers every bank once per 16-lane group; a ds_read_b64_tr_b16 of one 32-lane half
// touches four units {side 0, side 1} x {s even, s odd} of ONE frame group, which the rotation puts on four different
// bank quarters.
__host__ __device__ constexpr int unit_of(int fg, int side, int s) { return (fg + side + 2 * (s & 1)) & 7; }

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
    constexpr int KS = I::KS, XS = KS * 8 + 1;                   // row stride of the staged tile (odd: conflict-free column reads)
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
    // ---- the scaled f16 x2 frame operand: item = (k-step s, lane = side * 32 + frame): 8 features of one frame, both pieces
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
                val = side ? 0.f : 1.f;                          // x1: [1 | 0]  (a1: [k1 | 0], a2: [k2 | 1]: the sum is k1 + k2); product (2): S0
            }
            const _Float16 a = (_Float16)val;
            const _Float16 b = (d == D) ? (_Float16)0.f : (_Float16)(val - (float)a);
            h1[x] = __builtin_bit_cast(unsigned short, a);
            h2[x] = __builtin_bit_cast(unsigned short, b);
        }
        const int at = side * 32 + unit_of(f >> 2, side, s) * 4 + (f & 3);       // 16-byte position inside the block
        img[(I::B1 + 0 * KS + s) * 64 + at] = make_uint4(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16), h1[4] | ((unsigned)h1[5] << 16), h1[6] | ((unsigned)h1[7] << 16));
        img[(I::B1 + 1 * KS + s) * 64 + at] = make_uint4(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16), h2[4] | ((unsigned)h2[5] << 16), h2[6] | ((unsigned)h2[7] << 16));
    }
    if (ovf) atomicOr(&s_mask, ovf);
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
// consumer: AW waves x 32 mixtures of one state per workgroup; the slices of a state sit on block indices with equal
// residue mod 8 (one XCD's L2).
// FRESH: the statistics are all zero (first pass after pcl_stats_zero) and every (state, mixture) belongs to exactly one wave:
// the flush stores instead of read-modify-writing 7.7 GB of float64.
template <int D, bool FRESH>
__global__ __launch_bounds__(AW * 64, AW == 8 ? 1 : 2) void acc16_consumer_kernel(
    const uint4 *__restrict__ images, const uint4 *__restrict__ pm16f, const float *__restrict__ centers, const float *__restrict__ fscale,
    const double *__restrict__ means64, int M, int Mpad, int n_mtiles, int n_states, const int *__restrict__ work_states,
    const int *__restrict__ tile_off, int tile_base, double bias, double *__restrict__ st_acc, double *__restrict__ st_alpha,
    double *__restrict__ st_mean, double *__restrict__ st_cov, const int *__restrict__ npt, const int *__restrict__ nbad, const int *__restrict__ good_idx) {
    using I = Img<D>;
    constexpr int KS = I::KS, NCT = I::NCT, NB = I::NB;
    constexpr int NSLOT = PCL_ACC16_NSLOT;                       // tile t in slot t % NSLOT: t .. t + 2 being read, the rest landing
    constexpr int AHEAD = NSLOT - 1;                             // tile t + AHEAD is issued at the top of tile t (its slot held tile t - 1)
    __shared__ __attribute__((aligned(16))) uint4 slot[NSLOT][NB * 64];

    const int nslice = (n_mtiles + AW - 1) / AW;
    const int b = blockIdx.x;
    const int w = (b & 7) + 8 * (b / (8 * nslice));              // block b runs on XCD b % 8: all slices of a state on one XCD
    const int slice = (b >> 3) % nslice;
    if (w >= n_states) return;
    const int t0 = tile_off[w] - tile_base, t1 = tile_off[w + 1] - tile_base;
    if (t0 == t1) return;
    const int j = work_states[w];
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and the compiler must know: everything per-wave below
                                                                         // (DMA block loop, counted waits) is otherwise compiled as EXEC-masked vector control flow
    const int mt = slice * AW + wave;                            // this wave's m-tile
    const bool live = mt < npt[j];                               // (a wave past the tiles in use -- a split state's on-pipe mixtures sit at the front, round 6 -- only helps with the DMA)

    // parameters of this wave's m-tile: the scoring layout of variant 7 as it is (B operand of product (1))
    h8v pf[2][KS];
    {
        const uint4 *pq = pm16f + ((size_t)j * n_mtiles + (live ? mt : 0)) * (2 * KS * 64) + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < KS; ++s) pf[p][s] = __builtin_bit_cast(h8v, pq[(p * KS + s) * 64]);
    }
    f16v ST[NCT];                                                // S^T: rows = feature columns (registers), column = mixture (lane)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ST[ct][r] = 0.f;
    float E = E_INIT;                                            // this lane's mixture: sums are in units of 2^E
    double galpha = 0.0;

    auto dma = [&](int t) {                                      // tile t -> slot t % NSLOT; the NB blocks are dealt to the waves
        const uint4 *src = images + (size_t)t * (NB * 64);
        const unsigned int dst = __builtin_amdgcn_readfirstlane(lds_addr(&slot[t % NSLOT][0]));
#pragma unroll
        for (int k = 0; k < (NB + AW - 1) / AW; ++k) {
            const int p = wave + k * AW;
            if (p < NB) glds16(src + p * 64 + lane, dst + (unsigned int)p * 1024u);
        }
    };
    // product (1) row reads: lane = side * 32 + frame (side = half, frame = col); the unit rotation depends on s & 1
    const int row_even = half * 32 + unit_of(col >> 2, half, 0) * 4 + (col & 3);
    const int row_odd = half * 32 + unit_of(col >> 2, half, 1) * 4 + (col & 3);
    // (the frame fragments travel from one iteration to the next as plain 128-bit integers: carried as half vectors the
    //  compiler splits them into 16-bit halves at the loop edge and re-packs them with v_perm_b32, 80 VALU ops per tile)
    auto load1 = [&](int t, uint4 (&a1)[KS], uint4 (&a2)[KS]) {
        const uint4 *x1 = &slot[t % NSLOT][I::B1 * 64];
#pragma unroll
        for (int s = 0; s < KS; ++s) a2[s] = x1[(1 * KS + s) * 64 + ((s & 1) ? row_odd : row_even)];
#pragma unroll
        for (int s = 0; s < KS; ++s) a1[s] = x1[(0 * KS + s) * 64 + ((s & 1) ? row_odd : row_even)];
    };
    // The chain starts from the frames' coefficients cf (ln gamma - ln b in log2 units, -inf for a frame that is not in the
    // image): the posterior's exponent comes out of the matrix pipe complete; d comes in holding cf and goes out as D1.
    auto mfma1 = [&](const uint4 (&a1)[KS], const uint4 (&a2)[KS], f16v &d) {
#pragma unroll
        for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a2[s]), pf[0][s], d, 0, 0, 0);   // x2 a1
#pragma unroll
        for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a1[s]), pf[1][s], d, 0, 0, 0);   // x1 a2
#pragma unroll
        for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a1[s]), pf[0][s], d, 0, 0, 0);   // x1 a1
    };
    // product (2) transposed reads (A = X^T): the 16-lane group (lane >> 4) & 1 of a half takes k-step s = 2 ct + group; in
    // it lane 4 q + p supplies the address of frame row q, 8-byte chunk p (p >> 1 = side, p & 1 = which half of the 16 B) and
    // lane i receives column i (i < 8: x'^2 feature 8 s + i, i >= 8: x' feature 8 s + i - 8) of 4 frames.  The fragment of
    // k-step sp needs frames 16 sp + 4 h + {0..3} (elements 0..3) and + 8 (elements 4..7): two reads.  (KS odd: the last
    // group of the last column tile reads the block behind its piece -- in bounds, and its rows of S^T are never looked at.)
    const int tg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
    int tr_off[4];                                               // byte offset inside a piece for (sp, rd) -> k = 2 sp + rd, column tile 0
#pragma unroll
    for (int k = 0; k < 4; ++k)
        tr_off[k] = tg * 1024 + (tp >> 1) * 512 + unit_of(2 * k + half, tp >> 1, tg) * 64 + tq * 16 + 8 * (tp & 1);
    auto load2 = [&](int t, int sp, uint4 (&x)[2][NCT]) {
        const unsigned char *base = reinterpret_cast<const unsigned char *>(&slot[t % NSLOT][I::B1 * 64]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)(base + tr_off[2 * sp] + p * KS * 1024 + ct * 2048));
                const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)(base + tr_off[2 * sp + 1] + p * KS * 1024 + ct * 2048));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                x[p][ct] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
    };
    auto mfma2 = [&](const uint4 &g1, const uint4 &g2, const uint4 (&x)[2][NCT]) {
        // the NCT accumulators are independent, issued round robin; small cross terms first
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) ST[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, x[1][ct]), __builtin_bit_cast(h8v, g1), ST[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) ST[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, x[0][ct]), __builtin_bit_cast(h8v, g2), ST[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) ST[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, x[0][ct]), __builtin_bit_cast(h8v, g1), ST[ct], 0, 0, 0);
    };

    // Software pipeline, per wave, one basic block per tile t (the waves of a workgroup meet at ONE barrier per tile):
    //   region X   product (1) of tile t + 1  ||  VALU: D1 of tile t -> posteriors  ||  reads: product (2) fragments of tile t
    //   region Y   product (2) of tile t      ||  reads: cf and product (1) fragments of tile t + 2
    // LDS-DMA: tile t + AHEAD is issued at the top of tile t; at the end of tile t a wave waits (counted vmcnt) until its blocks
    // of tile t + 3 -- first read in tile t + 1 -- have landed and leaves the newer tiles (up to AHEAD - 3 of them) in flight.
    constexpr int MYB_HI = (NB + AW - 1) / AW, MYB_LO = NB / AW;  // blocks per tile this wave issues: waves < NB % AW one more
    const bool more_blocks = wave < NB % AW;
    for (int k = 0; k < AHEAD; ++k)
        if (t0 + k < t1) dma(t0 + k);
    // this wave's blocks have landed -- and, as a BUILTIN the compiler's wait-count pass sees, its own parameter loads too:
    // otherwise it re-waits for them (vmcnt(0)) at their first use inside the loop, every iteration, and drains the DMA
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();                                             // everyone's have
    // Two accumulator sets take turns (the tile loop is unrolled by two): while the posteriors of tile t are read out of one,
    // the chain of tile t + 1 runs in the other, which was loaded with that tile's cf a phase earlier -- no register copies.
    f16v dA, dB;
#pragma unroll
    for (int r = 0; r < 16; ++r) dA[r] = dB[r] = 0.f;
    uint4 a1[KS], a2[KS];
    auto load_cf = [&](int t, f16v &c) {                         // cf of tile t in the register order of D1 (lane half h, register r)
        const float4 *cfp = reinterpret_cast<const float4 *>(&slot[t % NSLOT][I::BM * 64]) + half * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = cfp[q];
            c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
        }
    };
    // largest log2 posterior of a tile for this lane's mixture (both lane halves hold frames of the SAME mixture).  It is taken
    // at the END of the step that ran the tile's product (1) -- beside product (2) of the tile before -- so that the next step
    // opens with one subtract and a wave-uniform branch instead of a dependent chain of ten instructions with nothing to overlap.
    auto tile_max = [&](const f16v &d) -> float {
        float m3 = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), d[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m3 = __builtin_fmaxf(__builtin_fmaxf(m3, d[r]), d[r + 1]);
        m3 = __builtin_fmaxf(m3, d[15]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m3), __float_as_uint(m3), false, false);
        return __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    float mx = -INFINITY;
    if (live) {
        load1(t0, a1, a2);
        load_cf(t0, dA);
        mfma1(a1, a2, dA);
        load_cf(t0 + 1, dB);                                     // (past the last tile: whatever the slot holds, nobody reads the product)
        load1(t0 + 1, a1, a2);
        mx = tile_max(dA);
    }
    auto tile_step = [&](int t, f16v &d1, f16v &dn) __attribute__((always_inline)) {
        const bool ahead = t + AHEAD < t1;
        if (ahead) dma(t + AHEAD);                               // its slot held tile t - 1: everyone left it at the last barrier
        const uint4 *cur = &slot[t % NSLOT][0];
        if (slice == 0 && wave == 0 && lane < 32) galpha += reinterpret_cast<const double *>(cur + I::BM * 64)[32 + lane];   // byte 256: gamma_f(j)
        if (live) {
            // ---- this mixture's scale: the largest log2 posterior of the tile (mx, taken at the end of the previous step)
            const float rel = mx - E;                            // (-inf when every frame of the tile is padding / masked: no trigger)
            if (__builtin_amdgcn_ballot_w64(rel > G_TOP) != 0ull) {      // rare, wave-uniform: the first tile, and a tile 8x above anything before it
                const float delta = rel > G_TOP ? __builtin_ceilf(rel) - G_SET : 0.f;
                const float f = __builtin_amdgcn_exp2f(-delta);  // exact power of two (0 when nothing real was summed yet)
                E += delta;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ST[ct][r] *= f;
            }
            uint4 x0[2][NCT], x1[2][NCT];
            // ---- region X: posteriors gamma_t(j,m) 2^-E (Clustering.py:660-661) of tile t in two f16 pieces = the B fragments of product (2)
            unsigned int u1[8], u2[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float ga = __builtin_amdgcn_exp2f(d1[2 * r] - E), gb = __builtin_amdgcn_exp2f(d1[2 * r + 1] - E);
                u1[r] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(ga, gb));
                const float ra = ga - (float)__builtin_bit_cast(_Float16, (unsigned short)(u1[r] & 0xffffu));
                const float rb = gb - (float)__builtin_bit_cast(_Float16, (unsigned short)(u1[r] >> 16));
                u2[r] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(ra, rb));
            }
            const uint4 g1a = make_uint4(u1[0], u1[1], u1[2], u1[3]), g1b = make_uint4(u1[4], u1[5], u1[6], u1[7]);
            const uint4 g2a = make_uint4(u2[0], u2[1], u2[2], u2[3]), g2b = make_uint4(u2[4], u2[5], u2[6], u2[7]);
            mfma1(a1, a2, dn);                                   // product (1) of tile t + 1, from its frames' coefficients
            load2(t, 0, x0);
            load2(t, 1, x1);
            // ---- region Y
            mfma2(g1a, g2a, x0);                                 // product (2) of tile t: S^T[feature][mixture] += Xe^T . g
            mfma2(g1b, g2b, x1);
            load_cf(t + 2, d1);                                  // (the posteriors of tile t have been read out of d1)
            load1(t + 2, a1, a2);
            mx = tile_max(dn);                                   // for the next step (past the last tile: never used)
#ifndef PCL_ACC16_NOSGB           // spread the fragment reads between the MFMAs instead of one burst in front of the first MFMA that needs them
            {
                constexpr int RX = 8 * NCT, RY = 4 + 2 * KS, NY = 6 * NCT, PERX = (RX + 2) / 3, PERY = (RY + NY - 1) / NY;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, PERX, 0);    // DS read
                }
#pragma unroll
                for (int i = 0; i < NY; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, PERY, 0);
                }
            }
#endif
        }
        // this wave's blocks of tile t + 3 have landed; those of the newer tiles stay in flight
        static_assert(MYB_HI * (AHEAD - 3) <= 16 && AHEAD >= 3, "counted waits below");
        {
            const int newest = min(t + AHEAD, t1 - 1);                           // the last tile issued so far
            const int keep = max(0, newest - (t + 3)) * (more_blocks ? MYB_HI : MYB_LO);       // wave-uniform
            switch (keep) {
#define W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
#undef W
                default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();                            // (raw: __syncthreads would drain the DMA in flight) tile t + 3 is there for everyone; slot t % NSLOT is free
    };
    for (int t = t0; t < t1; t += 2) {
        tile_step(t, dA, dB);
        if (t + 1 < t1) tile_step(t + 1, dB, dA);
    }

    // ---- flush: lane = mixture, register = feature row of S^T.  Row c of column tile ct <-> k-step s = 2 ct + (c >> 4), i = c & 15:
    //      x'^2 of feature 8 s + i (i < 8) or x' of feature 8 s + i - 8; c = (r & 3) + 8 (r >> 2) + 4 h, so a lane holds S2 of
    //      feature d in register r (r & 4 == 0) and S1 of the same d in register r + 4.  S0 is row (s = KS - 1, i = D & 7) of the last
    //      tile.  cov = S2 - 2 dl S1 + dl^2 S0, mean = S1 + (c + bias) S0, all times 2^E / the feature's power-of-two scale.
    if (!live) return;
    // the mixture this lane's layout row stands for: the row itself, or -- a split state, whose on-pipe mixtures are compacted to the front of
    // its tiles -- the row-th entry of the state's on-pipe list; rows beyond the list are padding
    const int row_ = mt * 32 + col, nb_ = nbad[j];
    const int m = good_idx ? ((row_ < M - nb_) ? (nb_ ? good_idx[(size_t)j * Mpad + row_] : row_) : M) : row_;      // (good_idx == nullptr: PCL_COMPACT_MAIN=0)
    constexpr int C0 = ((KS - 1) & 1) * 16 + (D & 7), CT0 = (KS - 1) >> 1;          // where S0 lives: row C0 of tile CT0
    constexpr int R0 = (C0 & 3) + 4 * (C0 >> 3), H0 = (C0 >> 2) & 1;
    float s0f = ST[CT0][R0];
    {   // v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its second: afterwards the
        // first result holds the LOWER half's values in both halves, the second result the UPPER half's
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0f), __float_as_uint(s0f), false, false);
        s0f = __uint_as_float(H0 ? sw[1] : sw[0]);
    }
    const int Ei = (int)E;
    const double S0 = ldexp((double)s0f, Ei);
    if (m < M) {
        const float *cen = centers + (size_t)j * D;
        const float *fs = fscale + (size_t)j * 2 * (KS * 8);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r & 4) continue;                             // the S1 partner of register r - 4
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half, s = 2 * ct + (c >> 4), d = 8 * s + (c & 7);
                if (s >= KS || d >= D) continue;
                const size_t o = ((size_t)j * Mpad + m) * D + d;
                const double cc = (double)cen[d], dl = means64[o] - cc;
                const double S2 = ldexp((double)ST[ct][r], Ei) / (double)fs[d], S1 = ldexp((double)ST[ct][r + 4], Ei) / (double)fs[KS * 8 + d];
                const double vm = S1 + (cc + bias) * S0;                           // Clustering.py:669-672
                // Clustering.py:674-678: a sum of gamma (o - mu)^2, never negative.  The raw-moment form can come out a hair below zero
                // (measured down to -2.6e-6 acc, tests/test_gpu_fuzz_estep.py) when a mixture's few frames sit on its mean in one
                // dimension; as ln(cov_acc) in a reference-format accumulator file that would be a NaN.  Each pass's share is clamped.
                const double vc = fmax(S2 - 2.0 * dl * S1 + dl * dl * S0, 0.0);
                if (FRESH) {
                    st_mean[o] = vm;
                    st_cov[o] = vc;
                } else {
                    st_mean[o] += vm;
                    st_cov[o] += vc;
                }
            }
        if (half == 0) {                                                           // Clustering.py:665 (both halves hold the same S0)
            if (FRESH) st_acc[(size_t)j * Mpad + m] = S0;
            else st_acc[(size_t)j * Mpad + m] += S0;
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
        case 47: return (size_t)Img<47>::NB * 1024;
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
    const int *ws = b->ctx->acc.d_work_states + first, *lo = b->ctx->acc.d_seg_lo + first, *hi = b->ctx->acc.d_seg_hi + first;
    hipLaunchKernelGGL(acc16_tiles_kernel, dim3(1), dim3(1024), 0, stream, lo, hi, b->ctx->acc.acc_off, ns, b->ctx->acc.acc16_tile_off[buf], b->ctx->acc.acc16_state_flag[buf]);
    // (max_tiles = 0: no frame of the group's states survived -- a batch scored by a collapsed model can have such groups; a grid of 0 is an
    //  invalid launch, tools/sweep_fuzz.py seed 514 -- one workgroup then finds nothing to do)
    const int pgrid = std::max(1, std::min(max_tiles, std::max(ctx->cus, 1) * 16));
#define PRODUCE16(DD)                                                                                                         \
    hipLaunchKernelGGL((acc16_producer_kernel<DD>), dim3(pgrid), dim3(256), 0, stream, ctx->frames32, ctx->centers32, ctx->fscale, \
                       ctx->kzero, ns, ws, lo, hi, b->ctx->acc.acc_off, b->ctx->acc.acc_list, b->ctx->acc.acc16_tile_off[buf], 0, reinterpret_cast<uint4 *>(b->ctx->acc.acc16_images[buf]),  \
                       b->ctx->acc.acc16_tile_mask[buf], b->ctx->acc.acc16_state_flag[buf])
    switch (ctx->D) {
        case 47: PRODUCE16(47); break;
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
    const int nmt = ctx->Mpad32 / 32, nslice = (nmt + AW - 1) / AW;
    const int nblocks = ((ns + 7) / 8) * 8 * nslice;
    const int *ws = b->ctx->acc.d_work_states + first;
#define CONSUME16F(DD, FR)                                                                                                    \
    hipLaunchKernelGGL((acc16_consumer_kernel<DD, FR>), dim3(nblocks), dim3(AW * 64), 0, stream,                              \
                       reinterpret_cast<const uint4 *>(b->ctx->acc.acc16_images[buf]), reinterpret_cast<const uint4 *>(ctx->pm16f), ctx->centers32, ctx->fscale, \
                       ctx->mean64, ctx->M, ctx->Mpad, nmt, ns, ws, b->ctx->acc.acc16_tile_off[buf], 0, 100.0, ctx->st_acc, ctx->st_alpha,      \
                       ctx->st_mean, ctx->st_cov, ctx->d_npt, ctx->d_nbad, (ctx->compact_main && ctx->D <= 48) ? ctx->d_good_idx : (const int *)nullptr)
#define CONSUME16(DD)                     \
    do {                                  \
        if (fresh) CONSUME16F(DD, true);  \
        else CONSUME16F(DD, false);       \
    } while (0)
    switch (ctx->D) {
        case 47: CONSUME16(47); break;
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

// gmm_score_split.hip -- GMM scoring with f32-accurate products on the bf16 / f16 matrix pipe (gfx950).
//
// Three kernels, one idea (split every f32 operand into exact low-precision pieces, keep the cross products that matter):
//   gmm_score_split_kernel      three bf16 pieces, six products                        PCL_SCORE_VARIANT=4
//   gmm_score_split16_kernel    two f16 pieces, three products; FOLD = false: constants on one extra bf16 MFMA (5),
//                               FOLD = true: constants in the spare K slot of the f16 passes (7, the default)
//   gmm_score_split16x_kernel   the two-piece f16 scheme on 16x16x32 MFMAs, one K axis     (6)
// The text below introduces the scheme on the bf16 kernel; each of the others starts with what it changes.
//
// Same reference rows as gmm_score.hip (A1/A4/A6: util.py:20-31, Clustering.py:740-767, LHMM.py:163-187) and the
// same contraction as gmm_score_mfma.hip:
//     v[f,m] = k'_m + sum_d ( a_md x'_fd^2 + b_md x'_fd ) - ref_f,      x' = x - c_j.
//
// Why not the f32-input MFMA.  On gfx950 v_mfma_f32_32x32x2_f32 runs at the VALU's rate (64 FLOP/clk/SIMD) AND
// blocks the VALU while it runs (tools/ubench_hybrid.hip: a v_fma wave beside an f32-MFMA wave makes ~5 % progress),
// so that kernel is bounded by 157 TFLOP/s minus its own log-sum-exp.  The bf16 pipe is 16x faster and a separate
// pipe.  Every f32 number is EXACTLY the sum of three bf16 numbers (3 x 8 significand bits, round-to-nearest
// residuals), and a bf16 x bf16 product is exact in f32, so
//     a x = (a1 + a2 + a3)(x1 + x2 + x3) = a1x1 + (a1x2 + a2x1) + (a1x3 + a2x2 + a3x1) + O(2^-24 |a x|)
// with f32 accumulation: six bf16 MFMAs per K-step instead of one f32 MFMA, 6/16 of the matrix-pipe time, and
// the dropped cross terms are below f32's own rounding.  Measured against float64 (tools/ubench_split.hip, sum of
// |terms| = 205): this scheme 1.8e-5, the exact f32 FMA chain 3.5e-5 (fewer, wider partial sums) -- it is not a
// reduced-precision path, and the parity tests hold it to the same tolerances.
//
// Mapping (v_mfma_f32_32x32x16_bf16: D[32 mixtures x 32 frames] += A[32 x 16] B[16 x 32]).  K-step s, lane l
// (r = l&31, h = l>>5), element j = feature d = 8s + j of the side h selects:
//   B (frames):     h = 0: x'_d^2 (1 at d = D),  h = 1: x'_d (-ref at d = D)   -- VGPR-resident, 3 pieces
//   A (parameters): h = 0: a_md   (k'_m at d = D), h = 1: b_md (1 at d = D)    -- layout [m-tile][piece][s][64][8],
//                   staged per m-tile in LDS by LDS-DMA (double buffered) and shared by the 4 waves.
// Pass order a3x1, a2x2, a2x1, a1x3, a1x2, a1x1: the small terms first, one parameter piece live at a time.
// Everything after the accumulator (reference-shifted log-sum-exp, slow path, merge) is gmm_score_mfma.hip's.
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

#ifndef PCL_SPLIT_WG
#define PCL_SPLIT_WG 256    // threads per workgroup = 64 x (waves sharing one LDS copy of the A tile)
#endif
#ifndef PCL_SPLIT_MINW
#define PCL_SPLIT_MINW 2    // __launch_bounds__ waves per SIMD (bf16 x3 kernel: 195 VGPRs)
#endif
constexpr int WG = PCL_SPLIT_WG;
#ifndef PCL_SPLIT_NT
#define PCL_SPLIT_NT 2      // frame column tiles (32 frames) per wave
#endif
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
    p1 = (__bf16)x;
    float r = x - (float)p1;
    p2 = (__bf16)r;
    r -= (float)p2;
    p3 = (__bf16)r;
}

template <int D, int NT>
__global__ __launch_bounds__(WG, PCL_SPLIT_MINW * 256 / WG > 0 ? PCL_SPLIT_MINW * 256 / WG : 1) void gmm_score_split_kernel(const float *__restrict__ frames, const uint4 *__restrict__ pm,
                                                                const float *__restrict__ centers, int n_mtiles,
                                                                const ScoreTile *__restrict__ tiles,
                                                                const ScoreSeg *__restrict__ segs, double *__restrict__ out) {
    constexpr int KS8 = (D + 8) / 8;       // K-steps of 16 (8 features per half-wave): D features + the constant slot
    constexpr int CH = 3 * KS8;            // 1-KiB chunks (64 lanes x 16 B) per m-tile
    constexpr int SC = D / 8, JC = D % 8;  // where the constant slot sits
    const ScoreTile tile = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    if (tile.seg_lo >= tile.seg_hi) return;   // padding tile of the XCD-aware order
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * NT * 32 < vend;   // a wave past the end still helps staging

    // ---- B operand: this lane's frames, centred (squared on the low half-wave), split into three bf16 pieces
    bf8v xb[NT][3][KS8];
    long long oidx[NT];
    bool valid[NT];
    const float *cen = centers + (size_t)tile.state * D;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        int v = tile.vstart + (wave * NT + c) * 32 + col;
        valid[c] = v < vend;
        if (!valid[c]) v = tile.vstart;
        // last segment with vstart <= v: almost always seg0 or its successor (two dependent loads instead of the
        // ~log2(#segments) of a full search, at the start of every tile), the search only for what is left
        int lo = tile.seg0, hi = tile.seg_hi - 1;
        if (lo < hi && segs[lo + 1].vstart <= v) {
            ++lo;
            if (lo < hi && segs[lo + 1].vstart <= v) {
                ++lo;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (segs[mid].vstart <= v) lo = mid; else hi = mid - 1;
                }
            }
        }
        const ScoreSeg sg = segs[lo];
        const long long t = v - sg.vstart;
        const float *fp = frames + (sg.frame0 + t) * D;
#pragma unroll
        for (int s = 0; s < KS8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * s + j;
                float val = 0.f;
                if (d < D) {
                    const float xc = fp[d] - cen[d];
                    val = half ? xc : xc * xc;
                } else if (d == D) {
                    val = half ? 0.f : 1.f;     // -ref (0 so far) | the constant's multiplier
                }
                __bf16 p1, p2, p3;
                split3(val, p1, p2, p3);
                xb[c][0][s][j] = p1;
                xb[c][1][s][j] = p2;
                xb[c][2][s][j] = p3;
            }
        oidx[c] = sg.out0 + t * (long long)sg.out_stride;
    }

    float sm[NT], ref[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        sm[c] = 0.f;
        ref[c] = 0.f;
    }

    __shared__ __attribute__((aligned(16))) uint4 abuf[2][CH * 64];
    const uint4 *pstate = pm + (size_t)tile.state * n_mtiles * (CH * 64);
    auto dma = [&](int buf, int mt) {
        const uint4 *src = pstate + (size_t)mt * (CH * 64);
        for (int p = wave; p < CH; p += WG / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)&abuf[buf][p * 64], 16, 0, 0);
    };

    auto process = [&](int mt) {
        const uint4 *ab = abuf[mt & 1];
        f16v acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        }
        auto pass = [&](int pa, int pb) {
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const bf8v a = *reinterpret_cast<const bf8v *>(&ab[(pa * KS8 + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[c][pb][s], acc[c], 0, 0, 0);
            }
        };
        pass(2, 0);
        pass(1, 1);
        pass(1, 0);
        pass(0, 2);
        pass(0, 1);
        pass(0, 0);
#ifdef PCL_DIAG_NOLSE
#pragma unroll
        for (int c = 0; c < NT; ++c) sm[c] += acc[c][0] + acc[c][15];   // diagnostic build: no log-sum-exp work (wrong results)
        return;
#endif
        // reference-shifted log-sum-exp, see gmm_score_mfma.hip; "log zero" (zero-weight and padding mixtures) is the
        // finite sentinel -3e38 here because an infinity would meet a zero piece of the other operand
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            f2v e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) e[r] = f2v{__builtin_amdgcn_exp2f(acc[c][2 * r]), __builtin_amdgcn_exp2f(acc[c][2 * r + 1])};
            const f2v t0 = (e[0] + e[1]) + (e[2] + e[3]), t1 = (e[4] + e[5]) + (e[6] + e[7]);
            const f2v t = t0 + t1;
            const float snew = sm[c] + (t.x + t.y);
            if (mt == 0 || __any(!(snew < 3.0e38f))) {
                float gm = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
                const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));   // max over the frame's 32 mixtures
                float s = sm[c];
                if ((mt == 0 || gp > 0.f) && gp > -1.0e37f) {
                    s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-gp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= gp;
                    ref[c] += gp;
                    __bf16 p1, p2, p3;
                    split3(-ref[c], p1, p2, p3);                                // exact: the pipe subtracts ref itself
                    if (half) {
                        xb[c][0][SC][JC] = p1;
                        xb[c][1][SC][JC] = p2;
                        xb[c][2][SC][JC] = p3;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
                sm[c] = s;
            } else {
                sm[c] = snew;
            }
        }
    };
    dma(0, 0);
    for (int mt = 0; mt < n_mtiles; ++mt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile mt have landed
        __syncthreads();                                    // everyone's pieces have; buffer (mt+1)&1 is free
        if (mt + 1 < n_mtiles) dma((mt + 1) & 1, mt + 1);
        if (wave_active) process(mt);
    }
    constexpr double LN2 = 0.693147180559945309417232121458;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const double S = (double)sm[c] + (double)__shfl_xor(sm[c], 32, 64);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2 * ((double)ref[c] + ::log2(S)) : -INFINITY;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Two-way f16 split: half the matrix work of the bf16 scheme.  f16 has 11 significand bits, so x = h1 + h2 carries 22
// and a x = a1x1 + a1x2 + a2x1 + O(2^-22 |a x|): three f16 MFMAs per K-step.  f16's narrow exponent range is handled
// by exact power-of-two scaling per (state, feature): A'' = coef 2^-e with max_m |A''| in [1, 2), B'' = feature 2^e
// (model_derive.hip writes 2^e next to the layout); subnormal second pieces are honoured by the f16 MFMA
// (tools/ubench_f16denorm.hip), so small features keep an ABSOLUTE error of 2^-25.  What does not fit f16 stays out
// of it: the constant k'_m, the reference shift -ref_f and the finite "log zero" ride ONE extra bf16 MFMA per tile
// (three pieces each, exact), and a frame whose scaled feature exceeds 6e4 (|x - c| beyond ~300 sigma of the tightest
// mixture) raises its tile's flag: the direct-form kernel then rescores flagged tiles in the same call
// (pcl_launch_score_fixup), so no input sees an overflowed result.
#ifndef PCL_SPLIT16_NT
#define PCL_SPLIT16_NT 2
#endif
#ifndef PCL_SPLIT16_MINW
#define PCL_SPLIT16_MINW 3    // waves per SIMD (f16 x2 kernel: fits 168 VGPRs)
#endif
#ifndef PCL_SPLIT16_MTS
#define PCL_SPLIT16_MTS 1     // m-tiles staged per workgroup barrier (2 and 4 measured no faster)
#endif
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

template <int D, int NT, bool FOLD>
__global__ __launch_bounds__(WG, PCL_SPLIT16_MINW * 256 / WG > 0 ? PCL_SPLIT16_MINW * 256 / WG : 1) void gmm_score_split16_kernel(
    const float *__restrict__ frames, const uint4 *__restrict__ pm, const float *__restrict__ fscale, const float *__restrict__ centers,
    int n_mtiles, const ScoreTile *__restrict__ tiles, const ScoreSeg *__restrict__ segs, double *__restrict__ out,
    int *__restrict__ flags, const double *__restrict__ kzero) {
    // FOLD (variant 7): no constant MFMA.  The spare slot d = D of each side carries the constants in f16:
    //   a1: [k1 | 0]   a2: [k2 | 1]   x1: [1 | -ref']   x2: [0 | 0]   ->  a2 x1 + a1 x2 + a1 x1 = k1 + k2 - ref'
    // with k1 + k2 = k'_m - K0_j (K0_j = max_m k'_m, added back in f64 at the end, so the 22 bits go to a small number)
    // and ref' the f16-rounded reference (any nearby value serves).  Log zero is -6e4; a frame whose reference leaves
    // the f16 range is flagged for the direct-form fix-up like an out-of-range feature.
    static_assert(!FOLD || D % 8 != 0, "the folded constants need a spare slot");
    constexpr int KS8 = (D + 7) / 8;       // K-steps of 16 over the 2D features (8 per half-wave)
    constexpr int CH = FOLD ? 2 * KS8 : 2 * KS8 + 1;   // 1-KiB chunks per m-tile: two f16 pieces (+ the bf16 constant chunk)
    constexpr int SC = D / 8, JC = D % 8;  // the spare slot
    constexpr float FMAXH = 6.0e4f;
    const ScoreTile tile = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    if (tile.seg_lo >= tile.seg_hi) {       // padding tile of the XCD-aware order
        if (threadIdx.x == 0) flags[blockIdx.x] = 0;
        return;
    }
    __shared__ int s_ovf;
    if (threadIdx.x == 0) s_ovf = 0;
    __syncthreads();
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * NT * 32 < vend;

    // ---- B operand: scaled features of this lane's frames in two f16 pieces, and the bf16 constant fragment
    h8v xb[NT][2][KS8];
    bf8v xc8[NT];
    long long oidx[NT];
    bool valid[NT];
    const float *cen = centers + (size_t)tile.state * D;
    const float *fs = fscale + ((size_t)tile.state * 2 + half) * (KS8 * 8);
    bool ovf = false;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        int v = tile.vstart + (wave * NT + c) * 32 + col;
        valid[c] = v < vend;
        if (!valid[c]) v = tile.vstart;
        // last segment with vstart <= v: almost always seg0 or its successor (two dependent loads instead of the
        // ~log2(#segments) of a full search, at the start of every tile), the search only for what is left
        int lo = tile.seg0, hi = tile.seg_hi - 1;
        if (lo < hi && segs[lo + 1].vstart <= v) {
            ++lo;
            if (lo < hi && segs[lo + 1].vstart <= v) {
                ++lo;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (segs[mid].vstart <= v) lo = mid; else hi = mid - 1;
                }
            }
        }
        const ScoreSeg sg = segs[lo];
        const long long t = v - sg.vstart;
        const float *fp = frames + (sg.frame0 + t) * D;
#pragma unroll
        for (int s = 0; s < KS8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * s + j;
                float val = 0.f;
                if (d < D) {
                    const float xc = fp[d] - cen[d];
                    val = (half ? xc : xc * xc) * fs[d];
                    ovf |= __builtin_fabsf(val) > FMAXH;
                    val = __builtin_fminf(__builtin_fmaxf(val, -FMAXH), FMAXH);
                }
                if (FOLD && d == D) val = half ? 0.f : 1.f;      // x1: [1 | -ref' (0 so far)]
                const _Float16 h1 = (_Float16)val;
                xb[c][0][s][j] = h1;
                xb[c][1][s][j] = (_Float16)(val - (float)h1);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) xc8[c][j] = (__bf16)((half == 0 && j < 3) ? 1.f : 0.f);   // 1 1 1 -ref1 -ref2 -ref3 0 0
        oidx[c] = sg.out0 + t * (long long)sg.out_stride;
    }
    if (__any(ovf) && lane == 0) s_ovf = 1;

    float sm[NT], ref[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        sm[c] = 0.f;
        ref[c] = 0.f;
    }
    bool ref_ovf = false;

    // MTS m-tiles per LDS stage: one workgroup barrier per MTS x 32 mixtures
    constexpr int MTS = PCL_SPLIT16_MTS;
    __shared__ __attribute__((aligned(16))) uint4 abuf[2][MTS * CH * 64];
    const uint4 *pstate = pm + (size_t)tile.state * n_mtiles * (CH * 64);
    auto dma = [&](int buf, int stage) {
        const uint4 *src = pstate + (size_t)stage * (MTS * CH * 64);
        const int nch = min(MTS, n_mtiles - stage * MTS) * CH;
        for (int p = wave; p < nch; p += WG / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)&abuf[buf][p * 64], 16, 0, 0);
    };

#ifdef PCL_SPLIT_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_t = 0;
#define SSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; }
#else
#define SSTAMP(k)
#endif
    auto process = [&](int mt) {
        const uint4 *ab = &abuf[(mt / MTS) & 1][(mt % MTS) * (CH * 64)];
        f16v acc[NT];
        if constexpr (FOLD) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        } else {
            const bf8v ac = *reinterpret_cast<const bf8v *>(&ab[(2 * KS8) * 64 + lane]);   // k'1 k'2 k'3 1 1 1 0 0
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                f16v z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac, xc8[c], z, 0, 0, 0);
            }
        }
        auto pass = [&](int pa, int pb) {
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const h8v a = *reinterpret_cast<const h8v *>(&ab[(pa * KS8 + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[c][pb][s], acc[c], 0, 0, 0);
            }
        };
#ifdef PCL_SPLIT_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        pass(1, 0);
        pass(0, 1);
        pass(0, 0);
        SSTAMP(1)
#ifdef PCL_SPLIT_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef PCL_DIAG_NOLSE
#pragma unroll
        for (int c = 0; c < NT; ++c) sm[c] += acc[c][0] + acc[c][15];
        return;
#endif
#ifndef PCL_SPLIT16_ONE_CHECK      // one wave-uniform slow-path test for both column tiles measured no faster (17.27 vs 17.20 ms)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#ifndef PCL_SPLIT_PKADD
            float es[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) es[r] = __builtin_amdgcn_exp2f(acc[c][r]);
            // plain v_add_f32 tree: under -O3 the compiler SLP-packs such adds into v_pk_add_f32, which costs far more than
            // its issue slot beside MFMAs (in-kernel stamps: log-sum-exp phase 1830 -> 1230 cycles per m-tile), so this
            // file is built with -fno-slp-vectorize (Makefile).  (Pinning the adds with inline asm instead returned wrong
            // sums: the hazard recogniser does not cover a transcendental result consumed inside an asm block.)
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int r = 0; r < w; ++r) es[r] += es[r + w];
            const float snew = sm[c] + es[0];
#else
            f2v e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) e[r] = f2v{__builtin_amdgcn_exp2f(acc[c][2 * r]), __builtin_amdgcn_exp2f(acc[c][2 * r + 1])};
            const f2v t0 = (e[0] + e[1]) + (e[2] + e[3]), t1 = (e[4] + e[5]) + (e[6] + e[7]);
            const f2v t = t0 + t1;
            const float snew = sm[c] + (t.x + t.y);
#endif
            if (mt == 0 || __any(!(snew < 3.0e38f))) {
                float gm = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
                const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));
                float s = sm[c];
                if constexpr (FOLD) {
                    if (mt == 0 && !(gp > -5.0e4f)) ref_ovf = true;      // log zero everywhere, or out of the f16 constants' reach
                    if ((mt == 0 || gp > 0.f) && gp > -5.0e4f) {
                        // the reference the pipe subtracts is the f16-rounded one: shift by what it actually moves
                        const _Float16 r1 = (_Float16)__builtin_fminf(__builtin_fmaxf(-(ref[c] + gp), -FMAXH), FMAXH);
                        const float nref = -(float)r1, dl = nref - ref[c];
                        ref_ovf |= __builtin_fabsf(nref) > 5.0e4f;
                        s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-dl);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[c][r] -= dl;
                        ref[c] = nref;
                        if (half) xb[c][0][SC][JC] = r1;
                    }
                } else if ((mt == 0 || gp > 0.f) && gp > -1.0e37f) {
                    s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-gp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= gp;
                    ref[c] += gp;
                    __bf16 p1, p2, p3;
                    split3(-ref[c], p1, p2, p3);
                    if (half == 0) {
                        xc8[c][3] = p1;
                        xc8[c][4] = p2;
                        xc8[c][5] = p3;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
                sm[c] = s;
            } else {
                sm[c] = snew;
            }
        }
#else
        // fast path for both column tiles first, ONE wave-uniform test for the slow path
        float snew[NT];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float es[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) es[r] = __builtin_amdgcn_exp2f(acc[c][r]);
            // plain v_add_f32 tree: this file is built with -fno-slp-vectorize (Makefile), see the note further up
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int r = 0; r < w; ++r) es[r] += es[r + w];
            snew[c] = sm[c] + es[0];
            bad |= !(snew[c] < 3.0e38f);
        }
        if (mt != 0 && !__any(bad)) {
#pragma unroll
            for (int c = 0; c < NT; ++c) sm[c] = snew[c];
        } else {
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                float gm = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
                const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));
                float s = sm[c];
                if ((mt == 0 || gp > 0.f) && gp > -1.0e37f) {
                    s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-gp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= gp;
                    ref[c] += gp;
                    __bf16 p1, p2, p3;
                    split3(-ref[c], p1, p2, p3);
                    if (half == 0) {
                        xc8[c][3] = p1;
                        xc8[c][4] = p2;
                        xc8[c][5] = p3;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
                sm[c] = s;
            }
        }
#endif
        SSTAMP(2)
    };
    const int n_stages = (n_mtiles + MTS - 1) / MTS;
    dma(0, 0);
#ifdef PCL_SPLIT_STAMPS
    st_t = __builtin_amdgcn_s_memtime();
#endif
    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage st have landed
        SSTAMP(3)
        __syncthreads();                                    // everyone's have; the other buffer is free
        SSTAMP(0)
        if (st == 0 && threadIdx.x == 0) flags[blockIdx.x] = s_ovf;
        if (st + 1 < n_stages) dma((st + 1) & 1, st + 1);
        if (wave_active) {
            const int mend = min(n_mtiles, (st + 1) * MTS);
            for (int mt = st * MTS; mt < mend; ++mt) process(mt);
        }
    }
#ifdef PCL_SPLIT_STAMPS
    if (blockIdx.x == 800 && lane == 0)
        printf("wave %d: per m-tile: dma-wait %llu  barrier %llu  dma-issue+lds+mfma %llu  lse %llu (memtime ticks)\n", wave,
               st_acc[3] / n_mtiles, st_acc[0] / n_mtiles, st_acc[1] / n_mtiles, st_acc[2] / n_mtiles);
#endif
    constexpr double LN2 = 0.693147180559945309417232121458;
    const double k0 = FOLD ? kzero[tile.state] : 0.0;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const double S = (double)sm[c] + (double)__shfl_xor(sm[c], 32, 64);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2 * ((double)ref[c] + k0 + ::log2(S)) : -INFINITY;
    }
    if constexpr (FOLD) {
        if (__any(ref_ovf) && lane == 0) s_ovf = 1;
        __syncthreads();
        if (threadIdx.x == 0 && s_ovf) flags[blockIdx.x] = 1;     // (the flag of the features was stored after the first barrier)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same two-way f16 split on v_mfma_f32_16x16x32_f16 (variant 6).  Under this kernel the chip is power limited
// (1.7-1.8 GHz, matrix pipe ~65 % busy), and on random operands the 16x16x32 shape delivers ~1.14 x the FLOP/s of
// 32x32x16 at equal cycles (tools/ubench_shape.hip: 1520 vs 1330 TFLOP/s with this kernel's LDS re-reads and
// exponentials; 1965 vs 1900 on all-zero operands: it is the clock, not the cycles).  To keep the cycles equal the three
// passes and the constants share ONE K axis, [a2 x1 | a1 x2 | a1 x1 | constants | 0] = 3 x 80 + 8 -> 256 for D = 39
// (eight K-steps of 32 per 16 x 16 tile, 64 MFMAs of 16 cycles per 32 mixtures x 64 frames: the 1024 cycles of the
// 32x32x16 kernel), which means the constants ride in f16 as well: k'_m, -ref_f in three f16 pieces each (33 bits), log
// zero = -6e4, and a state whose real k' exceeds 5e4 (model_derive.hip) or a frame whose reference drops below -5e4
// (flag + fix-up) leaves this kernel like an out-of-range feature does.
// Lane l = 16 g + r: A fragment = mixture r, K block 4 s + g; B fragment = frame r, same block; accumulator register
// q = mixture 4 g + q of frame r: a lane owns 8 of a frame's 32 mixtures per m-tile, the four lane groups are merged
// at the end (and in the slow path that raises ref).
typedef float f4v __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(WG, 2) void gmm_score_split16x_kernel(
    const float *__restrict__ frames, const uint4 *__restrict__ pm, const float *__restrict__ fscale, const float *__restrict__ centers,
    int n_mtiles, const ScoreTile *__restrict__ tiles, const ScoreSeg *__restrict__ segs, double *__restrict__ out,
    int *__restrict__ flags) {
    constexpr int SEG8 = (2 * D + 7) / 8;              // 8-element blocks per pass segment
    constexpr int CT = 3 * SEG8;                       // the constants block
    constexpr int NKS = (CT + 1 + 3) / 4;              // K-steps of 32
    constexpr int SCT = CT / 4, GCT = CT % 4;          // K-step and lane group that hold the constants block
    constexpr int KS8f = (D + 7) / 8;                  // fscale row stride / 8
    constexpr int NC = 4;                              // frame sub-tiles (16 frames) per wave
    constexpr int XS = D + 1;                          // LDS row stride of the staged frames
    constexpr float FMAXH = 6.0e4f;
    static_assert(WG == 256, "4 waves x 64 frames");
    const ScoreTile tile = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    if (tile.seg_lo >= tile.seg_hi) {                  // padding tile of the XCD-aware order
        if (threadIdx.x == 0) flags[blockIdx.x] = 0;
        return;
    }
    __shared__ int s_ovf;
    __shared__ float xs[4][64 * XS];                   // centred frames of each wave
    __shared__ long long oidx_s[4][64];
    __shared__ float cen_s[D], fs_s[2][D];
    __shared__ __attribute__((aligned(16))) uint4 abuf[2][2 * NKS * 64];
    if (threadIdx.x == 0) s_ovf = 0;
    for (int d = threadIdx.x; d < D; d += WG) {
        cen_s[d] = centers[(size_t)tile.state * D + d];
        fs_s[0][d] = fscale[((size_t)tile.state * 2 + 0) * (KS8f * 8) + d];
        fs_s[1][d] = fscale[((size_t)tile.state * 2 + 1) * (KS8f * 8) + d];
    }
    __syncthreads();
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * 64 < vend;
    {   // lane = frame: locate it, stage its centred row
        int v = tile.vstart + wave * 64 + lane;
        const bool ok = v < vend;
        if (!ok) v = tile.vstart;
        // last segment with vstart <= v: almost always seg0 or its successor (two dependent loads instead of the
        // ~log2(#segments) of a full search, at the start of every tile), the search only for what is left
        int lo = tile.seg0, hi = tile.seg_hi - 1;
        if (lo < hi && segs[lo + 1].vstart <= v) {
            ++lo;
            if (lo < hi && segs[lo + 1].vstart <= v) {
                ++lo;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (segs[mid].vstart <= v) lo = mid; else hi = mid - 1;
                }
            }
        }
        const ScoreSeg sg = segs[lo];
        const long long t = v - sg.vstart;
        const float *fp = frames + (sg.frame0 + t) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) xs[wave][lane * XS + d] = fp[d] - cen_s[d];
        oidx_s[wave][lane] = ok ? sg.out0 + t * (long long)sg.out_stride : -1;
    }
    __syncthreads();

    // ---- B operand: block 4 s + g of the long K axis for frame c * 16 + col, f16
    h8v xb[NC][NKS];
    bool ovf = false;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int t = 4 * s + g;
        const int seg = t / SEG8, tt = t - seg * SEG8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 8 * tt + j;
            const bool feat = t < CT && i < 2 * D;
            const int side = (i >= D), dd = feat ? i - side * D : 0;
            const float fsv = fs_s[side][dd];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float xc = xs[wave][(c * 16 + col) * XS + dd];
                float val = (side ? xc : xc * xc) * fsv;
                ovf |= feat && (__builtin_fabsf(val) > FMAXH);
                val = __builtin_fminf(__builtin_fmaxf(val, -FMAXH), FMAXH);
                const _Float16 h1 = (_Float16)val;
                _Float16 piece = (seg == 1) ? (_Float16)(val - (float)h1) : h1;      // x1 | x2 | x1
                if (!feat) piece = (t == CT && j < 3) ? (_Float16)1.f : (_Float16)0.f;   // constants block: 1 1 1 -r1 -r2 -r3 0 0
                xb[c][s][j] = piece;
            }
        }
    }
    if (__any(ovf) && lane == 0) s_ovf = 1;

    float sm[NC], ref[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sm[c] = 0.f;
        ref[c] = 0.f;
    }
    bool ref_ovf = false;

    const uint4 *pstate = pm + (size_t)tile.state * n_mtiles * (2 * NKS * 64);
    auto dma = [&](int buf, int mt) {
        const uint4 *src = pstate + (size_t)mt * (2 * NKS * 64);
        for (int p = wave; p < 2 * NKS; p += WG / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)&abuf[buf][p * 64], 16, 0, 0);
    };
    auto process = [&](int mt) {
        const uint4 *ab = abuf[mt & 1];
        f4v acc[2][NC];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[m][c] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const h8v a = *reinterpret_cast<const h8v *>(&ab[(m * NKS + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[c][s], acc[m][c], 0, 0, 0);
            }
#ifdef PCL_DIAG_NOLSE
#pragma unroll
        for (int c = 0; c < NC; ++c) sm[c] += acc[0][c][0] + acc[1][c][3];
        return;
#endif
        // fast path for all four frame sub-tiles first, ONE wave-uniform test for the slow path
        float snew[NC];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float es[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) es[r] = __builtin_amdgcn_exp2f(acc[r >> 2][c][r & 3]);
#pragma unroll
            for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                for (int r = 0; r < w; ++r) es[r] += es[r + w];
            snew[c] = sm[c] + es[0];
            bad |= !(snew[c] < 3.0e38f);
        }
        if (mt != 0 && !__any(bad)) {
#pragma unroll
            for (int c = 0; c < NC; ++c) sm[c] = snew[c];
            return;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float gm = acc[0][c][0];
#pragma unroll
            for (int r = 1; r < 8; ++r) gm = __builtin_fmaxf(gm, acc[r >> 2][c][r & 3]);
            gm = __builtin_fmaxf(gm, __shfl_xor(gm, 16, 64));
            const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));      // max over the frame's 32 mixtures
            float s = sm[c];
            // everything below -5e4: log zero, or a frame the f16 constants cannot follow -> the direct form decides
            if (mt == 0 && !(gp > -5.0e4f)) ref_ovf = true;
            if ((mt == 0 || gp > 0.f) && gp > -5.0e4f) {
                s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-gp);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r >> 2][c][r & 3] -= gp;
                ref[c] += gp;
                const float nr = __builtin_fminf(__builtin_fmaxf(-ref[c], -FMAXH), FMAXH);
                ref_ovf |= __builtin_fabsf(ref[c]) > 5.0e4f;
                const _Float16 r1 = (_Float16)nr;
                const _Float16 r2 = (_Float16)(nr - (float)r1);
                const _Float16 r3 = (_Float16)(nr - (float)r1 - (float)r2);
                if (g == GCT) {
                    xb[c][SCT][3] = r1;
                    xb[c][SCT][4] = r2;
                    xb[c][SCT][5] = r3;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) s += __builtin_amdgcn_exp2f(acc[r >> 2][c][r & 3]);
            sm[c] = s;
        }
    };
    dma(0, 0);
    for (int mt = 0; mt < n_mtiles; ++mt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile mt have landed
        __syncthreads();                                    // everyone's have; the other buffer is free
        if (mt + 1 < n_mtiles) dma((mt + 1) & 1, mt + 1);
        if (wave_active) process(mt);
    }
    if (__any(ref_ovf) && lane == 0) s_ovf = 1;
    constexpr double LN2 = 0.693147180559945309417232121458;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float t = sm[c] + __shfl_xor(sm[c], 16, 64);
        const double S = (double)t + (double)__shfl_xor(t, 32, 64);
        const long long o = oidx_s[wave][c * 16 + col];
        if (o >= 0 && g == 0) out[o] = (S > 0) ? LN2 * ((double)ref[c] + ::log2(S)) : -INFINITY;
    }
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = s_ovf;
}

template <int D>
void launch16x_t(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    hipLaunchKernelGGL((gmm_score_split16x_kernel<D>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32,
                       reinterpret_cast<const uint4 *>(ctx->pm16x), ctx->fscale, ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs, b->Bt,
                       b->d_tile_flags);
}

template <int D>
void launch16_t(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (ctx->score_variant == 7)
        hipLaunchKernelGGL((gmm_score_split16_kernel<D, PCL_SPLIT16_NT, true>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32,
                           reinterpret_cast<const uint4 *>(ctx->pm16f), ctx->fscale, ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs,
                           b->Bt, b->d_tile_flags, ctx->kzero);
    else
        hipLaunchKernelGGL((gmm_score_split16_kernel<D, PCL_SPLIT16_NT, false>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32,
                           reinterpret_cast<const uint4 *>(ctx->pm16h), ctx->fscale, ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs,
                           b->Bt, b->d_tile_flags, (const double *)nullptr);
}

template <int D>
void launch_t(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    hipLaunchKernelGGL((gmm_score_split_kernel<D, PCL_SPLIT_NT>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32,
                       reinterpret_cast<const uint4 *>(ctx->pm16), ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs, b->Bt);
}

}  // namespace

int pcl_score_split_tile_frames() { return WG / 64 * PCL_SPLIT_NT * 32; }

int pcl_score_split16_tile_frames() { return WG / 64 * PCL_SPLIT16_NT * 32; }

int pcl_launch_score_split16(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 39: launch16_t<39>(ctx, b, tiles, n_tiles); break;
        case 26: launch16_t<26>(ctx, b, tiles, n_tiles); break;
        case 13: launch16_t<13>(ctx, b, tiles, n_tiles); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no split-f16 scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_score_split16x(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 39: launch16x_t<39>(ctx, b, tiles, n_tiles); break;
        case 26: launch16x_t<26>(ctx, b, tiles, n_tiles); break;
        case 13: launch16x_t<13>(ctx, b, tiles, n_tiles); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no 16x16x32 split-f16 scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_score_split(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 39: launch_t<39>(ctx, b, tiles, n_tiles); break;
        case 26: launch_t<26>(ctx, b, tiles, n_tiles); break;
        case 13: launch_t<13>(ctx, b, tiles, n_tiles); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no split-bf16 scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

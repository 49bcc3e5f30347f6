// gmm_score_split.hip -- GMM scoring as an f32-class contraction on the f16 matrix pipe (gfx950): the default f32 path.
//
// Reference rows (as gmm_score.hip): A1/A4/A6 -- util.gaussian_function (util.py:20-31), Clustering.GMM.point
// (Clustering.py:740-767), LHMM.cal_observation_pro (LHMM.py:163-187).  Expanded around a per-state centre c_j the
// exponent of mixture m for frame f is a contraction over K = 2D + 2:
//     v[f,m] = k'_m + sum_d ( a_md x'_fd^2 + b_md x'_fd ) - ref_f,      x' = x - c_j
// (centring keeps the cancellation of the expanded form at the direct form's level; states whose expansion is still ill
// conditioned go to the direct-form kernel, pcl_model_conditioning).
//
// Which pipe.  v_mfma_f32_32x32x2_f32 runs at the VALU's rate AND blocks the VALU while it runs (gmm_score_mfma.hip,
// PCL_SCORE_VARIANT=3: 0.65 of its 157 TFLOP/s peak, the strict-f32 number kept in the bench line).  The f16 / bf16 pipe is
// 16x faster and separate.  f16 has 11 significand bits, so x = h1 + h2 carries 22 and
//     a x = a1 x1 + a1 x2 + a2 x1 + O(2^-22 |a x|)
// three v_mfma_f32_32x32x16_f16 per K-step with f32 accumulation.  f16's narrow exponent range is handled by exact
// power-of-two scaling per (state, feature): A'' = coef 2^-e with max_m |A''| in [1, 2), B'' = feature 2^e (model_derive.hip
// writes 2^e next to the layout); subnormal second pieces are honoured by the f16 MFMA (tools/ubench_f16denorm.hip), so small
// features keep an ABSOLUTE error of 2^-25.  The spare K slot d = D of each side carries the constants:
//     a1: [k1 | 0]   a2: [k2 | 1]   x1: [1 | -ref']   x2: [0 | 0]   ->  a2 x1 + a1 x2 + a1 x1 = k1 + k2 - ref'
// with k1 + k2 = k'_m - K0_j (K0_j = max_m k'_m, added back in f64 at the end, so the 22 bits go to a small number) and
// ref' the f16-rounded log-sum-exp reference (any nearby value serves).  Log zero is -6e4.  A frame whose scaled feature
// exceeds 6e4 (|x - c| beyond ~300 sigma of the tightest mixture) or whose reference leaves the f16 range raises its tile's
// flag: the direct-form kernel then rescored flagged tiles in the same call (pcl_launch_score_fixup), so no input sees an
// overflowed result.  Measured max |d ln b| against float64 at |ln b| ~ 85: 1.5e-5 (the exact f32 chain: 1.0e-5).
//
// Mapping (v_mfma_f32_32x32x16_f16: D[32 mixtures x 32 frames] += A[32 x 16] B[16 x 32]).  K-step s, lane l (r = l&31,
// h = l>>5), element j <-> feature d = 8s + j of the side h selects:
//   B (frames):     h = 0: x'_d^2 2^e,  h = 1: x'_d 2^e     -- two f16 pieces, resident in VGPRs (a wave owns NT = 2 tiles)
//   A (parameters): h = 0: a_md 2^-e,   h = 1: b_md 2^-e    -- layout [m-tile][piece 2][KS8][64 lanes][8 f16] = 10 KB per 32
//                   mixtures, staged in LDS by LDS-DMA one tile ahead (double buffered) and shared by the 4 waves.
// Pass order a2x1, a1x2, a1x1 (small terms first).  In the accumulator a lane owns 16 mixture values of ONE frame, so the
// log-sum-exp is per lane: s += exp2(v - ref) (one v_exp_f32 + one add per Gaussian); ref is a true earlier maximum,
// raised on a wave-uniform slow path (first tile, or when a sum overflows f32).  ln2 (ref + K0 + log2 s) is finished in f64.
//
// History (profiles/r01_split_variants.txt, r01_score_variants.txt; the superseded kernels were removed in round 2): VALU
// kernel 64 TFLOP/s -> f32-input MFMA 102 -> three-piece bf16 split (six products) 186 -> two-piece f16 split with a separate
// constant MFMA 274 -> constants folded into the spare slot 300; the same scheme on 16x16x32 MFMAs measured 16.7 ms against
// 15.1 (more B-operand registers, 45 % more LDS fragment traffic).
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

#ifndef PCL_SPLIT_WG
#define PCL_SPLIT_WG 256    // threads per workgroup = 64 x (waves sharing one LDS copy of the A tile)
#endif
constexpr int WG = PCL_SPLIT_WG;
typedef float f16v __attribute__((ext_vector_type(16)));

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

template <int D, int NT>
__global__ __launch_bounds__(WG, PCL_SPLIT16_MINW * 256 / WG > 0 ? PCL_SPLIT16_MINW * 256 / WG : 1) void gmm_score_split16_kernel(
    const float *__restrict__ frames, const uint4 *__restrict__ pm, const float *__restrict__ fscale, const float *__restrict__ centers,
    int nmt_max, const ScoreTile *__restrict__ tiles, const ScoreSeg *__restrict__ segs, double *__restrict__ out,
    int *__restrict__ flags, const double *__restrict__ kzero, const int *__restrict__ n_on_pipe, const int *__restrict__ npt) {
    static_assert(D % 8 != 0, "the folded constants need a spare slot");
    constexpr int KS8 = (D + 7) / 8;       // K-steps of 16 over the 2D features (8 per half-wave)
    constexpr int CH = 2 * KS8;            // 1-KiB chunks per m-tile: two f16 pieces
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
    // tiles of the state's layout in use: all of them, or -- a split state, whose on-pipe mixtures are compacted to the front (round 6) --
    // ceil(on-pipe / 32)
    const int n_mtiles = npt[tile.state];
    __shared__ int s_ovf;
    if (threadIdx.x == 0) s_ovf = 0;
    __syncthreads();
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * NT * 32 < vend;
    if (n_mtiles == 0) {                    // not one mixture of the state on the pipe: ln 0 for every frame (the coarse pass adds the rest), no flag
        if (wave_active) {
            for (int c = 0; c < NT; ++c) {
                const int v = tile.vstart + (wave * NT + c) * 32 + col;
                if (v < vend && half == 0) {
                    int lo = tile.seg0;
                    const int hi = tile.seg_hi - 1;
                    while (lo < hi && segs[lo + 1].vstart <= v) ++lo;
                    const ScoreSeg sg = segs[lo];
                    out[sg.out0 + (long long)(v - sg.vstart) * sg.out_stride] = -INFINITY;
                }
            }
        }
        if (threadIdx.x == 0) flags[blockIdx.x] = 0;
        return;
    }

    // ---- B operand: scaled features of this lane's frames in two f16 pieces (spare slot: x1 = [1 | -ref'])
    h8v xb[NT][2][KS8];
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
                if (d == D) val = half ? 0.f : 1.f;              // x1: [1 | -ref' (0 so far)]
                const _Float16 h1 = (_Float16)val;
                xb[c][0][s][j] = h1;
                xb[c][1][s][j] = (_Float16)(val - (float)h1);
            }
        oidx[c] = sg.out0 + t * (long long)sg.out_stride;
    }
    if (__any(ovf) && lane == 0) s_ovf = 1;

    float sm[NT], ref[NT];
    bool inited[NT];                          // the lane's frame has a log-sum-exp reference (a real value was seen)
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        sm[c] = 0.f;
        ref[c] = 0.f;
        inited[c] = false;
    }
    bool ref_ovf = false;

    // MTS m-tiles per LDS stage: one workgroup barrier per MTS x 32 mixtures
    constexpr int MTS = PCL_SPLIT16_MTS;
    __shared__ __attribute__((aligned(16))) uint4 abuf[2][MTS * CH * 64];
    const uint4 *pstate = pm + (size_t)tile.state * nmt_max * (CH * 64);
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
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
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
#pragma unroll
        for (int c = 0; c < NT; ++c) {
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
            // (round 6: the reference is taken at the first tile that HAS a real value, not at tile 0 -- with most of a state's mixtures off
            //  the pipe its first 32 are often all log zero, which used to flag every tile of the state for the direct-form fix-up: 34 ms
            //  per EM iteration at 91 % off-pipe mixtures; a frame that never sees a value above the f16 constants' reach is flagged at the end)
            if (__any(!inited[c]) || __any(!(snew < 3.0e38f))) {
                float gm = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
                const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));
                float s = sm[c];
                const bool first = !inited[c];
                if ((first || gp > 0.f) && gp > -5.0e4f) {
                    // the reference the pipe subtracts is the f16-rounded one: shift by what it actually moves
                    const _Float16 r1 = (_Float16)__builtin_fminf(__builtin_fmaxf(-(ref[c] + gp), -FMAXH), FMAXH);
                    const float nref = -(float)r1, dl = nref - ref[c];
                    ref_ovf |= __builtin_fabsf(nref) > 5.0e4f;
                    // (in two half-steps: this path is entered when a sum overflows, i.e. with dl ~ 128, and exp2(-128) is a denormal that
                    //  v_exp_f32 flushes to 0 -- which dropped everything summed so far, up to a third of the frame's mass, rounds 1-5)
#ifdef PCL_LSE_FLUSH_REPRO                                    // mutation build (tests are expected to FAIL on it): rounds 1-5
                    s = first ? 0.f : s * __builtin_amdgcn_exp2f(-dl);
#else
                    const float hs = __builtin_amdgcn_exp2f(-0.5f * dl);
                    s = first ? 0.f : (s * hs) * hs;
#endif
                    inited[c] = true;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= dl;
                    ref[c] = nref;
                    if (half) xb[c][0][SC][JC] = r1;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
                sm[c] = s;
            } else {
                sm[c] = snew;
            }
        }
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
    const double k0 = kzero[tile.state];
    // a frame no tile gave a reference: every on-pipe mixture log zero for it -- right (-inf) when the state HAS no on-pipe mixture, otherwise
    // out of the f16 constants' reach: the direct-form fix-up rescored the tile
    if (wave_active && n_on_pipe[tile.state] > 0) {
#pragma unroll
        for (int c = 0; c < NT; ++c) ref_ovf |= !inited[c];
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const double S = (double)sm[c] + (double)__shfl_xor(sm[c], 32, 64);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2 * ((double)ref[c] + k0 + ::log2(S)) : -INFINITY;
    }
    if (__any(ref_ovf) && lane == 0) s_ovf = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s_ovf) flags[blockIdx.x] = 1;         // (the flag of the features was stored after the first barrier)
}

template <int D>
void launch16_t(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    // pcl_score_occupancy(ctx, 2): dynamic LDS nobody reads, so that two instead of three scoring workgroups fit a CU and another kernel's
    // waves -- the token passing of a streamed decode -- find registers beside them (config 5 streamed: 0.79 -> 0.87 M frames/s; the
    // scoring kernel alone loses ~9 %, so the streamed decoder asks for it and nobody else does).  Per workgroup: 56 KB > 160 / 3.
    constexpr size_t static_lds = 2 * (size_t)(2 * ((D + 7) / 8)) * 64 * 16;
    const size_t pad = (ctx->score_wgs_per_cu == 2 && static_lds < (56u << 10)) ? (56u << 10) - static_lds : 0;
    hipLaunchKernelGGL((gmm_score_split16_kernel<D, PCL_SPLIT16_NT>), dim3(n_tiles), dim3(WG), pad, ctx->stream, ctx->frames32,
                       reinterpret_cast<const uint4 *>(ctx->pm16f), ctx->fscale, ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs,
                       b->Bt, b->d_tile_flags, ctx->kzero, ctx->d_non, ctx->d_npt);
}

}  // namespace

int pcl_score_split16_tile_frames() { return WG / 64 * PCL_SPLIT16_NT * 32; }

int pcl_launch_score_split16(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 47: launch16_t<47>(ctx, b, tiles, n_tiles); break;
        case 39: launch16_t<39>(ctx, b, tiles, n_tiles); break;
        case 26: launch16_t<26>(ctx, b, tiles, n_tiles); break;
        case 13: launch16_t<13>(ctx, b, tiles, n_tiles); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no split-f16 scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

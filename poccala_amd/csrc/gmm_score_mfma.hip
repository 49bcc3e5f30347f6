// gmm_score_mfma.hip -- GMM scoring with the quadratic form on the f32-input matrix pipe (gfx950).
//
// Same reference rows as gmm_score.hip (A1/A4/A6: util.py:20-31, Clustering.py:740-767, LHMM.py:163-187).
//
// Why MFMA at all.  The diagonal Gaussian exponent, expanded around a per-state centre c_j,
//     v[f,m] = k'_m + sum_d ( a_md x'_fd^2 + b_md x'_fd ),      x' = x - c_j,
//     a = -log2e / (2 var),  b = log2e (mu - c_j) / var,  k' = k2 - log2e sum_d (mu - c_j)^2 / (2 var)
// IS a dense contraction [frames x (2D+1)] . [(2D+1) x mixtures].  v_mfma_f32_32x32x2_f32 takes f32
// inputs and is bit-for-bit a k-ordered chain of f32 FMAs (MI355X_MICROARCH.md, "exact f32"), so this is
// the same arithmetic class as the VALU kernel -- no bf16/tf32 rounding anywhere -- but it issues on the
// matrix pipe, which sustains 140-155 TFLOP/s on this chip (tools/ubench_mfma.hip) where v_fma_f32
// sustains ~107 for the dependent y = x s + c; q += y y mix, and it leaves the VALU free for the
// log-sum-exp.  Centring on c_j keeps the expanded form's cancellation at the level of the direct form
// (the terms are O((x-c)^2/var), not O(x^2/var)).
//
// Mapping.  D[32 mixtures x 32 frames] += A[32 x 2] B[2 x 32] per instruction, K = 2D+2 (D = 39: 40 k-steps).
//   A (parameters): lane l holds P[m0 + (l&31)][2s + (l>>5)]  ->  lanes 0-31 carry a_ms (k' at s = D),
//                   lanes 32-63 carry b_ms (0 at s = D).  Streamed from L2 with dwordx4 loads, 10 per m-tile.
//   B (frames):     lane l holds X[f0 + (l&31)][2s + (l>>5)]  ->  lanes 0-31 carry x'^2 (1 at s = D),
//                   lanes 32-63 carry x' (0 at s = D).  Resident in VGPRs for the whole kernel.
//   D: lane l, reg r = mixture row (r&3) + 8 (r>>2) + 4 (l>>5) of frame column l&31: a lane owns 16 of the
//      32 mixture values of ONE frame, so the log-sum-exp over mixtures is per lane (reference-shifted,
//      see the loop) and the two half-waves are merged once at the end.
// A wave owns NT = 2 column tiles (64 frames); 4 waves per workgroup; 2 waves per SIMD.
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

#ifndef PCL_MFMA_WG
#define PCL_MFMA_WG 256     // threads per workgroup = 64 x (waves sharing one LDS copy of the A tile)
#endif
constexpr int WG = PCL_MFMA_WG;
#ifndef PCL_MFMA_NT
#define PCL_MFMA_NT 2      // frame column tiles (32 frames) per wave
#endif
#ifndef PCL_MFMA_MINW
#define PCL_MFMA_MINW 2    // __launch_bounds__ waves per SIMD
#endif
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int D, int NT>
__global__ __launch_bounds__(WG, PCL_MFMA_MINW) void gmm_score_mfma_kernel(const float *__restrict__ frames,
                                                               const float *__restrict__ pm,
                                                               const float *__restrict__ centers, int n_mtiles,
                                                               const ScoreTile *__restrict__ tiles,
                                                               const ScoreSeg *__restrict__ segs,
                                                               double *__restrict__ out) {
    constexpr int KS = D + 1;              // k-steps of 2: D feature pairs + the constant pair
    constexpr int KS4 = (KS + 3) / 4;      // dwordx4 loads per m-tile
    const ScoreTile tile = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    if (tile.seg_lo >= tile.seg_hi) return;   // padding tile of the XCD-aware order
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * NT * 32 < vend;   // a wave past the end still helps staging

    // ---- B operand: this lane's frames, centred, squared on the low half-wave
    float xb[NT][KS4 * 4];
    long long oidx[NT];
    bool valid[NT];
    const float *cen = centers + (size_t)tile.state * D;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        int v = tile.vstart + (wave * NT + c) * 32 + col;
        valid[c] = v < vend;
        if (!valid[c]) v = tile.vstart;
        int lo = tile.seg_lo, hi = tile.seg_hi - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (segs[mid].vstart <= v) lo = mid; else hi = mid - 1;
        }
        const ScoreSeg sg = segs[lo];
        const long long t = v - sg.vstart;
        const float *fp = frames + (sg.frame0 + t) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float xc = fp[d] - cen[d];
            xb[c][d] = half ? xc : xc * xc;
        }
        xb[c][D] = half ? 0.f : 1.f;
#pragma unroll
        for (int d = D + 1; d < KS4 * 4; ++d) xb[c][d] = 0.f;
        oidx[c] = sg.out0 + t * (long long)sg.out_stride;
    }

    float sm[NT], ref[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        sm[c] = 0.f;
        ref[c] = 0.f;
    }

    // parameters of this state: [m-tile][KS4][64 lanes][4] floats, staged per m-tile in LDS by LDS-DMA
    // (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR destination) and shared by the 4 waves.
    __shared__ __attribute__((aligned(16))) float abuf[2][KS4 * 64 * 4];
    const float *pstate = pm + (size_t)tile.state * n_mtiles * (KS4 * 64 * 4);
    auto dma = [&](int buf, int mt) {
        const float *src = pstate + (size_t)mt * (KS4 * 64 * 4);
        for (int p = wave; p < KS4; p += WG / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)&abuf[buf][p * 256], 16, 0, 0);
    };

    // Reference-shifted log-sum-exp.  The K dimension has one spare slot (2D+1 features in 2D+2): the
    // parameter side holds 1 there and the frame side holds -ref[frame], so the matrix pipe delivers
    // v - ref and the VALU only does  s += sum_r exp2(v_r - ref): 16 v_exp_f32 and a packed add tree per
    // 16 Gaussians -- no subtract, no max, no per-tile rescale.  (Every VALU instruction issued on a SIMD
    // costs matrix-pipe time: SQ_VALU_MFMA_BUSY_CYCLES showed the pipe 82 % busy at 1.3 VALU per MFMA.)
    // ref is a true maximum seen earlier for that frame, so the largest term is >= 1 and nothing
    // underflows.  It is raised on a wave-uniform slow path, taken on the first m-tile and whenever a sum
    // overflows f32 (a value more than ~2^127 above ref): that path recomputes the tile's sum from the
    // still-live accumulators with the usual max-rescale.  Both half-waves of a frame column share one
    // ref, so their partial sums simply add at the end.
    auto process = [&](const f4v (&av)[KS4], int mt) {
        f16v acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s >> 2][s & 3], xb[c][s], acc[c], 0, 0, 0);
        }
#ifdef PCL_DIAG_NOLSE
#pragma unroll
        for (int c = 0; c < NT; ++c) sm[c] += acc[c][0] + acc[c][15];   // diagnostic build: no log-sum-exp work (wrong results)
        return;
#endif
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
                if ((mt == 0 || gp > 0.f) && gp > -INFINITY) {
                    // first tile: gp << 0, 0 * exp2(-gp) would be 0 * inf.  Later: this path is entered when a sum overflows, i.e. with gp ~ 128,
                    // and exp2(-128) is a denormal that v_exp_f32 flushes to 0 -- which dropped everything summed so far, up to a third of the
                    // frame's mass (rounds 1-5; found by tests/test_gpu_fuzz_estep.py).  Two half-steps stay normal up to gp = 252.
#ifdef PCL_LSE_FLUSH_REPRO                                    // mutation build (tests are expected to FAIL on it): rounds 1-5
                    s = (mt == 0) ? 0.f : s * __builtin_amdgcn_exp2f(-gp);
#else
                    const float hs = __builtin_amdgcn_exp2f(-0.5f * gp);
                    s = (mt == 0) ? 0.f : (s * hs) * hs;
#endif
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= gp;
                    ref[c] += gp;
                    if (half) xb[c][D] = -ref[c];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
                sm[c] = s;
            } else {
                sm[c] = snew;
            }
        }
    };
    // One barrier per m-tile.  The DMA of tile i+1 is issued right after the barrier that publishes tile i
    // and has a whole tile of MFMAs (5120 cycles) to land; hipcc cannot sink it the way it sinks ordinary
    // global loads (it sank a register prefetch to just in front of the MFMAs that use it, exposing the L2
    // latency on every tile).
    dma(0, 0);
    for (int mt = 0; mt < n_mtiles; ++mt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile mt have landed
        __syncthreads();                                    // everyone's pieces have; buffer (mt+1)&1 is free
        if (mt + 1 < n_mtiles) dma((mt + 1) & 1, mt + 1);
        if (wave_active) {
            f4v a[KS4];
#pragma unroll
            for (int q = 0; q < KS4; ++q) a[q] = *reinterpret_cast<const f4v *>(&abuf[mt & 1][(q * 64 + lane) * 4]);
            process(a, mt);
        }
    }
    constexpr double LN2 = 0.693147180559945309417232121458;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const double S = (double)sm[c] + (double)__shfl_xor(sm[c], 32, 64);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2 * ((double)ref[c] + ::log2(S)) : -INFINITY;
    }
}

template <int D>
void launch_t(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    hipLaunchKernelGGL((gmm_score_mfma_kernel<D, PCL_MFMA_NT>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32, ctx->pm32,
                       ctx->centers32, ctx->Mpad32 / 32, tiles, b->d_segs, b->Bt);
}

}  // namespace

int pcl_score_mfma_tile_frames() { return WG / 64 * PCL_MFMA_NT * 32; }

bool pcl_score_mfma_supported(int D) { return D == 39 || D == 13 || D == 26 || D == 47; }

int pcl_launch_score_mfma(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 47: launch_t<47>(ctx, b, tiles, n_tiles); break;
        case 39: launch_t<39>(ctx, b, tiles, n_tiles); break;
        case 26: launch_t<26>(ctx, b, tiles, n_tiles); break;
        case 13: launch_t<13>(ctx, b, tiles, n_tiles); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no MFMA scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

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
//      32 mixture values of ONE frame, so the log-sum-exp over mixtures is 16 values per lane per m-tile
//      (one rescale per 16 -> 17/16 v_exp_f32 per Gaussian) and the two half-waves are merged once at the end.
// A wave owns NT = 2 column tiles (64 frames); 4 waves per workgroup; 2 waves per SIMD.
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

constexpr int WG = 256;
#ifndef PCL_MFMA_NT
#define PCL_MFMA_NT 2      // frame column tiles (32 frames) per wave
#endif
#ifndef PCL_MFMA_MINW
#define PCL_MFMA_MINW 2    // __launch_bounds__ waves per SIMD
#endif
#ifndef PCL_MFMA_LSE
#define PCL_MFMA_LSE 1     // 1: reference-shifted log-sum-exp (the shift rides in the spare K slot of the MFMA)
#endif
#ifndef PCL_MFMA_PIPE
#define PCL_MFMA_PIPE 0    // 1: log-sum-exp of m-tile i-1 is issued behind the MFMAs of m-tile i
#endif
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

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
    if (tile.vstart + wave * NT * 32 >= vend) return;   // whole wave past the end of the state's frames

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

    float mx[NT], sm[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        mx[c] = -1.0e30f;
        sm[c] = 0.f;
    }

    // parameters of this state: [m-tile][KS4][64 lanes][4] floats
    const f4v *pa = reinterpret_cast<const f4v *>(pm) + (size_t)tile.state * n_mtiles * (KS4 * 64) + lane;
    f4v a[KS4];
#pragma unroll
    for (int q = 0; q < KS4; ++q) a[q] = pa[q * 64];

#if PCL_MFMA_LSE
    // Reference-shifted log-sum-exp.  The K dimension has one spare slot (2D+1 features in 2D+2): the
    // parameter side holds 1 there and the frame side holds -ref[frame], so the matrix pipe delivers
    // v - ref and the VALU only has to do  s += exp2(v - ref)  (one v_exp_f32 + one add per Gaussian, no
    // subtract, no per-tile rescale).  ref is a true maximum seen earlier for that frame, so the largest
    // term is >= 1 and nothing underflows; it is raised only when some value exceeds it by > 2^64
    // (wave-uniform slow path, always taken on the first m-tile).  Both half-waves of a frame column share
    // one ref, so their partial sums simply add at the end.
    float ref[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) ref[c] = 0.f;
    for (int mt = 0; mt < n_mtiles; ++mt) {
        const f4v *pn = pa + (size_t)(mt + 1 < n_mtiles ? mt + 1 : mt) * (KS4 * 64);
        f4v an[KS4];
#pragma unroll
        for (int q = 0; q < KS4; ++q) an[q] = pn[q * 64];
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
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s >> 2][s & 3], xb[c][s], acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float gm = acc[c][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
            if (mt == 0 || __any(gm > 64.f)) {
                const float gp = __builtin_fmaxf(gm, __shfl_xor(gm, 32, 64));   // max over the frame's 32 mixtures
                if ((mt == 0 || gp > 0.f) && gp > -INFINITY) {
                    sm[c] = (mt == 0) ? 0.f : sm[c] * __builtin_amdgcn_exp2f(-gp);   // first tile: gp << 0, 0 * exp2(-gp) would be 0 * inf
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] -= gp;
                    ref[c] += gp;
                    if (half) xb[c][D] = -ref[c];
                }
            }
            float s = sm[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(acc[c][r]);
            sm[c] = s;
        }
#pragma unroll
        for (int q = 0; q < KS4; ++q) a[q] = an[q];
    }
    constexpr double LN2R = 0.693147180559945309417232121458;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const double S = (double)sm[c] + (double)__shfl_xor(sm[c], 32, 64);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2R * ((double)ref[c] + ::log2(S)) : -INFINITY;
    }
    return;
#else
    auto lse_update = [&](const f16v &t, int c) {
        float gm = t[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, t[r]);
        const float nm = __builtin_fmaxf(mx[c], gm);
        float s = sm[c] * __builtin_amdgcn_exp2f(mx[c] - nm);
#pragma unroll
        for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(t[r] - nm);
        sm[c] = s;
        mx[c] = nm;
    };
#if PCL_MFMA_PIPE
    f16v prev[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[c][r] = -INFINITY;   // contributes exp2(-inf) = 0
    }
#endif
    for (int mt = 0; mt < n_mtiles; ++mt) {
        // prefetch the next m-tile's A operand (the last iteration re-reads the current tile)
        const f4v *pn = pa + (size_t)(mt + 1 < n_mtiles ? mt + 1 : mt) * (KS4 * 64);
        f4v an[KS4];
#pragma unroll
        for (int q = 0; q < KS4; ++q) an[q] = pn[q * 64];

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
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s >> 2][s & 3], xb[c][s], acc[c], 0, 0, 0);
        }
        // online log-sum-exp over this lane's 16 mixtures of each frame column (log2 domain)
#if PCL_MFMA_PIPE
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            lse_update(prev[c], c);
            prev[c] = acc[c];
        }
#else
#pragma unroll
        for (int c = 0; c < NT; ++c) lse_update(acc[c], c);
#endif
#pragma unroll
        for (int q = 0; q < KS4; ++q) a[q] = an[q];
    }
#if PCL_MFMA_PIPE
#pragma unroll
    for (int c = 0; c < NT; ++c) lse_update(prev[c], c);
#endif

#endif
    constexpr double LN2 = 0.693147180559945309417232121458;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        // merge the two half-waves (mixture rows 4h..4h+3 mod 8 of every m-tile)
        const float m2 = __shfl_xor(mx[c], 32, 64);
        const float s2 = __shfl_xor(sm[c], 32, 64);
        const float M = __builtin_fmaxf(mx[c], m2);
        const double S = (double)sm[c] * (double)__builtin_amdgcn_exp2f(mx[c] - M) +
                         (double)s2 * (double)__builtin_amdgcn_exp2f(m2 - M);
        if (valid[c] && half == 0) out[oidx[c]] = (S > 0) ? LN2 * ((double)M + ::log2(S)) : -INFINITY;
    }
}

template <int D>
void launch_t(pcl_ctx *ctx, pcl_batch *b) {
    hipLaunchKernelGGL((gmm_score_mfma_kernel<D, PCL_MFMA_NT>), dim3(b->n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32, ctx->pm32,
                       ctx->centers32, ctx->Mpad32 / 32, b->d_tiles, b->d_segs, b->Bt);
}

}  // namespace

int pcl_score_mfma_tile_frames() { return WG / 64 * PCL_MFMA_NT * 32; }

bool pcl_score_mfma_supported(int D) { return D == 39 || D == 13 || D == 26; }

int pcl_launch_score_mfma(pcl_ctx *ctx, pcl_batch *b) {
    if (b->n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score");
    switch (ctx->D) {
        case 39: launch_t<39>(ctx, b); break;
        case 26: launch_t<26>(ctx, b); break;
        case 13: launch_t<13>(ctx, b); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no MFMA scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

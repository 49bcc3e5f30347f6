// gmm_score.hip -- batched diagonal-GMM log-likelihood  ln b_j(o_t)  for gfx950 (MI355X).
//
// Replaces the reference's HOT LOOP 1:
//   LHMM.cal_observation_pro          StatisticalModel/LHMM.py:163-187
//   -> Clustering.GMM.point(log=True)  StatisticalModel/Clustering.py:740-767
//   -> util.gaussian_function(log=True) StatisticalModel/util.py:20-31   (quirk Q1 constant)
//
// This is the VALU formulation: the float64 parity mode always uses it, the float32 mode uses it for
// feature dimensions the MFMA kernel (gmm_score_mfma.hip) has no instantiation for, or on request
// (PCL_SCORE_VARIANT=1).  A third variant that fed the parameters through SGPR pairs into
// v_pk_fma_f32 op_sel measured slower with compiler-scheduled scalar loads (profiles/r01_score_variants.txt)
// and was removed.
//
// Mapping (VALU bound, no MFMA -- see DESIGN.md):
//   * state-major batching: a workgroup scores TILE frames of ONE GMM state, gathered from all
//     utterances of the batch that contain the state, so the state's M x (2D+1) parameter block is
//     streamed through LDS once per TILE frames.
//   * lanes = frames.  Each lane keeps R frames x D features in VGPRs; the per-mixture parameters
//     are wave-uniform and read from LDS as broadcast ds_read_b128 (no bank conflicts).
//   * per (frame, mixture, dim) exactly two FMAs:  y = x*s + c ;  q += y*y   with
//        s = sqrt(log2e / (2 var)),  c = -mu*s   =>  q = log2e * (x-mu)^2 / (2 var)
//     and the mixture value (log2 domain)  v = const2 - q,
//        const2 = log2e * (ln w - D/2 ln 2pi - 1/2 sum(var))          [util.py:29, quirk Q1]
//   * log-sum-exp over mixtures is an online (running max, running sum) per lane in the log2 domain,
//     rescaled once per group of G mixtures: (1 + 1/G) v_exp_f32 per Gaussian, no cross-lane traffic.
//   * result  ln b = ln2 * (max + log2(sum))  is finished in float64 and written to the time-major
//     emission matrix consumed by the DP kernels.
#include <stdlib.h>

#include "pcl_internal.h"

namespace {

constexpr int WG = 256;  // 4 waves
#ifndef PCL_GROUP
#define PCL_GROUP 4
#endif
#ifndef PCL_R32
#define PCL_R32 3   // measured on MI355X: R=3 (3 waves/SIMD) 64.3 TF vs R=4 (2 waves/SIMD) 52.8, R=5 54.7
#endif
#ifndef PCL_CH32
#define PCL_CH32 64   // mixtures per LDS chunk (f32)
#endif
#ifndef PCL_PDE_D1
#define PCL_PDE_D1 6    // partial-distance elimination: features before the first / second test (0 = off; D2 <= D1: one test); (6,16) / (4,10) / (2,8) / (3,-) measured 35.2 / 36.8 / 37.3 / 42.4 ms against 105 on the collapsed C4 model
#endif
#ifndef PCL_PDE_D2
#define PCL_PDE_D2 16
#endif
#ifndef PCL_PDE_SUBSET
#define PCL_PDE_SUBSET 0   // the test in the subset launch of split states (A/B: tools/build_variant.sh)
#endif
constexpr int GROUP = PCL_GROUP;  // mixtures per LSE rescale (Mpad is a multiple of 4 >= this)

template <typename real>
struct Fast;
template <>
struct Fast<float> {
    static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float neg_big() { return -1.0e30f; }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
};
template <>
struct Fast<double> {
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double neg_big() { return -1.0e300; }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
};

// The float64 master copy of a state's parameters, for the FIX-UP launches (MASTER = true): the rows [s_d c_d ... k2] the kernel
// reads are made on the fly from (mean, var, weight) with derive_kernel's arithmetic (model_derive.hip) while they are staged in
// LDS, so the 3 GB of direct-form layouts (params32, mean32) need not exist unless a whole state is scored by this kernel: they
// fed the fix-up paths only, and a flagged tile pays D square roots per mixture against 2 D x 256 FMAs.
// Split states (pcl_internal.h): `bad` marks the mixtures that are off the matrix pipe.  A fix-up launch (MASTER) rescoring a flagged tile
// leaves them out like the pipe did; the SUBSET launch scores exactly them -- the state's compacted list bad_idx[0 .. nbad) -- for every
// frame of the split states and log-adds the result to what the pipe wrote.
struct MasterModel {
    const double *mean64, *var64, *w64;
    int M, Dhost, flags;
    const unsigned char *bad;
    const int *bad_idx, *nbad;
};

// (MASTER / SUBSET: three workgroups per CU -- the float64 row arithmetic of the staging took 205 VGPRs under a bound of two, and the subset
//  launch's short workgroups need the third wave per SIMD to cover their prologues: 9.5 -> 7.2 ms per batch at 7 % off-pipe mixtures, with
//  50 registers spilled in the staging; R = 2 x 4 workgroups 7.1, R = 3 x 4 23.7 (spills in the loop), R = 4 x 2 8.2)
#ifndef PCL_RSUB
#define PCL_RSUB 3          // frames per lane of the subset launch
#endif
#ifndef PCL_SUBSET_MINB
#define PCL_SUBSET_MINB 3   // its workgroups per CU
#endif
template <int D, int R, int CH, typename real, bool MASTER = false, bool SUBSET = false>
__global__ __launch_bounds__(WG, SUBSET ? PCL_SUBSET_MINB : (MASTER && sizeof(real) == 4) ? 3 : 2) void gmm_score_kernel(const real *__restrict__ frames,
                                                          const real *__restrict__ params, int Mpad,
                                                          const ScoreTile *__restrict__ tiles,
                                                          const ScoreSeg *__restrict__ segs,
                                                          double *__restrict__ out, const int *__restrict__ flags, int n_tiles, MasterModel mm) {
    constexpr int ROW = (2 * D + 1 + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) real lds[CH * ROW];

  auto score_tile = [&](const int ti) {
    const ScoreTile tile = tiles[ti];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (tile.seg_lo >= tile.seg_hi) return;   // padding tile of the XCD-aware order
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;

    real x[R][D];
    long long oidx[R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int v = tile.vstart + (wave * R + r) * 64 + lane;
        valid[r] = v < vend;
        if (!valid[r]) v = tile.vstart;  // any in-range frame: keeps loads safe, result discarded
        // last segment with vstart <= v: tile.seg0 holds the tile's first frame, so it is seg0 or one of its next few successors (a tile
        // of 768 frames spans ~3 utterances' runs): short dependent chains instead of the ~log2(#segments) of a search from seg_lo, which
        // is most of the prologue of the short subset launches; the search only for what is left
        int lo = tile.seg0, hi = tile.seg_hi - 1;
#pragma unroll
        for (int step = 0; step < 3; ++step)
            if (lo < hi && segs[lo + 1].vstart <= v) ++lo; else hi = lo;
        hi = tile.seg_hi - 1;
        if (lo < hi && segs[lo + 1].vstart <= v) {
            ++lo;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (segs[mid].vstart <= v) lo = mid; else hi = mid - 1;
            }
        }
        const ScoreSeg sg = segs[lo];
        const long long t = v - sg.vstart;
        const real *fp = frames + (sg.frame0 + t) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) x[r][d] = fp[d];
        oidx[r] = sg.out0 + t * (long long)sg.out_stride;
    }

    real mx[R], sm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mx[r] = Fast<real>::neg_big();
        sm[r] = 0;
    }

    // partial-distance elimination (see the mixture loop): the float launches only -- the float64 parity mode evaluates everything
    // -- and not the subset launch of a split state: its mixtures are tight but rarely collapsed (a state whose mixtures have collapsed is off
    //    the pipe as a whole), few groups skip, and the tests cost its three-wave build more than they save (15.8 against 7.3 ms per batch)
    constexpr bool PDE = sizeof(real) == 4 && (!SUBSET || PCL_PDE_SUBSET) && PCL_PDE_D1 > 0 && D > PCL_PDE_D1;
    constexpr int PD1 = PCL_PDE_D1, PD2 = (PCL_PDE_D2 > PCL_PDE_D1 && PCL_PDE_D2 < D) ? PCL_PDE_D2 : PCL_PDE_D1;
    constexpr int PDE_MARGIN = 40;                                   // log2 units below the lane's own running maximum: nothing for an f32 sum that holds that maximum's 1
    constexpr int PDE_MARGIN_PIPE = 64;                              // ... below the matrix pipe's part of a split state: the two parts are merged in float64
    real ref2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ref2[r] = Fast<real>::neg_big();
        if (PDE && SUBSET) {                                         // the rest of the state, as the matrix pipe left it (ln -> log2)
            const double old = out[oidx[r]];
            if (old > -1.0e30) ref2[r] = (real)(old * 1.4426950408889634074) - (real)PDE_MARGIN_PIPE;
        }
    }
    static_assert(!SUBSET || MASTER, "the subset launch makes its rows from the master copy");
    const real *pbase = params + (size_t)tile.state * Mpad * ROW;
    const int n_sub = SUBSET ? mm.nbad[tile.state] : 0;
    const int Mloop = SUBSET ? (n_sub + GROUP - 1) / GROUP * GROUP : Mpad;       // (whole groups: the filler rows are log zero)
    for (int c0 = 0; c0 < Mloop; c0 += CH) {
        const int n = min(CH, Mloop - c0);
        __syncthreads();
        if (MASTER) {
            constexpr double LOG2E = 1.4426950408889634074, LOG_2PI = 1.8378770664093454836;
            const size_t jm0 = (size_t)tile.state * Mpad + c0;
            const size_t js0 = (size_t)tile.state * Mpad;
            // 8 lanes per mixture, 32 mixtures per pass of the workgroup (derive_kernel's arrangement and its order of the per-mixture
            // sums, so k2 has the bits of the derived rows): every load of a pass is in flight at once.  (The first version gave a
            // mixture's constant to ONE thread -- 39 float64 loads in a row per chunk while three waves waited at the barrier: 55 % of
            // the subset launch's time.)
            const int sub = threadIdx.x & 7;
            for (int ml = threadIdx.x >> 3; ml < (n + 31) / 32 * 32; ml += WG / 8) {
                const bool in_chunk = ml < n;
                const bool have = in_chunk && (SUBSET ? (c0 + ml) < n_sub : ((c0 + ml) < mm.M && !(mm.bad && mm.bad[jm0 + ml])));
                const bool rows = in_chunk && (SUBSET ? (c0 + ml) < n_sub : (c0 + ml) < mm.M);      // (a mixture the pipe left out still gets its s, c)
                size_t jm = jm0 + ml;
                if (SUBSET && rows) jm = js0 + mm.bad_idx[js0 + c0 + ml];
                double sumvar = 0.0, sumlog = 0.0;
                for (int d = sub; d < D; d += 8) {
                    real sv = 0, cv = 0;
                    if (rows && d < mm.Dhost) {
                        const double v = mm.var64[jm * D + d], mu = mm.mean64[jm * D + d];
                        const float a = (float)(-LOG2E * (0.5 / v));             // (derive_kernel: s = sqrt(-a) from the f32 coefficient)
                        const float sf = sqrtf(-a);
                        sv = (real)sf;
                        cv = (real)(float)(-mu * (double)sf);
                        sumvar += v;
                        if (mm.flags & PCL_MODEL_LOGDET) sumlog += log(v);
                    }
                    if (in_chunk) {
                        lds[ml * ROW + 2 * d] = sv;
                        lds[ml * ROW + 2 * d + 1] = cv;
                    }
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    sumvar += __shfl_xor(sumvar, o, 64);
                    sumlog += __shfl_xor(sumlog, o, 64);
                }
                if (sub == 0 && in_chunk) {
                    double k2 = -INFINITY;
                    if (have) {                                                  // util.py:29 (quirk Q1): sum(var); the log-determinant only on request
                        const double tail = (mm.flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;
                        k2 = LOG2E * (log(mm.w64[jm]) - 0.5 * mm.Dhost * LOG_2PI - 0.5 * tail);
                    }
                    lds[ml * ROW + 2 * D] = (real)(float)k2;
                }
            }
        } else {
            constexpr int VEC = 16 / sizeof(real);
            const real *src = pbase + (size_t)c0 * ROW;
            for (int i = threadIdx.x * VEC; i < n * ROW; i += WG * VEC) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) lds[i + k] = src[i + k];
            }
        }
        __syncthreads();
        for (int m = 0; m < n; m += GROUP) {
            real v[GROUP][R];
            // Partial-distance elimination (f32 launches).  q only grows with every feature, so k2 - q after the first D1 (then D2) features
            // is an UPPER bound of the mixture's value; when for every frame of the wavefront that bound is already 2^-40 below the
            // frame's running maximum (a term that far below adds nothing to an f32 sum that holds its maximum's 1) or 2^-64 below what
            // the matrix pipe wrote for the rest of a split state (the two parts are merged in float64), the rest of the features
            // cannot change any result and the wavefront skips them (mixtures are wave-uniform here: a ballot, no divergence).  The
            // results keep their bits (tools/em_iter_probe.py hashes them under -DPCL_PDE_D1=0).  What it is for: re-estimated models
            // whose mixtures have collapsed onto single frames (variances at the floor of GMM.update_param, Clustering.py:682-693; the
            // reference's driver passes 1e-6, init.py:30): such a mixture is 1e5 .. 1e6 nats away from every frame but its own, and
            // after a few features that is settled.
            real thr[R];
            if (PDE) {
#pragma unroll
                for (int r = 0; r < R; ++r) thr[r] = Fast<real>::max(mx[r] - (real)PDE_MARGIN, ref2[r]);
            }
            bool any_alive = !PDE;
#pragma unroll
            for (int g = 0; g < GROUP; ++g) {
                const real *p = &lds[(m + g) * ROW];
                const real k2 = p[2 * D];
                real q[R];
#pragma unroll
                for (int r = 0; r < R; ++r) q[r] = 0;
                auto feats = [&](const int d0, const int d1) {
#pragma unroll
                    for (int d = d0; d < d1; ++d) {
                        const real s = p[2 * d], c = p[2 * d + 1];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const real y = Fast<real>::fma(x[r][d], s, c);
                            q[r] = Fast<real>::fma(y, y, q[r]);
                        }
                    }
                };
                auto dead = [&]() {                                              // wave-uniform: no frame of the wavefront can still be reached
                    bool alive = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) alive |= (k2 - q[r] >= thr[r]);
                    return __builtin_amdgcn_ballot_w64(alive) == 0ull;
                };
                bool gone = false;
                if (PDE) {
                    feats(0, PD1);
                    gone = dead();
                    if (!gone && PD2 > PD1) {
                        feats(PD1, PD2);
                        gone = dead();
                    }
                    if (!gone) feats(PD2 > PD1 ? PD2 : PD1, D);
                } else {
                    feats(0, D);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) v[g][r] = gone ? -(real)INFINITY : k2 - q[r];
                any_alive |= !gone;
            }
            if (!any_alive) continue;                                            // (all four gone: the update below would multiply by 1 and add zeros)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                real gm = v[0][r];
#pragma unroll
                for (int g = 1; g < GROUP; ++g) gm = Fast<real>::max(gm, v[g][r]);
                const real nm = Fast<real>::max(mx[r], gm);
                real acc = sm[r] * Fast<real>::exp2(mx[r] - nm);
#pragma unroll
                for (int g = 0; g < GROUP; ++g) acc += Fast<real>::exp2(v[g][r] - nm);
                sm[r] = acc;
                mx[r] = nm;
            }
        }
    }
    constexpr double LN2 = 0.693147180559945309417232121458;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (valid[r]) {
            // all-(-inf) components (every weight zero): sum stays 0 -> log2(0) = -inf, as util.py:63-65
            double res = (sm[r] > 0) ? LN2 * ((double)mx[r] + ::log2((double)sm[r])) : -INFINITY;
            if (SUBSET) {                                                    // ln (e^pipe + e^these): the pipe's part is in the buffer
                const double old = out[oidx[r]];
                const double hi = fmax(old, res), lo = fmin(old, res);
                res = (hi > -INFINITY) ? hi + log1p(exp(lo - hi)) : -INFINITY;
            }
            out[oidx[r]] = res;
        }
    }
  };
    if (!flags) {
        score_tile(blockIdx.x);
        return;
    }
    // Fix-up launch (the tiles a matrix-pipe kernel flagged: normally none).  A FIXED grid reads the flags in one coalesced round --
    // thread t of workgroup b looks at tile b + t gridDim.x -- and rescans what it finds: a workgroup per tile that returns at once cost
    // 30 us per scoring call at the C4 shard (72000 workgroups) and was 10 % of config 2's step.
    __shared__ int todo[WG];
    __shared__ int n_todo;
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    for (int base = blockIdx.x; base < n_tiles; base += WG * gridDim.x) {
        const int ti = base + threadIdx.x * gridDim.x;
        if (ti < n_tiles && flags[ti]) todo[atomicAdd(&n_todo, 1)] = ti;
        __syncthreads();
        const int n = n_todo;
        // (the order of the list does not matter: tiles are independent and each writes its own outputs)
        for (int k = 0; k < n; ++k) {
            score_tile(todo[k]);
            __syncthreads();                                                  // (the next tile's staging overwrites lds)
        }
        __syncthreads();
        if (threadIdx.x == 0) n_todo = 0;
        __syncthreads();
    }
}

// rows of the sentence HMMs that are not GMM states: entry -> 0, exit -> -inf
__global__ void fill_virtual_rows_kernel(const UttDesc *__restrict__ utt, const int32_t *__restrict__ row_state,
                                         double *__restrict__ Bt) {
    const int u = blockIdx.y;
    const UttDesc d = utt[u];
    const long long total = (long long)d.T * d.N;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(e % d.N);
        const int st = row_state[d.vec_off + n];
        if (st == PCL_ROW_ENTRY) Bt[d.b_off + e] = 0.0;
        else if (st == PCL_ROW_EXIT) Bt[d.b_off + e] = -INFINITY;
    }
}

// (N,T) row-major host layout <-> (T,N) time-major device layout, per utterance
__global__ void transpose_kernel(const UttDesc *__restrict__ utt, const double *__restrict__ src,
                                 double *__restrict__ dst, int to_time_major) {
    const UttDesc d = utt[blockIdx.y];
    const long long total = (long long)d.T * d.N;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        // e indexes the destination
        if (to_time_major) {
            const int t = (int)(e / d.N), n = (int)(e % d.N);
            dst[d.b_off + e] = src[d.b_off + (long long)n * d.T + t];
        } else {
            const int n = (int)(e / d.T), t = (int)(e % d.T);
            dst[d.b_off + e] = src[d.b_off + (long long)t * d.N + n];
        }
    }
}

template <int D, int R, int CH, typename real>
void launch_score_t(pcl_ctx *ctx, pcl_batch *b, const real *frames, const real *params, const ScoreTile *tiles, int n_tiles) {
    hipLaunchKernelGGL((gmm_score_kernel<D, R, CH, real>), dim3(n_tiles), dim3(WG), 0, ctx->stream, frames, params,
                       ctx->Mpad, tiles, b->d_segs, b->Bt, (const int *)nullptr, n_tiles, MasterModel{});
}

// frames per lane for each (D, precision); the tile is WG * R frames.  x[R][D] must stay in VGPRs
// (2 waves per SIMD => 256 VGPRs per lane).
constexpr int r32(int D) { return D <= 40 ? PCL_R32 : 2; }
constexpr int r64(int D) { return D <= 40 ? 2 : 1; }
constexpr int rsub(int D) { return D <= 40 ? PCL_RSUB : 2; }

}  // namespace

int pcl_score_tile_frames(int D, int precision) {
    return WG * (precision == PCL_F64 ? r64(D) : r32(D));
}
int pcl_score_subset_tile_frames(int D) { return WG * rsub(D); }

int pcl_launch_fill_virtual_rows(pcl_ctx *ctx, pcl_batch *b) {
    dim3 grid(8, b->U);
    hipLaunchKernelGGL(fill_virtual_rows_kernel, grid, dim3(256), 0, ctx->stream, b->d_utt, b->d_row_state, b->Bt);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_transpose(pcl_ctx *ctx, pcl_batch *b, const double *src, double *dst, int to_time_major) {
    dim3 grid(8, b->U);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, b->d_utt, src, dst, to_time_major);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// direct-form rescoring of the tiles a matrix-pipe kernel flagged (same tile list: R * 256 frames per workgroup)
int pcl_launch_score_fixup(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles, const int *flags) {
    if (n_tiles == 0) return PCL_OK;
    const int tf = pcl_score_split16_tile_frames(), R = tf / WG;      // frames per lane so that a workgroup covers the same tile
    if (tf % WG || (R != 1 && R != 2)) PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: fix-up tile size mismatch");
    pcl_timer_begin(ctx, "score_fixup");
    const MasterModel mm{ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Dhost, ctx->model_flags, ctx->d_bad, nullptr, nullptr};      // (no params32 needed: see MasterModel)
    const int fix_grid = std::min(n_tiles, 1024);                        // (a fixed grid scans the flags: see the kernel)
#define LAUNCHF(DD, RR) hipLaunchKernelGGL((gmm_score_kernel<DD, RR, PCL_CH32, float, true>), dim3(fix_grid), dim3(WG), 0, ctx->stream, ctx->frames32, \
                                           (const float *)nullptr, ctx->Mpad, tiles, b->d_segs, b->Bt, flags, n_tiles, mm)
#define CASEF(DD) case DD: if (R == 1) LAUNCHF(DD, 1); else LAUNCHF(DD, 2); break;
    switch (ctx->D) {
        CASEF(13) CASEF(26) CASEF(39) CASEF(47)
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no fix-up scoring kernel for D=%d", ctx->D);
    }
#undef CASEF
#undef LAUNCHF
    pcl_timer_end(ctx, "score_fixup");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// the off-pipe mixtures of the split states, direct form, log-added to the matrix pipe's result (tiles: the direct-form tile size)
int pcl_launch_score_subset(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    pcl_timer_begin(ctx, "score_subset");
    const MasterModel mm{ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Dhost, ctx->model_flags, ctx->d_bad, ctx->d_bad_idx, ctx->d_nbad};
    switch (ctx->D) {
#define CASES(DD) case DD: hipLaunchKernelGGL((gmm_score_kernel<DD, rsub(DD), PCL_CH32, float, true, true>), dim3(n_tiles), dim3(WG), 0, ctx->stream, ctx->frames32, \
                                               (const float *)nullptr, ctx->Mpad, tiles, b->d_segs, b->Bt, (const int *)nullptr, n_tiles, mm); break;
        CASES(13) CASES(26) CASES(39) CASES(47)
#undef CASES
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no subset scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score_subset");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// ... the same for the tiles the coarse pass flagged (gmm_score_coarse.hip: a scaled feature out of the f16 range): a fixed grid scans the flags,
// one frame per lane so that a workgroup covers a matrix-pipe tile
int pcl_launch_score_subset_flagged(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles, const int *flags) {
    if (n_tiles == 0) return PCL_OK;
    if (pcl_coarse_tile_frames() != WG) PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: coarse tile size mismatch");
    pcl_timer_begin(ctx, "score_subset_fixup");
    const MasterModel mm{ctx->mean64, ctx->var64, ctx->w64, ctx->M, ctx->Dhost, ctx->model_flags, ctx->d_bad, ctx->d_bad_idx, ctx->d_nbad};
    const int fix_grid = std::min(n_tiles, 1024);
    switch (ctx->D) {
#define CASESF(DD) case DD: hipLaunchKernelGGL((gmm_score_kernel<DD, 1, PCL_CH32, float, true, true>), dim3(fix_grid), dim3(WG), 0, ctx->stream, ctx->frames32, \
                                                (const float *)nullptr, ctx->Mpad, tiles, b->d_segs, b->Bt, flags, n_tiles, mm); break;
        CASESF(13) CASESF(26) CASESF(39) CASESF(47)
#undef CASESF
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no subset scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score_subset_fixup");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_score(pcl_ctx *ctx, pcl_batch *b, int precision, const ScoreTile *tiles, int n_tiles) {
    if (n_tiles == 0) return PCL_OK;
    const int D = ctx->D;
    const char *tname = (tiles == b->d_tiles_v) ? "score_direct" : "score";   // the ill-conditioned remainder is timed apart
    if (precision == PCL_F32) TRY(pcl_ensure_layouts(ctx, PCL_LAYOUT_P32));     // (derived on first use: whole states on this kernel are the exception)
    pcl_timer_begin(ctx, tname);
    if (precision == PCL_F32) {
        switch (D) {
#define CASE32(DD) case DD: launch_score_t<DD, r32(DD), PCL_CH32, float>(ctx, b, ctx->frames32, ctx->params32, tiles, n_tiles); break;
            CASE32(13) CASE32(26) CASE32(39) CASE32(47)
            CASE32(48) CASE32(64)
#undef CASE32
            default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f32 scoring kernel for padded D=%d", D);
        }
    } else {
        switch (D) {
#define CASE64(DD) case DD: launch_score_t<DD, r64(DD), 32, double>(ctx, b, ctx->frames64, ctx->params64, tiles, n_tiles); break;
            CASE64(13) CASE64(26) CASE64(39) CASE64(47)
            CASE64(48) CASE64(64)
#undef CASE64
            default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no f64 scoring kernel for padded D=%d", D);
        }
    }
    pcl_timer_end(ctx, tname);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}


// ---------------------------------------------------------------- emission rows that repeat another row of their utterance
namespace {
__global__ void dup_rows_kernel(const DupRow *__restrict__ dups, int n, double *__restrict__ Bt) {
    const DupRow r = dups[blockIdx.x];
    for (int t = threadIdx.x; t < r.T; t += blockDim.x) Bt[r.dst + (long long)t * r.N] = Bt[r.src + (long long)t * r.N];
}
}  // namespace

int pcl_launch_dup_rows(pcl_ctx *ctx, pcl_batch *b) {
    if (b->dups.empty()) return PCL_OK;
    hipLaunchKernelGGL(dup_rows_kernel, dim3((unsigned)b->dups.size()), dim3(256), 0, ctx->stream, b->d_dups, (int)b->dups.size(), b->Bt);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

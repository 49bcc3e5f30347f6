// gmm_score_coarse.hip -- the off-pipe ("tight") mixtures of split states: a COARSE pass on the f16 matrix pipe that proves almost
// every (frame, mixture) pair negligible, and a direct-form evaluation of the few pairs it cannot (round 6).
//
// Reference rows: as gmm_score.hip (A1/A4/A6: util.gaussian_function util.py:20-31, Clustering.GMM.point Clustering.py:740-767); the
// models in question are the ones Clustering.GMM.update_param (Clustering.py:682-693, variance floor init.py:30 -> Controller.py:151)
// leaves after one or two M-steps: mixtures much tighter than their state's spread.
//
// Why.  A mixture whose cancelling term cond_m = log2e sum_d (mu_md - c_jd)^2 / (2 var_md) is beyond cond_max cannot be evaluated by the
// centred f32-class expansion of gmm_score_split.hip to 5e-5 nats (error ~ 5e-7 cond_m), so it is left out of the matrix-pipe layouts
// and evaluated in direct form, 78 FMAs per (frame, mixture) pair on the VALU (gmm_score.hip, SUBSET): at config 4's third EM iteration
// 57 % of the mixtures are such, and the direct-form launch is 52 ms per 1024-utterance batch beside the pipe's 15 (whole states in
// direct form, the round 4-5 route: 66 ms).  But 99.9 % of those pairs contribute nothing: a tight mixture is hundreds of nats below
// the frame's likelihood for every frame that is not within a few of ITS sigmas.  Proving that needs no precision, only a BOUND:
//
//   v_up[f,m] = k2_m - log2e sum_d (x_fd - mu_md)^2 / (2 var'_md),   var' = max(var, vfloor_m)  >=  the mixture's true log2 value
//
// (a larger variance can only raise the exponent; the constant k2_m is the TRUE one -- quirk Q1's constant has no determinant).  v_up is
// the same contraction over [x'^2, x', 1] as the main kernel's, with coefficients that fit the f16 pieces because of the floor:
// vfloor_m = max(2^-10, log2e/2 sum_d (mu_md - c_jd)^2 / 2.5e4) bounds a' = log2e / (2 var') by 739 and the cancelling term by 2.5e4.
// The pipe computes it with an absolute error below E_m = 2^-20 (|k''_m - K0| + 8 kq'_m + 2 |k2_m|) + 0.25 (a few times the nominal
// 2^-22 of the two-piece products; kq' = the cancelling term under var'), which is ADDED to the mixture's folded constant.  With the
// spare-slot reference set to  t_f = log2e ln b1_f - 36 - K0  (ln b1 = what the pipe wrote for the state's on-pipe mixtures; 36: a
// term 2^-36 below the total changes ln b by 1e-11 even if all 2048 were dropped at the threshold), the accumulator of a pair is
//   acc >= 0   <=>   the pair MAY reach 2^-36 of the frame's likelihood,
// one v_max tree + one ballot per 32 x 32 tile instead of the log-sum-exp.  Pairs that pass are evaluated in DIRECT FORM, a lane per pair
// (rows made from the master copy in the f32 arithmetic of the direct-form kernels, see `flush`), into a per-frame online log-sum-exp
// (float64); the threshold is raised when such a value lifts the frame's maximum and what is still waiting is tested again (a state
// whose on-pipe part says little settles after the first rows of its first tile); the result is log-added to the pipe's in float64:
// ln b = ln(e^pipe + e^tight), the reference's sum over all mixtures -- deterministic (a lane owns its frame; no atomics).  Frames whose
// scaled features leave the f16 range raise the tile's flag and the direct-form subset kernel rescoring flagged tiles follows in the
// same call, as for the main kernel.  States stay on this route up to 99 % off-pipe mixtures (pcl_model_upload); beyond, the on-pipe
// part is too thin a reference (with none at all and every mixture collapsed the bound under the floored variance sits far above the
// true values and every pair passes) and whole states take the direct form with its partial-distance test.
//
// Layout of the tight mixtures: [J][Mpad32/32 tiles][2 pieces][KS8][64 lanes][8 f16], the state's bad_idx list in order, 32 per tile,
// nct[j] tiles used; own power-of-two feature scales fscale_c and K0 (kzero_c); constants k2c[j][idx] (float64).  Derived on first use
// after a model change (pcl_ensure_coarse), one workgroup per state.  The same file holds compact_main_kernel: a split state's ON-pipe
// mixtures compacted to the front of the main layout, so that the matrix-pipe kernels walk ceil(on-pipe / 32) tiles.
#include <stdlib.h>

#include <vector>

#include "pcl_internal.h"

namespace {

#ifndef PCL_COARSE_NT
#define PCL_COARSE_NT 2
#endif
constexpr int WG = 64 * (256 / (32 * PCL_COARSE_NT));     // a workgroup covers 256 frames: 4 waves of two 32-frame groups (NT = 1: 8 waves of one -- measured slower)
static_assert(PCL_COARSE_NT == 1 || PCL_COARSE_NT == 2, "32-frame groups per wave");
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr double LOG2E = 1.4426950408889634074, LOG_2PI = 1.8378770664093454836;
#ifdef PCL_COARSE_MARGIN_REPRO                // mutation build (tools/gpu_mutation_check.sh: the tests are expected to FAIL on it): a pair as large as
constexpr float COARSE_MARGIN = 0.f;          // everything the pipe summed is ruled out
#else
constexpr float COARSE_MARGIN = 36.f;         // log2 units below the frame's likelihood: nothing (see above)
#endif
constexpr double VMIN_C = 0.0009765625;       // 2^-10: the variance floor of the bound
constexpr double KQ_MAX = 2.5e4;              // the cancelling term the folded constant may carry (f16 pieces reach 6e4)
constexpr double EPS_C = 9.5367431640625e-07; // 2^-20
// The one-product pass (NP = 1).  Operands are single f16 values: a coefficient and a feature each carry a relative error 2^-11, a product
// (1 + d1)(1 + d2) - 1 <= 2^-10 (1 + 2^-12); the matrix pipe's f32 sums add ~2^-20 of the terms.  QUADRATIC terms a x'^2 (a <= 0) need no
// allowance: both factors are rounded toward zero, so the term can only come out too HIGH (the bound stays a bound).  LINEAR terms b x'
// change sign: with u_d = sqrt(a'_d) x'_d, w_d = sqrt(a'_d) (mu_d - c_d) (kq' = sum w^2, q' = sum (u - w)^2 = the distance term of v_up),
//   sum |b x'| = 2 sum |u w| <= sum u^2 / 2 + 2 kq' <= q' + 3 kq'          (u^2 <= 2 (u - w)^2 + 2 w^2)
// and a pair that matters has q' <= k2_m - threshold <= G_f := k2max_j - threshold_f.  So the error is below EPS1 (G_f + 3 kq'_m): the
// mixture's part goes into its folded constant (E_m += EPS1 3 kq'_m, and 2^-10 |constant| for the constant's own single piece), the
// frame's part lowers the frame's threshold (set_threshold).  For a collapsed mixture (kq' = 2.5e4) that is ~75-125 log2 units on a
// distance of tens of thousands; for a moderately tight one (kq' in the hundreds) about one.
constexpr double EPS1 = 0.0009765625 * 1.02;  // 2^-10 (1 + 2^-12) + the f32 sums, rounded up

struct CoarseExact {
    const float2 *rows;          // [J][Mpad][D] (s_d, c_d) of the tight mixtures in bad_idx order: the direct form's f32 rows (coarse_derive_kernel)
    const double *k2c;
    const int *nbad;
    int Mpad, Dhost;
};

#ifndef PCL_COARSE_MINW
#define PCL_COARSE_MINW 2    // waves per SIMD the register allocation aims at: 2 (212 VGPRs at D = 39 with one product, nothing spilled).  3 measured 4 % faster on the shard probe and 8-10 % slower inside config 4's EM iterations on the first form of the kernel (profiles/r06_coarse_ab.txt)
#endif

// f16 rounded TOWARD ZERO (the one-product pass: a quadratic term a x'^2, a <= 0, may only come out too high)
__device__ __forceinline__ _Float16 f16_toward_zero(float val) {
    _Float16 h = (_Float16)val;
    if (__builtin_fabsf((float)h) > __builtin_fabsf(val)) h = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, h) - 1));
    return h;
}

// NP = 3: the two-piece operands and their three products (round 6's first form, error 2^-20 of the terms); NP = 1: ONE product of the
// leading pieces -- a third of the matrix-pipe work and half the layout traffic -- with what that costs in precision paid for in the
// bound instead (see EPS1 above): the pairs that pass are evaluated in direct form either way.
template <int D, int NT, int NP>
__global__ __launch_bounds__(WG, PCL_COARSE_MINW) void gmm_score_coarse_kernel(
    const float *__restrict__ frames, const uint4 *__restrict__ pm, const float *__restrict__ fscale, const float *__restrict__ centers,
    int nmt_max, const int *__restrict__ nct, const ScoreTile *__restrict__ tiles, const ScoreSeg *__restrict__ segs, double *__restrict__ out,
    int *__restrict__ flags, const double *__restrict__ kzero, const float *__restrict__ kgap, CoarseExact ex,
    unsigned long long *__restrict__ counters) {
    static_assert(D % 8 != 0, "the folded constants need a spare slot");
    constexpr int KS8 = (D + 7) / 8;
    constexpr int CH = 2 * KS8;
    constexpr int SC = D / 8, JC = D % 8;
    constexpr float FMAXH = 6.0e4f, TMAX = 5.0e4f;
    const ScoreTile tile = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int n_mtiles = (tile.seg_lo >= tile.seg_hi) ? 0 : nct[tile.state];
    if (n_mtiles == 0) {                      // padding tile of the XCD-aware order, or a state without tight mixtures
        if (threadIdx.x == 0) flags[blockIdx.x] = 0;
        return;
    }
    __shared__ int s_ovf, s_bail[2];
    if (threadIdx.x == 0) s_ovf = s_bail[0] = s_bail[1] = 0;
    __syncthreads();
    const int vend = segs[tile.seg_hi - 1].vstart + segs[tile.seg_hi - 1].len;
    const bool wave_active = tile.vstart + wave * NT * 32 < vend;

    // ---- B operand: as gmm_score_split16_kernel, with the coarse layout's scales
    constexpr int NPB = NP == 1 ? 1 : 2;
    constexpr int PCS = NP == 1 ? KS8 : CH, MT = NP == 1 ? 4 : 2;
    constexpr int SLOTS = NT * 32;
    static_assert(SLOTS <= 64, "a lane per frame slot");
    __shared__ __attribute__((aligned(16))) uint4 abuf[2][MT * PCS * 64];   // the stages of the mixture tiles (below); first, the wave's frames
    __shared__ long long frow_tab[WG / 64][SLOTS];
    static_assert(sizeof(uint4) * 2 * MT * PCS * 64 >= sizeof(float) * (WG / 64) * SLOTS * D, "the frames of a workgroup fit the stage buffers");
    h8v xb[NT][NPB][KS8];
    long long oidx[NT], frow[NT];
    bool valid[NT];
    const float *cen = centers + (size_t)tile.state * D;
    const float *fs = fscale + ((size_t)tile.state * 2 + half) * (KS8 * 8);
    bool ovf = false;
#pragma unroll
    for (int c = 0; c < NT; ++c) {                               // which frame each column is (the tile's segments: ragged utterances)
        int v = tile.vstart + (wave * NT + c) * 32 + col;
        valid[c] = v < vend;
        if (!valid[c]) v = tile.vstart;
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
        frow[c] = sg.frame0 + t;
        oidx[c] = sg.out0 + t * (long long)sg.out_stride;
        if (half == 0) frow_tab[wave][c * 32 + col] = frow[c];
    }
    // The wave's 64 frames come through LDS: element e = column * D + d of a 32-frame group is loaded by lane e % 64 -- consecutive lanes,
    // consecutive addresses (frames of a segment are adjacent) -- straight into the stage buffer (which the mixture tiles need only after
    // the barrier below), and a column reads its own row back.  A lane loading its own frame's 39 values one by one was 78 load
    // instructions of 32 cache lines each per wave: 2.2 ms per batch of texture-unit time, more than the products of 40 mixture tiles.
    float *stg = reinterpret_cast<float *>(&abuf[0][0]) + wave * (SLOTS * D);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int i = 0; i < (32 * D + 63) / 64; ++i) {
            const int e = i * 64 + lane;
            if (e < 32 * D) {
                const int cx = e / D, d = e - cx * D;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(frames + frow_tab[wave][c * 32 + cx] * D + d),
                                                 (__attribute__((address_space(3))) void *)(stg + c * 32 * D + i * 64), 4, 0, 0);
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const float *fp = stg + (c * 32 + col) * D;
#pragma unroll
        for (int s = 0; s < KS8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * s + j;
                float val = 0.f;
                if (d < D) {
                    const float xc = fp[d] - cen[d];
                    val = (half ? xc : xc * xc) * fs[d];
                    ovf |= valid[c] && __builtin_fabsf(val) > FMAXH;
                    val = __builtin_fminf(__builtin_fmaxf(val, -FMAXH), FMAXH);
                }
                if (d == D) val = half ? 0.f : 1.f;              // x1: [1 | -t (set below)]
                if constexpr (NP == 1) {
                    xb[c][0][s][j] = (half || d >= D) ? (_Float16)val : f16_toward_zero(val);
                } else {
                    const _Float16 h1 = (_Float16)val;
                    xb[c][0][s][j] = h1;
                    xb[c][NPB - 1][s][j] = (_Float16)(val - (float)h1);
                }
            }
    }
    if (__any(ovf) && lane == 0) s_ovf = 1;
    __syncthreads();
    if (s_ovf) {                              // a frame out of the f16 range: the direct-form subset kernel rescoring flagged tiles does this one
        if (threadIdx.x == 0) flags[blockIdx.x] = 1;
        return;
    }
    if (threadIdx.x == 0) flags[blockIdx.x] = 0;

    // ---- per frame: what the pipe wrote (ln b of the on-pipe mixtures) and the threshold t = log2e ln b1 - margin - K0 its spare slot carries
    const float k0f = (float)kzero[tile.state];
    const float kg = (NP == 1) ? kgap[tile.state] : 0.f;      // >= (the state's largest k2) - K0
    double pipe_ln[NT];
    float tcur[NT];                           // (f16-exact)
    auto set_threshold = [&](int c, float t) {
        // one product: the frame's share of the linear terms' rounding, EPS1 (k2max - threshold) (see coarse_derive_kernel)
        if constexpr (NP == 1) t -= (float)EPS1 * __builtin_fmaxf(kg - t, 0.f);
        // rounded so that the f16 value is not ABOVE what was asked for (a higher threshold could miss a pair): less 2^-10 |t|
        t = t - __builtin_fabsf(t) * 0.0009765625f;
        t = __builtin_fminf(__builtin_fmaxf(t, -TMAX), TMAX);
        const _Float16 r1 = (_Float16)(-t);
        tcur[c] = -(float)r1;
        if (half) xb[c][0][SC][JC] = r1;
    };
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        pipe_ln[c] = valid[c] ? out[oidx[c]] : 0.0;
        const double t2 = pipe_ln[c] * LOG2E - (double)COARSE_MARGIN - (double)k0f;
        set_threshold(c, valid[c] ? (t2 > -1.0e30 ? (float)t2 : -TMAX) : TMAX);
    }

    // A stage = MT mixture tiles (one product: 4 tiles' leading pieces, 20 KB at D = 39; three: 2 tiles, both pieces), by LDS-DMA one
    // stage ahead.  The kernel stays latency bound (matrix pipe 17 % busy at a clock the power cap does not reach, SQ_WAIT_ANY 61 % of the
    // wave time: profiles/r06_coarse_summary.txt) and the obvious suspects were each measured and are NOT it (profiles/r06_coarse_rework.txt):
    // the compiler's s_waitcnt vmcnt(0) before LDS reads that follow an LDS-DMA (reads through inline asm: 10.8 ms against 10.3), the
    // stages themselves (global loads to registers + ds_write, the classic double buffer: 14.1; every wave reading its A operand from
    // global memory with no stage and no barrier at all: 9.3), occupancy (one 32-frame group per wave, 8 waves per workgroup, at 2 / 3 / 4
    // waves per SIMD: 17.6 / 15.3 / 12.5).  What is left is the serial shape of a wave's own work: products -> max tree -> branch.
    const uint4 *pstate = pm + (size_t)tile.state * nmt_max * (CH * 64);
    auto dma = [&](int buf, int stage) {
        const int nt = min(MT, n_mtiles - stage * MT);
        const uint4 *src = pstate + (size_t)stage * MT * (CH * 64);
        for (int r = wave; r < nt * PCS; r += WG / 64) {
            const int i = r / PCS, p = r - i * PCS;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)i * (CH * 64) + p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)&abuf[buf][r * 64], 16, 0, 0);
        }
    };
    const int n_tight = ex.nbad[tile.state];
    const size_t srow = (size_t)tile.state * ex.Mpad;
    unsigned long long n_cand = 0;

    // ---- the pairs that pass wait in a queue of their wave and are evaluated 64 at a time, a lane per PAIR (one lane at a time beside
    //      63 idle ones, with its 117 dependent loads, made the exact part ten times the matrix pipe's: 43 ms per batch at 73 % off-pipe
    //      mixtures).  Queue order = (m-tile, frame slot c, accumulator row, lane): fixed, so the sums are too.  Lane L owns frame slot
    //      L = c * 32 + col: the online log-sum-exp (log2 domain, float64) of the slot's exact values.
    constexpr int QCAP = 64;
    __shared__ int q_idx[WG / 64][QCAP], q_slot[WG / 64][QCAP];
    __shared__ double q_val[WG / 64][QCAP];
    int qn = 0;                               // wave-uniform
    // The way out.  The bound is only as good as its reference: a state whose few on-pipe mixtures are themselves nearly collapsed gives
    // frames far from all of them a likelihood so low that every tight mixture passes, and a lane per pair with gathered rows is several
    // times the direct-form kernel's cost per pair (config 4's seventh EM iteration: 673 such states, 252 ms of coarse pass against the
    // 45 ms their direct form takes).  A wave that has evaluated more than max(4096, 2 x tight mixtures) pairs -- 3 % of its 64 frames'
    // -- gives up: the workgroup raises the tile's flag at the next stage boundary and the direct-form subset kernel rescoring flagged
    // tiles does the tile (nothing has been written yet: results leave the kernel at its end).
    int n_eval = 0;                           // wave-uniform
    double tmaxL = -INFINITY, tsumL = 0.0;
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < qn) {
            const int idx = q_idx[wave][lane];
            double v = -INFINITY;
#ifdef PCL_COARSE_EXP                         // timing builds only (tools/coarse_time_probe.py): the launch WITHOUT a part, wrong results on purpose --
            if (idx < n_tight && !(PCL_COARSE_EXP & 1)) {        // 1: the surviving pairs are not evaluated, 2: no products, 4: nothing passes
#else
            if (idx < n_tight) {
#endif
                const double k2 = ex.k2c[srow + idx];
                if (k2 > -1.0e300) {                             // (not a zero weight)
                    const float2 *rw = ex.rows + (srow + idx) * D;
                    const float *fp = frames + frow_tab[wave][q_slot[wave][lane]] * D;
                    // The direct form in the f32 arithmetic of the accumulate pass's subset kernel (gmm_accumulate.hip MASTER rows: s = sqrtf of the
                    // f32 coefficient, c = (float)(-mu s), y = fma(x, s, c), q = fma(y, y, q) over d ascending, k2 as a float) -- on purpose.  A first
                    // version evaluated these pairs in float64; the soak then failed 18 of 1520 random E-steps on `acc` by 1.2-3.9e-4: the posterior the
                    // accumulate pass forms, exp2(v_m - ln b), is exp2 of the DIFFERENCE of its own f32 value of a tight mixture and the scoring's value,
                    // and where one such mixture carries the frame the two must round alike to cancel (as they did in rounds 4-5, both direct form).
                    // The rows (s, c) come ready from coarse_derive_kernel: made here from the float64 master copy (a division, a square root and
                    // two 8-byte loads from random rows of two 1.9 GB arrays per term, one after the other) a flush took ~50 us and the evaluation
                    // of 0.02 % of the pairs 36 % of the launch (profiles/r06_coarse_parts.txt).
                    constexpr int CHK = 13;                      // (terms loaded together: 39 registers in flight, not 117)
                    float q = 0.f;
#pragma unroll
                    for (int d0 = 0; d0 < D; d0 += CHK) {
                        float2 r[CHK];
                        float xv[CHK];
#pragma unroll
                        for (int i = 0; i < CHK; ++i)
                            if (d0 + i < D) {
                                r[i] = rw[d0 + i];
                                xv[i] = fp[d0 + i];
                            }
#pragma unroll
                        for (int i = 0; i < CHK; ++i)
                            if (d0 + i < D && d0 + i < ex.Dhost) {
                                const float y = __builtin_fmaf(xv[i], r[i].x, r[i].y);
                                q = __builtin_fmaf(y, y, q);
                            }
                    }
                    v = (double)((float)k2 - q);
                    ++n_cand;
                }
            }
            q_val[wave][lane] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // The values into their frames' log-sum-exps, in queue order.  In three steps so that the float64 exp2 -- a hundred instructions --
        // runs twice per flush for all lanes at once: (1) the owner lane of a frame slot takes the maximum of its running one and its new
        // values; (2) every entry's lane forms exp2(value - its slot's maximum), every owner rescales its running sum; (3) the owner adds
        // its entries' terms in queue order.  (An online update per entry ran the exp2 once per ENTRY with one lane active: with ~17
        // entries a flush that was most of the time the direct-form evaluation took.)
        double nm = tmaxL;
        bool got = false;
#pragma unroll 1
        for (int i = 0; i < qn; ++i)
            if (q_slot[wave][i] == lane) {
                const double vi = q_val[wave][i];
                if (vi > -1.0e300) {
                    nm = ::fmax(nm, vi);
                    got = true;
                }
            }
        {
            const int my_slot = lane < qn ? q_slot[wave][lane] : lane;
            const double vmine = lane < qn ? q_val[wave][lane] : -INFINITY;
            const double nm_slot = __shfl(nm, my_slot, 64);
            const double e = (vmine > -1.0e300) ? ::exp2(vmine - nm_slot) : 0.0;
            if (got) {                                       // (tsumL == 0 with tmaxL = -inf before the first value: 0 * exp2(-inf) = 0)
                tsumL *= ::exp2(tmaxL - nm);
                tmaxL = nm;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane < qn) q_val[wave][lane] = e;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int i = 0; i < qn; ++i)
            if (q_slot[wave][i] == lane) tsumL += q_val[wave][i];
        n_eval += qn;
        qn = 0;
        // an exact value that lifts a frame's maximum raises its threshold
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const double tm = __shfl(tmaxL, c * 32 + col, 64);
            const float want = (float)(tm - (double)COARSE_MARGIN - (double)k0f);
            if (valid[c] && tm > -1.0e300 && want > tcur[c] + 4.f) set_threshold(c, want);
        }
    };

    auto process = [&](int mt, const uint4 *ab) {
        f16v acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // (the compiler puts s_waitcnt vmcnt(0) before these reads -- LDS-DMA writes LDS -- so a tile's first read waits for the next stage's
        //  loads; reading through inline asm instead, without that wait, measured no faster: 10.8 ms against 10.3, profiles/r06_coarse_parts.txt)
        auto pass = [&](int pa, int pb) {
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const h8v a = *reinterpret_cast<const h8v *>(&ab[(pa * KS8 + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[c][pb][s], acc[c], 0, 0, 0);
            }
        };
        if constexpr (NP != 1) {
            pass(1, 0);
            pass(0, NPB - 1);
        }
#ifdef PCL_COARSE_EXP
        if (!(PCL_COARSE_EXP & 2))
#endif
        pass(0, 0);
#ifdef PCL_COARSE_EXP
        if (PCL_COARSE_EXP & 2) for (int c = 0; c < NT; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = -1.f - (float)ab[lane].x * 1e-30f;
        if (PCL_COARSE_EXP & 4) for (int c = 0; c < NT; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = -1.f - __builtin_fabsf(acc[c][r]) * 1e-30f;
#endif
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float gm = acc[c][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) gm = __builtin_fmaxf(gm, acc[c][r]);
            if (!__any(gm >= 0.f)) continue;                     // the common case: nobody of this tile can reach any of the 32 frames
            unsigned int mask = 0u;                              // (the rows that passed, as bits: no dynamic index into the accumulator registers)
#pragma unroll
            for (int r = 0; r < 16; ++r) mask |= (acc[c][r] >= 0.f ? 1u : 0u) << r;
            const float t_mfma = tcur[c];                        // the threshold these accumulators are relative to
            unsigned int rows_any = 0u;                          // (wave-uniform: the accumulator rows in which some lane passed -- one or two of the 16)
#pragma unroll
            for (int r = 0; r < 16; ++r) rows_any |= (__ballot(acc[c][r] >= 0.f) != 0ull ? 1u : 0u) << r;
            rows_any = __builtin_amdgcn_readfirstlane(rows_any);
            while (rows_any) {
                const int r = __builtin_ctz(rows_any);
                rows_any &= rows_any - 1u;
                bool hit = (mask >> r) & 1u;
                unsigned long long bal = __ballot(hit);
                if (!bal) continue;
                int cnt = __popcll(bal);
                if (qn + cnt > QCAP) {
                    flush();
                    // the flush may have raised this frame's threshold (a state whose on-pipe part says little: the first rows of its first tile
                    // all pass): what is still waiting in `mask` is tested again before it is queued
                    const float dlt = tcur[c] - t_mfma;
                    if (__any(dlt > 0.f)) {
                        unsigned int again = 0u;
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) again |= (acc[c][rr] >= dlt ? 1u : 0u) << rr;
                        mask &= again;
                        hit = (mask >> r) & 1u;
                        bal = __ballot(hit);
                        if (!bal) continue;
                        cnt = __popcll(bal);
                    }
                }
                if (hit) {
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    q_idx[wave][pos] = mt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    q_slot[wave][pos] = c * 32 + col;
                }
                qn += cnt;
            }
        }
    };

    const int n_stages = (n_mtiles + MT - 1) / MT;
    const int budget = max(4096, 2 * n_tight);
    dma(0, 0);
    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (slot st & 1 was written during stage st - 1 and is read by every wave after this barrier; what stage st raises goes to the other slot)
        if (s_bail[st & 1]) {
            if (threadIdx.x == 0) flags[blockIdx.x] = 1;
            if (counters && lane == 0 && n_eval) atomicAdd(counters, (unsigned long long)n_eval);
            if (counters && threadIdx.x == 0) atomicAdd(counters + 1, 1ull);
            return;
        }
        if (st + 1 < n_stages) dma((st + 1) & 1, st + 1);
        if (wave_active) {
#pragma unroll 1
            for (int i = 0; i < MT && n_eval <= budget; ++i)
                if (st * MT + i < n_mtiles) process(st * MT + i, &abuf[st & 1][i * PCS * 64]);
            if (n_eval > budget && lane == 0) s_bail[(st + 1) & 1] = 1;
        }
    }
    if (qn) flush();
    constexpr double LN2 = 0.693147180559945309417232121458;
    if (lane < SLOTS && tsumL > 0.0) {
        // lane L = c * 32 + col finishes frame slot L: its own col, the c-th of its frames
        bool ok = false;
        long long oi = 0;
        double old = 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c)
            if ((lane >> 5) == c) {
                ok = valid[c];
                oi = oidx[c];
                old = pipe_ln[c];
            }
        if (ok) {
            const double res = LN2 * (tmaxL + ::log2(tsumL));
            const double hi = ::fmax(old, res), lo = ::fmin(old, res);
            out[oi] = (hi > -INFINITY) ? hi + ::log1p(::exp(lo - hi)) : -INFINITY;
        }
    }
    if (counters) {                           // diagnostics (PCL_COARSE_STATS=1): pairs evaluated in direct form
        for (int o = 32; o >= 1; o >>= 1) n_cand += __shfl_xor(n_cand, o, 64);
        if (lane == 0 && n_cand) atomicAdd(counters, n_cand);
    }
}

// ---------------------------------------------------------------- the coarse layout of one state's tight mixtures
// One workgroup per state.  Pass A: per-feature maxima of the coefficients under the floored variances and the largest k''; pass B:
// the tiles.  8 lanes per mixture, 32 mixtures per step, derive_kernel's arrangement (model_derive.hip).
template <int DMAX>
__global__ __launch_bounds__(256) void coarse_derive_kernel(const double *__restrict__ mean64, const double *__restrict__ var64,
                                                            const double *__restrict__ w64, const float *__restrict__ centers, int M, int Mpad,
                                                            int Mpad32, int D, int Dhost, int flags, const int *__restrict__ bad_idx,
                                                            const int *__restrict__ nbad, uint4 *__restrict__ pmc, float *__restrict__ fscale_c,
                                                            double *__restrict__ kzero_c, double *__restrict__ k2c, int *__restrict__ nct,
                                                            int np, float *__restrict__ kgap_c, float2 *__restrict__ rows_c) {
    const int j = blockIdx.x, tid = threadIdx.x;
    const int nb = nbad[j], ntl = (nb + 31) / 32, nmt = Mpad32 / 32, KS8 = (D + 7) / 8;
    if (tid == 0) {
        nct[j] = ntl;
        kgap_c[j] = 0.f;
    }
    if (nb == 0) return;
    __shared__ float fa[32 * DMAX], fb[32 * DMAX], cen[DMAX], isc[2 * 64];
    __shared__ float kc[32];
    __shared__ unsigned int mxa[64], mxb[64];
    __shared__ int kbits, k2bits;
    for (int d = tid; d < D; d += 256) cen[d] = centers[(size_t)j * D + d];
    if (tid < 64) mxa[tid] = mxb[tid] = 0u;
    if (tid == 0) kbits = k2bits = (int)0x80808080;
    __syncthreads();
    const int ml = tid >> 3, sub = tid & 7;
    const size_t srow = (size_t)j * Mpad;
    constexpr int PER = (DMAX + 7) / 8;
    // one tile's rows: coefficients under the floored variance into fa / fb (pass B) or the maxima (pass A); returns k'' (lane sub == 0)
    auto rows = [&](int t, bool write, double &k2_out, double &kq_out, bool &real_out) {
        const int idx = t * 32 + ml;
        const bool real_m = idx < nb;
        const size_t jm = (srow + (real_m ? bad_idx[srow + idx] : 0)) * D;
        double v[PER], dm[PER];
        double s2 = 0.0, sumvar = 0.0, sumlog = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int dd = sub + 8 * k;
            v[k] = 1.0;
            dm[k] = 0.0;
            if (dd < D && dd < Dhost && real_m) {
                v[k] = var64[jm + dd];
                dm[k] = mean64[jm + dd] - (double)cen[dd];
                s2 += dm[k] * dm[k];
                sumvar += v[k];
                if (flags & PCL_MODEL_LOGDET) sumlog += log(v[k]);
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s2 += __shfl_xor(s2, o, 64);
            sumvar += __shfl_xor(sumvar, o, 64);
            sumlog += __shfl_xor(sumlog, o, 64);
        }
        const double vfloor = fmax(VMIN_C, 0.5 * LOG2E * s2 / KQ_MAX);
        double kq = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int dd = sub + 8 * k;
            if (dd < D) {
                float a = 0.f, b = 0.f;
                if (dd < Dhost && real_m) {
                    const double ap = LOG2E * 0.5 / fmax(v[k], vfloor);
                    kq += ap * dm[k] * dm[k];
                    a = (float)(-ap);
                    b = (float)(2.0 * ap * dm[k]);
                    if (!write) {
                        atomicMax(&mxa[dd], __float_as_uint(-a));
                        atomicMax(&mxb[dd], __float_as_uint(fabsf(b)));
                    }
                }
                if (write) {
                    fa[ml * D + dd] = a;
                    fb[ml * D + dd] = b;
                    if (real_m) {
                        // the direct form's row of this mixture, in the arithmetic of the accumulate pass's MASTER rows (gmm_accumulate.hip) -- see `flush`
                        float2 rc = make_float2(0.f, 0.f);
                        if (dd < Dhost) {
                            const float a32 = (float)(-LOG2E * (0.5 / v[k]));
                            const float sf = sqrtf(-a32);
                            rc = make_float2(sf, (float)(-mean64[jm + dd] * (double)sf));
                        }
                        rows_c[(srow + idx) * D + dd] = rc;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) kq += __shfl_xor(kq, o, 64);
        double k2 = -INFINITY;
        if (real_m) {
            const double tail = (flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;      // util.py:29 (quirk Q1)
            k2 = LOG2E * (log(w64[srow + bad_idx[srow + idx]]) - 0.5 * Dhost * LOG_2PI - 0.5 * tail);
        }
        k2_out = k2;
        kq_out = kq;
        real_out = real_m;
    };
    // ---- pass A
    float kmax = -INFINITY, k2max = -INFINITY;
    bool any = false;
    for (int t = 0; t < ntl; ++t) {
        double k2, kq;
        bool real_m;
        rows(t, false, k2, kq, real_m);
        if (sub == 0 && real_m) {
            k2c[srow + t * 32 + ml] = k2;
            if (k2 > -INFINITY) {
                const float kp = (float)(k2 - kq), k2f = (float)k2;
                kmax = any ? fmaxf(kmax, kp) : kp;
                k2max = any ? fmaxf(k2max, k2f) : k2f;
                any = true;
            }
        }
    }
    if (any) {
        const int i = __float_as_int(kmax), i2 = __float_as_int(k2max);
        atomicMax(&kbits, i >= 0 ? i : i ^ 0x7fffffff);
        atomicMax(&k2bits, i2 >= 0 ? i2 : i2 ^ 0x7fffffff);
    }
    __syncthreads();
    for (int t = tid; t < 2 * KS8 * 8; t += 256) {
        const int h = t / (KS8 * 8), dd = t % (KS8 * 8);
        const float mx = (dd < Dhost && dd < 64) ? __uint_as_float(h ? mxb[dd] : mxa[dd]) : 0.f;
        int e = 1;
        if (mx > 0.f && mx < 1e30f) (void)frexpf(mx, &e);             // mx = f 2^e, f in [0.5, 1): the largest coefficient lands in [1, 2)
        e = min(max(e - 1, -60), 60);
        const float sc = ldexpf(1.0f, e);
        fscale_c[((size_t)j * 2 + h) * (KS8 * 8) + dd] = sc;
        isc[h * 64 + dd] = 1.0f / sc;
    }
    double k0 = 0.0;
    {
        const int i = kbits;
        const float f = __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
        if (i != (int)0x80808080 && f > -3.0e38f) k0 = (double)f;
    }
    if (tid == 0) {
        kzero_c[j] = k0;
        const int i2 = k2bits;
        const float f2 = __int_as_float(i2 >= 0 ? i2 : i2 ^ 0x7fffffff);
        // (the state's largest k2) - K0, rounded up: what a frame's threshold is measured from in the one-product pass
        kgap_c[j] = (i2 != (int)0x80808080 && f2 > -3.0e38f) ? (float)((double)f2 - k0) + 1.0f + 1.0e-6f * fabsf(f2) : 0.f;
    }
    __syncthreads();
    // ---- pass B
    for (int t = 0; t < ntl; ++t) {
        double k2, kq;
        bool real_m;
        rows(t, true, k2, kq, real_m);
        if (sub == 0) {
            float c = -6.0e4f;                                    // log zero: padding rows, zero weights
            if (real_m && k2 > -INFINITY) {
                const double kpp = k2 - kq - k0;
                double em = EPS_C * (fabs(kpp) + 8.0 * kq + 2.0 * fabs(k2)) + 0.25;
                if (np == 1) em += EPS1 * 3.0 * kq + 0.5;         // the mixture's share of the linear terms' rounding + what f16 subnormals lose
                double kk = kpp + em;
                if (np == 1) kk += 0.0009765625 * fabs(kk);       // its own single piece (2^-11 |kk| to nearest), and (float) below
                c = (kk > -5.0e4) ? (float)kk : -6.0e4f;          // (cannot be: kq <= KQ_MAX)
            }
            kc[ml] = c;
        }
        __syncthreads();
        uint4 *pf = pmc + ((size_t)j * nmt + t) * (2 * KS8 * 64);
        for (int e = tid; e < 2 * KS8 * 64; e += 256) {
            const int p = (e >> 6) / KS8, s = (e >> 6) % KS8, ln = e & 63, half = ln >> 5, cl = ln & 31;
            unsigned short h[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int dd = 8 * s + x;
                float val = 0.f;
                bool is_const = false;
                if (dd < D) {
                    val = (half ? fb[cl * D + dd] : fa[cl * D + dd]) * isc[half * 64 + dd];
                    if (kc[cl] <= -6.0e4f) val = 0.f;
                } else if (dd == D) {
                    is_const = true;
                    val = half ? 1.f : kc[cl];
                }
                const _Float16 h1 = (np == 1 && !is_const && half == 0) ? f16_toward_zero(val) : (_Float16)val;
                _Float16 hp;
                if (!is_const) hp = p ? (_Float16)(val - (float)h1) : h1;
                else if (half == 0) hp = p ? ((val <= -6.0e4f) ? (_Float16)0.f : (_Float16)(val - (float)h1)) : h1;    // k1 | k2
                else hp = (p != 0) == (np != 1) ? (_Float16)1.f : (_Float16)0.f;       // the threshold's slot: a1: 0, a2: 1 (one product: a1: 1, a2: 0)
                h[x] = __builtin_bit_cast(unsigned short, hp);
            }
            pf[e] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- split states: the on-pipe mixtures compacted to the front of the main layout
// derive_kernel (model_derive.hip) writes pm16f tile by tile in mixture order, an off-pipe mixture as a zero-weight row -- so the matrix-pipe
// kernels walked all M / 32 tiles of a state even when most of its rows were such (13.5 ms per batch beside the coarse pass at 57 %
// off-pipe mixtures).  For a state that HAS off-pipe mixtures this kernel rewrites the state's tiles from its on-pipe list (good_idx,
// ascending): row r = mixture good_idx[r], ceil(on-pipe / 32) tiles in use (npt[j], written by the prepass), the rest untouched and
// never read.  Same coefficients, scales, K0 and arithmetic as derive_kernel (one division per element, the per-mixture sums in its
// order).  The scoring kernel reads npt[j]; the accumulate consumer maps a row back through good_idx at its flush.  One workgroup per state.
template <int DMAX>
__global__ __launch_bounds__(256) void compact_main_kernel(const double *__restrict__ mean64, const double *__restrict__ var64,
                                                           const double *__restrict__ w64, const float *__restrict__ centers, int M, int Mpad,
                                                           int Mpad32, int D, int Dhost, int flags, int j0, const int *__restrict__ good_idx,
                                                           const int *__restrict__ nbad, const float *__restrict__ fscale,
                                                           const double *__restrict__ kzero, uint4 *__restrict__ pm16f) {
    const int j = j0 + blockIdx.x, tid = threadIdx.x;
    const int nb = nbad[j];
    if (nb == 0) return;                                         // not a split state: derive_kernel's tiles stand
    const int non = M - nb, ntl = (non + 31) / 32, nmt = Mpad32 / 32, KS8 = (D + 7) / 8;
    __shared__ float fa[32 * DMAX], fb[32 * DMAX], cen[DMAX], isc[2 * 64];
    __shared__ float kc[32];
    for (int d = tid; d < D; d += 256) cen[d] = centers[(size_t)j * D + d];
    for (int t = tid; t < 2 * KS8 * 8; t += 256) isc[(t / (KS8 * 8)) * 64 + t % (KS8 * 8)] = 1.0f / fscale[(size_t)j * 2 * (KS8 * 8) + t];
    __syncthreads();
    const double k0 = kzero[j];
    const int ml = tid >> 3, sub = tid & 7;
    const size_t srow = (size_t)j * Mpad;
    for (int t = 0; t < ntl; ++t) {
        const int idx = t * 32 + ml;
        const bool real_m = idx < non;
        const int m = real_m ? good_idx[srow + idx] : 0;
        const size_t jm = (srow + m) * D;
        double sumvar = 0.0, sumlog = 0.0, kq = 0.0;
        for (int dd = sub; dd < D; dd += 8) {
            float a = 0.f, b = 0.f;
            if (dd < Dhost && real_m) {
                const double v = var64[jm + dd], dm = mean64[jm + dd] - (double)cen[dd];
                const double hr = 0.5 / v;
                sumvar += v;
                if (flags & PCL_MODEL_LOGDET) sumlog += log(v);
                kq += dm * dm * hr;
                a = (float)(-LOG2E * hr);
                b = (float)(2.0 * LOG2E * dm * hr);
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
            float c = -6.0e4f;
            if (real_m) {
                const double tail = (flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;      // util.py:29 (quirk Q1)
                const double k2 = LOG2E * (log(w64[srow + m]) - 0.5 * Dhost * LOG_2PI - 0.5 * tail);
                const double kp = k2 - LOG2E * kq - k0;
                if (kp > -5.0e4) c = (float)kp;
            }
            kc[ml] = c;
        }
        __syncthreads();
        uint4 *pf = pm16f + ((size_t)j * nmt + t) * (2 * KS8 * 64);
        for (int e = tid; e < 2 * KS8 * 64; e += 256) {
            const int p = (e >> 6) / KS8, s = (e >> 6) % KS8, ln = e & 63, half = ln >> 5, cl = ln & 31;
            unsigned short h[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int dd = 8 * s + x;
                float val = 0.f;
                bool is_const = false;
                if (dd < D) {
                    val = (half ? fb[cl * D + dd] : fa[cl * D + dd]) * isc[half * 64 + dd];
                } else if (dd == D) {
                    is_const = true;
                    val = half ? 1.f : kc[cl];
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
        __syncthreads();
    }
}

template <int D, int NP>
void launch_coarse_p(pcl_ctx *ctx, pcl_batch *b, const CoarseExact &ex, unsigned long long *counters) {
    hipLaunchKernelGGL((gmm_score_coarse_kernel<D, PCL_COARSE_NT, NP>), dim3(b->n_tiles_c), dim3(WG), 0, ctx->stream, ctx->frames32,
                       reinterpret_cast<const uint4 *>(ctx->pmc), ctx->fscale_c, ctx->centers32, ctx->Mpad32 / 32, ctx->d_nct, b->d_tiles_c, b->d_segs,
                       b->Bt, b->d_tile_flags_c, ctx->kzero_c, ctx->kgap_c, ex, counters);
}
template <int D>
void launch_coarse_t(pcl_ctx *ctx, pcl_batch *b, const CoarseExact &ex, unsigned long long *counters) {
    if (ctx->coarse_np == 1) launch_coarse_p<D, 1>(ctx, b, ex, counters);
    else launch_coarse_p<D, 3>(ctx, b, ex, counters);
}

}  // namespace

bool pcl_coarse_enabled_for(const pcl_ctx *ctx, int D) {
    return ctx->coarse_on && ctx->score_variant == 7 && (D == 13 || D == 26 || D == 39 || D == 47);
}
bool pcl_coarse_enabled(const pcl_ctx *ctx) { return pcl_coarse_enabled_for(ctx, ctx->D); }

int pcl_coarse_tile_frames() { return WG / 64 * PCL_COARSE_NT * 32; }

void pcl_coarse_release(pcl_ctx *ctx) {
    dev_free(ctx->pmc);
    dev_free(ctx->fscale_c);
    dev_free(ctx->kzero_c);
    dev_free(ctx->kgap_c);
    dev_free(ctx->rows_c);
    dev_free(ctx->k2c);
    dev_free(ctx->d_nct);
    dev_free(ctx->d_coarse_counter);
    ctx->coarse_gen = -1;
}

// the coarse layout of the current model (derived on first use after the model changed)
int pcl_ensure_coarse(pcl_ctx *ctx) {
    if (ctx->coarse_gen == ctx->model_gen && ctx->pmc) return PCL_OK;
    const int KS8 = (ctx->D + 7) / 8, nmt = ctx->Mpad32 / 32;
    if (!ctx->pmc) {
        TRY(dev_alloc(ctx, &ctx->pmc, (size_t)ctx->J * nmt * (2 * KS8 * 64) * 8));      // unsigned short elements: 8 per uint4
        TRY(dev_alloc(ctx, &ctx->fscale_c, (size_t)ctx->J * 2 * KS8 * 8));
        TRY(dev_alloc(ctx, &ctx->kzero_c, (size_t)ctx->J));
        TRY(dev_alloc(ctx, &ctx->kgap_c, (size_t)ctx->J));
        TRY(dev_alloc(ctx, &ctx->rows_c, (size_t)ctx->J * ctx->Mpad * ctx->D * 2));
        TRY(dev_alloc(ctx, &ctx->k2c, (size_t)ctx->J * ctx->Mpad));
        TRY(dev_alloc(ctx, &ctx->d_nct, (size_t)ctx->J));
        TRY(dev_alloc(ctx, &ctx->d_coarse_counter, (size_t)2));              // [pairs evaluated in direct form, tiles given up]
        HIPCHK(ctx, hipMemsetAsync(ctx->d_coarse_counter, 0, 2 * sizeof(unsigned long long), ctx->stream));
    }
    pcl_timer_begin(ctx, "derive_coarse");
    if (ctx->D > 48) PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: coarse layout for D=%d", ctx->D);
    hipLaunchKernelGGL((coarse_derive_kernel<48>), dim3(ctx->J), dim3(256), 0, ctx->stream, ctx->mean64, ctx->var64, ctx->w64, ctx->centers32, ctx->M,
                       ctx->Mpad, ctx->Mpad32, ctx->D, ctx->Dhost, ctx->model_flags, ctx->d_bad_idx, ctx->d_nbad, reinterpret_cast<uint4 *>(ctx->pmc),
                       ctx->fscale_c, ctx->kzero_c, ctx->k2c, ctx->d_nct, ctx->coarse_np, ctx->kgap_c, reinterpret_cast<float2 *>(ctx->rows_c));
    pcl_timer_end(ctx, "derive_coarse");
    HIPCHK(ctx, hipGetLastError());
    ctx->coarse_gen = ctx->model_gen;
    return PCL_OK;
}

// the tight mixtures of the split states: coarse pass + exact evaluation of what it cannot rule out, log-added to the pipe's result
int pcl_launch_score_coarse(pcl_ctx *ctx, pcl_batch *b) {
    if (b->n_tiles_c == 0) return PCL_OK;
    TRY(pcl_ensure_coarse(ctx));
    const CoarseExact ex{reinterpret_cast<const float2 *>(ctx->rows_c), ctx->k2c, ctx->d_nbad, ctx->Mpad, ctx->Dhost};
    unsigned long long *counters = ctx->coarse_stats ? ctx->d_coarse_counter : nullptr;
    pcl_timer_begin(ctx, "score_coarse");
    switch (ctx->D) {
        case 47: launch_coarse_t<47>(ctx, b, ex, counters); break;
        case 39: launch_coarse_t<39>(ctx, b, ex, counters); break;
        case 26: launch_coarse_t<26>(ctx, b, ex, counters); break;
        case 13: launch_coarse_t<13>(ctx, b, ex, counters); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no coarse scoring kernel for D=%d", ctx->D);
    }
    pcl_timer_end(ctx, "score_coarse");
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_compact_main(pcl_ctx *ctx, int j_lo, int j_hi) {
    if (j_hi <= j_lo) return PCL_OK;
    if (!ctx->compact_main || ctx->D > 48) {                                                   // (the tile counts then cover the whole layout)
        std::vector<int> full((size_t)(j_hi - j_lo), ctx->Mpad32 / 32);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_npt + j_lo, full.data(), full.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return PCL_OK;
    }
    hipLaunchKernelGGL((compact_main_kernel<48>), dim3(j_hi - j_lo), dim3(256), 0, ctx->stream, ctx->mean64, ctx->var64, ctx->w64, ctx->centers32, ctx->M,
                       ctx->Mpad, ctx->Mpad32, ctx->D, ctx->Dhost, ctx->model_flags, j_lo, ctx->d_good_idx, ctx->d_nbad, ctx->fscale, ctx->kzero,
                       reinterpret_cast<uint4 *>(ctx->pm16f));
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_coarse_counters(pcl_ctx *ctx, unsigned long long *pairs, unsigned long long *tiles_given_up, int reset) {
    if (!ctx || !pairs) return PCL_ERR_INVALID;
    unsigned long long both[2] = {0, 0};
    *pairs = 0;
    if (tiles_given_up) *tiles_given_up = 0;
    if (!ctx->d_coarse_counter) return PCL_OK;
    HIPCHK(ctx, hipMemcpyAsync(both, ctx->d_coarse_counter, sizeof(both), hipMemcpyDeviceToHost, ctx->stream));
    if (reset) HIPCHK(ctx, hipMemsetAsync(ctx->d_coarse_counter, 0, sizeof(both), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *pairs = both[0];
    if (tiles_given_up) *tiles_given_up = both[1];
    return PCL_OK;
}

int pcl_coarse_counter(pcl_ctx *ctx, unsigned long long *pairs, int reset) { return pcl_coarse_counters(ctx, pairs, nullptr, reset); }

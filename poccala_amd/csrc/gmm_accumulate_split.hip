// gmm_accumulate_split.hip -- E-step sufficient statistics on the bf16 matrix pipe (gfx950), f32-class accuracy.
//
// Same reference rows as gmm_accumulate.hip (A13: Clustering.GMM.update_acc, StatisticalModel/Clustering.py:653-680,
// called from LHMM.update_acc, StatisticalModel/LHMM.py:497-505) and the same two chained contractions as its
// f32-input MFMA kernel:
//   (1) v[f,m] + cf[f] = Xe[f,:] . P[:,m]          K = 2D+2: features [x'^2_d | x'_d], the constant, and the per-frame
//                                                   coefficient cf = log2e (ln gamma_t(j) - ln b_j(o_t)) in the spare slot
//   (2) S[m,:]        += sum_f g[f,m] Xe[f,:]       g = exp2(v + cf) = gamma_t(j,m); columns [x'^2_d, x'_d | 1] -> S2, S1, S0
// but on the bf16 pipe, which is 16x faster than the f32-input MFMA and leaves the VALU alone (gmm_score_split.hip):
//   (1) every operand is the exact sum of three bf16 pieces, six cross products kept (error below the f32 chain's);
//   (2) the posteriors are written as TWO bf16 pieces g1 + g2 (16 significand bits) and the features as three:
//       products g1x1 g1x2 g1x3 g2x1 g2x2.  What is lost is g3 and g2x3: the statistics are the EXACT moments of the
//       frame under posteriors perturbed by < 2^-16 relative -- the same perturbation on S0, S1 and S2, so the
//       cancellation in cov = S2 - 2 d S1 + d^2 S0 does not amplify it.  bf16 keeps f32's exponent range, so a
//       rarely responsible mixture (g ~ 1e-30) keeps its relative accuracy, which f16 pieces would not.
// Orientation as in the f32 kernel: product (1) is D1[frame rows][mixture cols], so a lane holds, for ITS mixture,
// 16 frame rows; registers 8s..8s+7 converted to bf16 are the A fragment of k-step s of product (2) (accumulator as
// operand: element j of lane half h is frame 16s + 8(j>>2) + 4h + (j&3)), no data movement between the products.
// The 32-frame tile is staged in LDS twice, in the two fragment layouts: frame-major for (1), feature-major in that
// permuted frame order for (2); both are read with conflict-free ds_read_b128.
// A wave owns one 32-mixture tile of one state: its 60 parameter registers and 48 moment accumulators stay resident
// while the workgroup (8 waves) walks the state's list of surviving frames.
#include "pcl_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
#ifndef PCL_ACCS_AW
#define PCL_ACCS_AW 8
#endif
constexpr int AW = PCL_ACCS_AW;   // waves (32-mixture tiles) per workgroup
#ifndef PCL_ACCS_P2_INTERLEAVE
#define PCL_ACCS_P2_INTERLEAVE 1
#endif
#ifndef PCL_ACCS_P1_SPLIT
#define PCL_ACCS_P1_SPLIT 0
#endif

__device__ __forceinline__ unsigned short bf16_bits(float x) {      // round to nearest even (finite inputs)
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned int)b << 16); }
__device__ __forceinline__ void split3_bits(float x, unsigned short &p1, unsigned short &p2, unsigned short &p3) {
    p1 = bf16_bits(x);
    float r = x - bf16_val(p1);
    p2 = bf16_bits(r);
    r -= bf16_val(p2);
    p3 = bf16_bits(r);
}

// LDS fragment blocks: 64 chunks of 16 B read by the 64 lanes of one ds_read_b128 (conflict free).  Blocks are 65
// chunks apart, not 64: the staging threads' lanes run along the feature index, i.e. ACROSS blocks at equal chunk
// positions, and with 1 KiB between blocks they would all hit the same banks.
constexpr int BS = 65;

template <int D>
__global__ __launch_bounds__(AW * 64, 2) void gmm_accumulate_split_kernel(   // 2 waves per SIMD: one workgroup of 8 waves or two of 4
   
    const float *__restrict__ frames, const uint4 *__restrict__ pm16, const float *__restrict__ centers,
    const double *__restrict__ means64, int M, int Mpad, int n_mtiles, int n_states, const int *__restrict__ work_states,
    const int *__restrict__ seg_lo, const int *__restrict__ seg_hi, const long long *__restrict__ off,
    const ActiveFrame *__restrict__ list, double bias, double *__restrict__ st_acc, double *__restrict__ st_alpha,
    double *__restrict__ st_mean, double *__restrict__ st_cov) {
    constexpr int KS8 = (D + 8) / 8;                 // K-steps of 16 of product (1): D features per side + the constant slot
    constexpr int SC = D / 8, JC = D % 8;            // where the constant slot sits
    constexpr int NCT = (2 * D + 1 + 31) / 32;       // 32-column tiles of product (2): 2D feature columns + the constant
    constexpr int L1 = 3 * KS8 * BS;                 // uint4 per buffer, layout [piece][s][lane]   (frame-major fragments)
    constexpr int L2 = 3 * NCT * 2 * BS;             // uint4 per buffer, layout [piece][ct][s'][lane] (feature-major fragments)
    __shared__ __attribute__((aligned(16))) uint4 xe[2][L1 + L2];

    // XCD-aware mapping: the 8 slices of one state sit on block indices with equal residue mod 8
    const int nslice = (n_mtiles + AW - 1) / AW;
    const int b = blockIdx.x;
    int w, slice;
    {   // block b runs on XCD b % 8: states are dealt to the XCDs in groups of 8, all slices of a state on one XCD
        w = (b & 7) + 8 * (b / (8 * nslice));
        slice = (b >> 3) % nslice;
    }
    if (w >= n_states) return;
    const int j = work_states[w];
    const long long beg = off[seg_lo[w]], end = off[seg_hi[w]];
    if (beg == end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int mt = slice * AW + wave;
    const bool live = mt < n_mtiles;
#ifdef PCL_ACCS_PRIO
    if (wave < AW / 2) __builtin_amdgcn_s_setprio(PCL_ACCS_PRIO);
#endif
    const float *cen = centers + (size_t)j * D;

    // parameters of this wave's m-tile: the B operand of product (1) is the scoring layout as it is
    bf8v pf[3][KS8];
    {
        const uint4 *pq = pm16 + ((size_t)j * n_mtiles + (live ? mt : 0)) * (3 * KS8 * 64) + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const uint4 t = pq[(p * KS8 + s) * 64];
                pf[p][s] = __builtin_bit_cast(bf8v, t);
            }
    }
    f16v S[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[ct][r] = 0.f;
    double galpha = 0.0;
    constexpr double LOG2E = 1.4426950408889634074;

    // LDS image: everything that does not depend on the frame is written once (zero padding, the constant 1 of
    // product (1)'s slot d = D on the x'^2 side, column 2D = 1 of product (2))
    {
        unsigned short *h16 = reinterpret_cast<unsigned short *>(&xe[0][0]);
        for (int i = threadIdx.x; i < 2 * (L1 + L2); i += AW * 64) xe[0][i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * 32; i += AW * 64) {
            const int bufi = i >> 5, f = i & 31;
            unsigned short *base = h16 + (size_t)bufi * (L1 + L2) * 8;
            base[((0 * KS8 + SC) * BS + f) * 8 + JC] = 0x3f80;                              // (1): piece 1 of 1.0, lane = frame, side 0
            constexpr int c2 = 2 * D, ct = c2 >> 5, c = c2 & 31;
            const int sp = f >> 4, fp = f & 15, hh = (fp >> 2) & 1, jj = ((fp >> 3) << 2) | (fp & 3);
            base[(size_t)L1 * 8 + ((((0 * NCT + ct) * 2 + sp) * BS + hh * 32 + c) * 8 + jj)] = 0x3f80;   // (2): column 2D = 1
        }
    }

    // Gather pipeline (issue early / write late), as in the f32 kernel: the loads of tile i+1 are issued before the
    // MFMAs of tile i and written to the other LDS buffer between and after them.  A work item is a 2 x 2 block
    // (frames 2 fp, 2 fp + 1 x dimensions 2 dp, 2 dp + 1): v_cvt_pk_bf16_f32 of a (d, d+1) pair IS the dword of the
    // frame-major layout, and one v_perm of two such dwords is the (f, f+1) dword of the feature-major layout, so a
    // block costs ~105 instructions instead of 4 x 62 with per-element splits and two-byte writes (which made
    // staging 40 % of the kernel: all 8 waves issue it, and instruction issue, not the matrix pipe, set the time).
    // (A variant with one thread per 16-byte fragment group needs gathers across 8 frame rows and measured slower.)
    constexpr int NDP = (D + 1) / 2, NITEM = 16 * NDP, NPT = (NITEM + AW * 64 - 1) / (AW * 64);   // blocks per thread
    int item_fp[NPT], d0[NPT];
    bool item_on[NPT], pair_ok[NPT];
    float cen0[NPT], cen1[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int item = threadIdx.x + k * AW * 64;
        item_on[k] = item < NITEM;
        item_fp[k] = item / NDP;                                               // lanes run along d: coalesced row reads
        d0[k] = 2 * (item % NDP);
        pair_ok[k] = d0[k] + 1 < D;                                            // D odd: the last block has one dimension
        cen0[k] = item_on[k] ? cen[d0[k]] : 0.f;
        cen1[k] = (item_on[k] && pair_ok[k]) ? cen[d0[k] + 1] : 0.f;
    }
    // three stages in flight: rows of tile i+1 (xv1, stored during tile i), rows of tile i+2 (xv2, loading during tile
    // i), list entries of tile i+3 (ix3): a dependent pair of loads (entry -> frame row) under load takes longer than one
    // tile of MFMAs
    float xv1[NPT][2][2], xv2[NPT][2][2];
    int vm1[NPT], vm2[NPT];
    int ix2[NPT][2], ix3[NPT][2];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        vm1[k] = vm2[k] = 0;
        ix2[k][0] = ix2[k][1] = ix3[k][0] = ix3[k][1] = -1;
    }
    double cf1 = -INFINITY, cf2 = -INFINITY, cf3 = -INFINITY;     // ln gamma - ln b, raw (scaled and clamped when stored)
    double lg1 = 0.0, lg2 = 0.0, lg3 = 0.0;
    auto load_index = [&](long long f0, int (&ix)[NPT][2], double &cf, double &lg) {
        const int nf = (f0 < end) ? (int)min(32LL, end - f0) : 0;
#pragma unroll
        for (int k = 0; k < NPT; ++k)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int f = 2 * item_fp[k] + a;
                ix[k][a] = (item_on[k] && f < nf) ? (int)list[f0 + f].frame : -1;
            }
        cf = -INFINITY;                                    // padding frame: g = 0
        lg = 0.0;
        if ((int)threadIdx.x < nf) {
            const ActiveFrame a = list[f0 + threadIdx.x];
            cf = a.coef;
            lg = a.lg;
        }
    };
    // nothing here may consume a loaded value (no centring, no select on the data): the compiler would wait for the
    // load on the spot and the whole gather latency would sit at the top of every tile (measured with in-kernel
    // stamps: 2460 of 8200 cycles per tile).  Padding rows read frame 0 and are masked when they are stored.
    auto load_rows = [&](const int (&ix)[NPT][2], float (&x)[NPT][2][2], int (&vmask)[NPT]) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            vmask[k] = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float *row = frames + (long long)max(ix[k][a], 0) * D + d0[k];
                x[k][a][0] = row[0];
                x[k][a][1] = row[pair_ok[k] ? 1 : 0];
                vmask[k] |= (ix[k][a] >= 0) << a;
            }
        }
    };
    // one chunk = one side (x'^2 | x') of one block: 6 conversions, 12 four-byte LDS writes
    auto store_chunk = [&](int buf, int chunk, const float (&xall)[NPT][2][2], const int (&vmall)[NPT], bool force = false) {
#ifdef PCL_ACCS_DIAG_NOSTORE
        if (!force) return;                               // diagnostic: only the prologue stages (both buffers), timing only
#endif
        const int k = chunk >> 1, side = chunk & 1;
        if (!item_on[k]) return;
        const float (&xraw)[2][2] = xall[k];
        const int vmask = vmall[k];
        unsigned int *w1 = reinterpret_cast<unsigned int *>(&xe[buf][0]);
        unsigned int *w2 = w1 + (size_t)L1 * 4;
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        unsigned int P[3][2];                              // [piece][frame]: dword = pieces of (d0, d0 + 1)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const bool ok = (vmask >> a) & 1;
            const float x0 = ok ? xraw[a][0] - cen0[k] : 0.f, x1 = (ok && pair_ok[k]) ? xraw[a][1] - cen1[k] : 0.f;
            float r0 = side ? x0 : x0 * x0, r1 = side ? x1 : x1 * x1;
#ifndef PCL_ACCS_TRUNC_SPLIT       // the truncating variant below measured the same (47.9 vs 48.0 ms)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const bf2v c = bf2v{(__bf16)r0, (__bf16)r1};                       // v_cvt_pk_bf16_f32, round to nearest even
                const unsigned int u = __builtin_bit_cast(unsigned int, c);
                P[p][a] = u;
                r0 -= __uint_as_float(u << 16);
                r1 -= __uint_as_float(u & 0xffff0000u);
            }
#else
            // truncating split: a piece is the top 16 bits of what is left (8 significand bits each, 24 in all: still an
            // exact decomposition); one v_perm packs the (d, d + 1) pair, and there is no conversion to undo
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned int u0 = __float_as_uint(r0), u1 = __float_as_uint(r1);
                P[p][a] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);              // (hi16 of r0) | (hi16 of r1) << 16
                if (p < 2) {
                    r0 -= __uint_as_float(u0 & 0xffff0000u);
                    r1 -= __uint_as_float(u1 & 0xffff0000u);
                }
            }
#endif
        }
        const int s = d0[k] >> 3, jd = d0[k] & 7;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int f = 2 * item_fp[k] + a;
            const int o = ((0 * KS8 + s) * BS + side * 32 + f) * 4 + (jd >> 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (pair_ok[k]) w1[o + p * (KS8 * BS * 4)] = P[p][a];
                else reinterpret_cast<unsigned short *>(w1)[2 * (o + p * (KS8 * BS * 4))] = (unsigned short)P[p][a];   // slot d = D is not ours
            }
        }
        // feature-major: dword = the same piece of frames (2 fp, 2 fp + 1) for one column
        const int f0 = 2 * item_fp[k], sp = f0 >> 4, fq = f0 & 15, hh = (fq >> 2) & 1, jj = ((fq >> 3) << 2) | (fq & 3);
#pragma unroll
        for (int bdim = 0; bdim < 2; ++bdim) {
            if (bdim == 1 && !pair_ok[k]) break;
            const int c2 = 2 * (d0[k] + bdim) + side, ct = c2 >> 5, c = c2 & 31;
            const int o = (((0 * NCT + ct) * 2 + sp) * BS + hh * 32 + c) * 4 + (jj >> 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned int q = bdim ? __builtin_amdgcn_perm(P[p][1], P[p][0], 0x07060302u)      // hi halves: (f0 | f0+1 << 16)
                                            : __builtin_amdgcn_perm(P[p][1], P[p][0], 0x05040100u);     // lo halves
                w2[o + p * (NCT * 2 * BS * 4)] = q;
            }
        }
    };
    auto store_cf = [&](int buf, double cfv, double lgv) {
        if (threadIdx.x < 32) {
            unsigned short *h1 = reinterpret_cast<unsigned short *>(&xe[buf][0]);
            const int f = threadIdx.x;
            if (slice == 0) galpha += lgv;                    // gamma_t(j) itself (the list holds exp(ln gamma))
            unsigned short c1, c2p, c3;
            const float cfl2 = __builtin_fmaxf((float)(cfv * LOG2E), -3.0e38f);   // finite: its pieces meet zeros of the other operand
            split3_bits(cfl2, c1, c2p, c3);
            if (cfl2 < -1.0e37f) c2p = c3 = 0;
            h1[((0 * KS8 + SC) * BS + 32 + f) * 8 + JC] = c1;               // slot d = D on the x' side carries cf
            h1[((1 * KS8 + SC) * BS + 32 + f) * 8 + JC] = c2p;
            h1[((2 * KS8 + SC) * BS + 32 + f) * 8 + JC] = c3;
        }
    };
    __syncthreads();
    load_index(beg, ix2, cf2, lg2);
    load_rows(ix2, xv1, vm1);
#pragma unroll
    for (int ch = 0; ch < 2 * NPT; ++ch) store_chunk(0, ch, xv1, vm1, true);
    store_cf(0, cf2, lg2);
#ifdef PCL_ACCS_DIAG_NOSTORE
#pragma unroll
    for (int ch = 0; ch < 2 * NPT; ++ch) store_chunk(1, ch, xv1, vm1, true);
#endif
    load_index(beg + 32, ix2, cf1, lg1);
    load_rows(ix2, xv1, vm1);                            // tile 1
    load_index(beg + 64, ix2, cf2, lg2);                 // tile 2
    __syncthreads();
    // one tile; the register sets alternate between calls (no moves: a move would wait for the loads just issued)
    //   xs/cfs/lgs: rows of tile i+1 to store      xl: rows of tile i+2 to load, from the entries ixs
    //   ixl/cfl/lgl: list entries of tile i+3 to load
#ifdef PCL_ACCS_STAMPS
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned int stamp_n = 0;
#define STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += t_ - stamp_t; stamp_t = t_; }
#else
#define STAMP(k)
#endif
    auto tile_step = [&](long long f0, int buf, const float (&xs)[NPT][2][2], const int (&vms)[NPT], double cfs, double lgs,
                         float (&xl)[NPT][2][2], int (&vml)[NPT], const int (&ixs)[NPT][2], int (&ixl)[NPT][2], double &cfl,
                         double &lgl) {
        const bool more = f0 + 32 < end;
#ifdef PCL_ACCS_STAMPS
        unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
        ++stamp_n;
#endif
        load_index(f0 + 96, ixl, cfl, lgl);              // tile i+3: list entries (issued first: vmcnt retires in order, and
        load_rows(ixs, xl, vml);                         // tile i+2's rows must not be waited for before the next step)
        __builtin_amdgcn_sched_barrier(0);               // keep the loads in front of the matrix work
        STAMP(0)
        // The staging of tile i+1 is cut into its two sides and pinned between the MFMA groups of tile i: all 8 waves
        // run the same phase at the same time (workgroup barrier per tile), so anything left after the last MFMA is time
        // the matrix pipe idles.
        int chunk = 0;
        auto emit = [&](int upto) {
            __builtin_amdgcn_sched_barrier(0);
            for (; chunk < upto * NPT && chunk < 2 * NPT; ++chunk)
                if (more) store_chunk(buf ^ 1, chunk, xs, vms);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (live) {
            const uint4 *x1 = &xe[buf][0], *x2 = &xe[buf][L1];
            // (1) D1[frame][mixture] = Xe . P (log2 domain, + cf): small cross terms first
            f16v d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[r] = 0.f;
#if PCL_ACCS_P1_SPLIT
            f16v d1b;                                    // second chain: the small cross terms
#pragma unroll
            for (int r = 0; r < 16; ++r) d1b[r] = 0.f;
#endif
            auto pass = [&](int px, int pp) {
#pragma unroll
                for (int s = 0; s < KS8; ++s) {
                    const bf8v a = __builtin_bit_cast(bf8v, x1[(px * KS8 + s) * BS + lane]);
#if PCL_ACCS_P1_SPLIT
                    if ((s & 1) == 0) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[pp][s], d1, 0, 0, 0);
                    else d1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[pp][s], d1b, 0, 0, 0);
#else
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[pp][s], d1, 0, 0, 0);
#endif
                }
            };
#ifndef PCL_ACCS_DIAG_NOP1
            pass(2, 0);
            pass(1, 1);
            pass(1, 0);
            pass(0, 2);
            emit(1);
            pass(0, 1);
#endif
            pass(0, 0);
#if PCL_ACCS_P1_SPLIT
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[r] += d1b[r];
#endif
            STAMP(1)
            // posteriors gamma_t(j,m) (Clustering.py:660-661) in two bf16 pieces = the A fragments of product (2)
            bf8v g1[2], g2[2];
            {
                typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
                unsigned int u1[8], u2[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float ga = __builtin_amdgcn_exp2f(d1[2 * r]), gb = __builtin_amdgcn_exp2f(d1[2 * r + 1]);
                    const bf2v c = bf2v{(__bf16)ga, (__bf16)gb};                   // v_cvt_pk_bf16_f32
                    u1[r] = __builtin_bit_cast(unsigned int, c);
                    const bf2v e = bf2v{(__bf16)(ga - __uint_as_float(u1[r] << 16)), (__bf16)(gb - __uint_as_float(u1[r] & 0xffff0000u))};
                    u2[r] = __builtin_bit_cast(unsigned int, e);
                }
                g1[0] = __builtin_bit_cast(bf8v, make_uint4(u1[0], u1[1], u1[2], u1[3]));
                g1[1] = __builtin_bit_cast(bf8v, make_uint4(u1[4], u1[5], u1[6], u1[7]));
                g2[0] = __builtin_bit_cast(bf8v, make_uint4(u2[0], u2[1], u2[2], u2[3]));
                g2[1] = __builtin_bit_cast(bf8v, make_uint4(u2[4], u2[5], u2[6], u2[7]));
            }
            STAMP(2)
            // (2) S[mixture][feature] += g^T . Xe
#if PCL_ACCS_P2_INTERLEAVE
            // the NCT column tiles are independent accumulators: issue them round robin so that a dependent MFMA is
            // NCT instructions behind its producer (a single chain runs at half the pipe rate)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf8v bq[3][NCT];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) bq[p][ct] = __builtin_bit_cast(bf8v, x2[((p * NCT + ct) * 2 + sp) * BS + lane]);
#ifndef PCL_ACCS_DIAG_NOP2
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], bq[2][ct], S[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[sp], bq[1][ct], S[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[sp], bq[0][ct], S[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], bq[1][ct], S[ct], 0, 0, 0);
#endif
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], bq[0][ct], S[ct], 0, 0, 0);
                if (sp == 0) emit(2);
            }
#else
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf8v b1 = __builtin_bit_cast(bf8v, x2[((0 * NCT + ct) * 2 + sp) * BS + lane]);
                    const bf8v b2 = __builtin_bit_cast(bf8v, x2[((1 * NCT + ct) * 2 + sp) * BS + lane]);
                    const bf8v b3 = __builtin_bit_cast(bf8v, x2[((2 * NCT + ct) * 2 + sp) * BS + lane]);
#ifndef PCL_ACCS_DIAG_NOP2
                    S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], b3, S[ct], 0, 0, 0);
                    S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[sp], b2, S[ct], 0, 0, 0);
                    S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[sp], b1, S[ct], 0, 0, 0);
                    S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], b2, S[ct], 0, 0, 0);
#endif
                    S[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[sp], b1, S[ct], 0, 0, 0);
                    if (ct == 0 && sp == 1) emit(2);
                }
#endif
        }
        STAMP(3)
        emit(2);                                         // whatever is left (and everything on a wave without an m-tile)
        if (more) store_cf(buf ^ 1, cfs, lgs);           // nobody reads xe[buf^1] until the barrier below
        STAMP(4)
#ifndef PCL_ACCS_DIAG_NOBARRIER
        __syncthreads();
#endif
        STAMP(5)
    };
    // state before tile i: xv1 / cf1 = tile i+1, ix2 / cf2 = entries of tile i+2.  A step loads tile i+2's rows into the
    // other row set and tile i+3's entries into the other index set; only the two scalars per thread are moved.
    for (long long f0 = beg; f0 < end; f0 += 64) {
        tile_step(f0, 0, xv1, vm1, cf1, lg1, xv2, vm2, ix2, ix3, cf3, lg3);
        cf1 = cf2; lg1 = lg2; cf2 = cf3; lg2 = lg3;
        if (f0 + 32 >= end) break;
        tile_step(f0 + 32, 1, xv2, vm2, cf1, lg1, xv1, vm1, ix3, ix2, cf3, lg3);
        cf1 = cf2; lg1 = lg2; cf2 = cf3; lg2 = lg3;
    }

#ifdef PCL_ACCS_STAMPS
    if (blockIdx.x == 64 && lane == 0 && stamp_n)
        printf("wave %d tiles %u: loads-issue %llu  P1(+chunk) %llu  exp/cvt %llu  P2(+chunk) %llu  tail stores %llu  barrier %llu  (memtime ticks per tile)\n",
               wave, stamp_n, stamp_acc[0] / stamp_n, stamp_acc[1] / stamp_n, stamp_acc[2] / stamp_n, stamp_acc[3] / stamp_n,
               stamp_acc[4] / stamp_n, stamp_acc[5] / stamp_n);
#endif
    // ---- flush: lane = feature column, register = mixture row; cov = S2 - 2 d S1 + d^2 S0, mean = S1 + (c + bias) S0
    if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float s0 = __shfl(S[(2 * D) >> 5][r], (lane & 32) + ((2 * D) & 31), 64);   // column 2D = the constant feature
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float s1 = __shfl_xor(S[ct][r], 1, 64);                              // odd neighbour: x' column of the same d
                const int cidx = ct * 32 + col;
                if (m < M && !(cidx & 1) && cidx < 2 * D) {
                    const int d = cidx >> 1;
                    const size_t o = ((size_t)j * Mpad + m) * D + d;
                    const double c = (double)cen[d], dl = means64[o] - c;
                    const double S0 = (double)s0, S1 = (double)s1, S2 = (double)S[ct][r];
                    st_mean[o] += S1 + (c + bias) * S0;                            // Clustering.py:669-672
                    st_cov[o] += S2 - 2.0 * dl * S1 + dl * dl * S0;                // Clustering.py:674-678
                }
                if (m < M && cidx == 2 * D) st_acc[(size_t)j * Mpad + m] += (double)S[ct][r];   // Clustering.py:665
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

}  // namespace

// states [0, ns) of the batch's accumulate order (the well-conditioned ones)
int pcl_launch_accumulate_split(pcl_ctx *ctx, pcl_batch *b, int ns) {
    if (ns == 0) return PCL_OK;
    const int nmt = ctx->Mpad32 / 32, nslice = (nmt + AW - 1) / AW;
    const int nblocks = ((ns + 7) / 8) * 8 * nslice;
#define LAUNCH_SPLIT(DD)                                                                                                  \
    hipLaunchKernelGGL((gmm_accumulate_split_kernel<DD>), dim3(nblocks), dim3(AW * 64), 0, ctx->stream, ctx->frames32,   \
                       reinterpret_cast<const uint4 *>(ctx->pm16), ctx->centers32, ctx->mean64, ctx->M, ctx->Mpad, nmt, ns, \
                       b->d_work_states, b->d_seg_lo, b->d_seg_hi, b->acc_off, b->acc_list, 100.0, ctx->st_acc, ctx->st_alpha, \
                       ctx->st_mean, ctx->st_cov)
    switch (ctx->D) {
        case 39: LAUNCH_SPLIT(39); break;
        case 26: LAUNCH_SPLIT(26); break;
        case 13: LAUNCH_SPLIT(13); break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "internal: no split accumulate kernel for D=%d", ctx->D);
    }
#undef LAUNCH_SPLIT
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// hmm_decode_lr.hip -- the token-passing decoder for LEFT-TO-RIGHT units (gfx950): same rules, same bits as
// hmm_decode_kernel (hmm_decode.hip; SURVEY.md 8(a) A16 / 8(f) rank 3, Decoder.py:91-167,250-288), one LANE per token.
//
// Every unit HMM the reference builds is left to right: the entry state reaches state 1 only, an emitting state itself and
// its successor only (AcousticModel.py:176-181; the transition M-step keeps the zeros, LHMM.py:509-524).  Then a token's
// recursion  p_j = max_i(p_i + ln A_ij) + B_j  (Decoder.py:278-283) has two finite terms per state, the entry state is
// -inf from its second step on and the exit state always, so a token is SIX float64 values and a step is a short serial
// chain one lane can run: no cross-lane traffic at all (the general kernel spends half its step in ds_bpermute and
// LDS reads of the 8 x 8 predecessor table), and every array pass is one access per field, 4 or 8 bytes per lane (round 5: the int4
// meta record became three int arrays, so a 64-byte line serves 16 tokens' node / history / unit pair):
//     score [2][cap] f64 | p [2][6][cap] f64 (state-major) | node, hist, upair [2][cap] i32 | src [2][cap]
// Two buffers of each, ONE parity: a frame reads a token's state where the previous frame left it (through its src entry) and writes
// it at the token's dense index into the other buffers; the end of a frame compacts only the index list.
// Terms the general kernel adds with ln A = -inf are -inf there and absent here; max and + are exact, so the bits agree.
//
// The kernel is bound by the LATENCY of dependent memory round trips (a frame is a chain of phases, 300 frames a chain of
// frames; all 417 utterances of a C5 shard are resident at once, so there is nothing else to switch to): every phase issues
// all the loads of a lane's tokens before it uses any.  A frame, for the workgroup of one utterance (wave w owns tokens
// [w C, (w+1) C), lane l its tokens w C + 64 k + l, so an ORDERED prefix over the tokens is a ballot per 64 tokens plus one
// exchange of wave totals):
//   A   every live token steps: src of all rows, then per group of 2 rows the old p / score / node / history / unit pair (gathered
//       through src: an ascending sequence with gaps, nearly coalesced), step, dense stores of all of them.  Finished
//       tokens = donors: appended, in token order, to the wave's donor list in LDS (token, node, history, score) -> barrier
//   A2  a lane per donor: the node's children (one gather), child offsets inside the wave, the wave's best / first
//       word-end donor                                                                                           -> barrier
//   C   the frame's hand-overs are ONE flattened list of (donor, target) pairs in the order the restated rules create tokens;
//       four pairs per thread, found by a search over the wave totals and the wave's list (all in LDS): a live target takes
//       the score if strictly better (Decoder.py:126-134), otherwise a new token is made behind the old ones and takes its
//       first step at once (:135-140), by the same thread                                                        -> 2 barriers per 2048 pairs
//   E   pruning (Decoder.py:159-167) over the old unfinished tokens, keys in registers: 12-bit radix digit below the keys'
//       common prefix, then the few keys of the selected bin ranked directly in LDS                               -> 6 barriers
//   F   stable compaction of the index list (src); nothing else moves, and the node -> token map is stamped per frame by
//       the step itself (round 5: entries of another frame are stale and read as "no token")                                   -> 2 barriers
#include <math.h>
#include <stdio.h>

#include "hmm_decode_args.h"

namespace {

// 512 threads at 4 waves per SIMD: two workgroups share a CU, so the 417 utterances of a C5 shard run in ONE round of the 256 CUs
// (1024 threads: two rounds, 63 ms against 51; groups of 4 rows spill and lose 9 ms: profiles/r03_decode_c5.txt)
#ifndef PCL_DECLR_DW
#define PCL_DECLR_DW 512
#define PCL_DECLR_WAVES 4
#endif
constexpr int LW = PCL_DECLR_DW;    // threads per workgroup
constexpr int LNW = LW / 64;
constexpr int E = 3, NE = 2 * E;    // emitting states per unit (S = 5) and per token at most
constexpr int TS = 7;               // doubles per unit in LDS: ln A self[3] | next[3] (k -> k + 1, the last one leaves the unit) | entry -> 1
                                    // (an odd stride: units spread over all banks; 8 gave 16-way conflicts)
constexpr int HB = 12, HBINS = 1 << HB, BPT = HBINS / LW;    // radix digit of the pruning select
constexpr int CAND = 1024;          // keys of the selected bin ranked directly
constexpr int DLW = 128;            // donors of a wave whose list entries live in LDS (more: in the utterance's seg_* arrays in HBM)
constexpr int NONE = 0x7fffffff;
constexpr int SLOT_BITS = 14;       // node -> token map entry = (frame << SLOT_BITS) | dense token index (cap <= 16 LW = 8192 < 2^14), -1 = none
constexpr int FRESH = (int)0x80000000;   // src: the token's last step was its first (its entry state still holds ln pi)
#ifndef PCL_DECLR_G
#define PCL_DECLR_G 2
#endif
#ifndef PCL_DECLR_PMAX
#define PCL_DECLR_PMAX 4
#endif
constexpr int G = PCL_DECLR_G;      // rows of a lane whose old p / meta / score are in flight together
constexpr int PMAX = PCL_DECLR_PMAX; // pairs per thread in flight together
constexpr unsigned long long NOKEY = ~0ull;
constexpr size_t LIST_BYTES = (size_t)LNW * DLW * 20, SEL_BYTES = (size_t)HBINS * 4 + (size_t)CAND * 8;
constexpr size_t POOL_BYTES = LIST_BYTES > SEL_BYTES ? LIST_BYTES : SEL_BYTES;

// One step of one token by one lane.  first: p = ln pi + B[:,t] (Decoder.py:270); else the max recursion (:278-283) in the
// general kernel's operand order (predecessor i ascending: the state before, then the state itself).  best = max_j p_j,
// fin (D1) <=> the FIRST argmax is the last emitting state.
__device__ __forceinline__ void lr_step(const double *tab, const double *Bs, int up, bool first, bool fresh, double lpi1, double lpi2,
                                        const double (&po)[NE], double (&pn)[NE], double &best, int &fin) {
    const int u0 = up & 0xffff, u1r = (int)((unsigned int)up >> 16);
    const bool two = u1r != 0xffff;
    const int u1 = two ? u1r : u0;
    const double *t0 = tab + u0 * TS, *t1 = tab + u1 * TS;
    const double *b0 = Bs + 1 + u0 * E, *b1 = Bs + 1 + u1 * E;
    const double lpi = two ? lpi2 : lpi1;
    const double ninf = -INFINITY;
    if (first) {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            pn[k] = lpi + b0[k];
            pn[E + k] = two ? lpi + b1[k] : ninf;
        }
        best = lpi + 0.0;                                          // the entry VirtualState scores ln 1 (AcousticModel.py:218)
    } else {
        const double e_old = fresh ? lpi + 0.0 : ninf;
        pn[0] = fmax(e_old + t0[6], po[0] + t0[0]) + b0[0];
        pn[1] = fmax(po[0] + t0[3], po[1] + t0[1]) + b0[1];
        pn[2] = fmax(po[1] + t0[4], po[2] + t0[2]) + b0[2];
        const double q3 = fmax(po[2] + t0[5], po[3] + t1[0]) + b1[0];
        const double q4 = fmax(po[3] + t1[3], po[4] + t1[1]) + b1[1];
        const double q5 = fmax(po[4] + t1[4], po[5] + t1[2]) + b1[2];
        pn[3] = two ? q3 : ninf;
        pn[4] = two ? q4 : ninf;
        pn[5] = two ? q5 : ninf;
        best = ninf;                                               // (the entry state is -inf from the second step on)
    }
    int arg = 0;
#pragma unroll
    for (int j = 0; j < NE; ++j)
        if (pn[j] > best) {                                        // strictly greater: the first argmax (Decoder.py:263-268)
            best = pn[j];
            arg = j + 1;
        }
    fin = arg == (two ? NE : E);
}

#ifdef PCL_DECLR_WAVES
#define PCL_DECLR_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PCL_DECLR_WAVES, PCL_DECLR_WAVES)))
#else
#define PCL_DECLR_WAVES_ATTR
#endif
template <int KMAX>
__global__ __launch_bounds__(LW) PCL_DECLR_WAVES_ATTR void hmm_decode_lr_kernel(DecArgs a, int NbP) {
    extern __shared__ double dyn[];                                // unit table [n_units][TS] | two emission rows [NbP]
    // one pool, two tenants: the donor lists (phases A - C) and the pruning select's histogram + candidates (phase E)
    __shared__ __attribute__((aligned(16))) unsigned char pool[POOL_BYTES];
    double *L_s = (double *)pool;                                  // [LNW][DLW] donor score
    int *L_a = (int *)(pool + (size_t)LNW * DLW * 8);              // token index, then the child offset inside the wave
    int *L_b = L_a + LNW * DLW;                                    // node, then its first child
    int *L_h = L_b + LNW * DLW;                                    // history
    unsigned int *hist = (unsigned int *)pool;                     // [HBINS]
    unsigned long long *cand = (unsigned long long *)(pool + (size_t)HBINS * 4);   // [CAND]
    __shared__ unsigned int occ[256];
    __shared__ int wd_cnt[LNW], wd_ch[LNW], wd_fw[LNW], wd_fwend[LNW], wd_bwi[LNW], wd_bwnode[LNW], wd_bwhist[LNW];
    __shared__ double wd_bw[LNW];
    __shared__ int wp[PMAX][LNW], we_cnt[LNW], ws_scan[LNW], w_eq[LNW], wf_keep[LNW];
    __shared__ unsigned long long red_u[2][LNW];
    __shared__ int red_i[LNW];
    __shared__ double red_d[LNW];
    __shared__ unsigned long long s_key;
    __shared__ int s_sel, s_rank, s_cnt, s_cn;
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef PCL_DECLR_NOPRIO
    // latency bound, few instructions, 300 frames of dependent phases: beside the scoring kernel of the next chunk (the C5 pipeline)
    // its waves issue first -- it costs the matrix-pipe-bound scoring waves little and keeps a workgroup's barrier phases short
    __builtin_amdgcn_s_setprio(3);
#endif
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const UttDesc d = a.utts[u];
    const int T = d.T, Nb = d.N, cap = a.cap;
    const double *__restrict__ B = a.Bt + d.b_off;
    // two buffers of each token array, picked by arithmetic (an indexed pointer array would live in scratch)
    auto scb = [&](int b) { return a.score + ((size_t)u * 2 + b) * cap; };
    auto pb = [&](int b) { return a.p + ((size_t)u * 2 + b) * cap * 8; };
    auto ndb = [&](int b) { return a.node + ((size_t)u * 2 + b) * cap; };
    auto hsb = [&](int b) { return a.hist + ((size_t)u * 2 + b) * cap; };
    auto upb = [&](int b) { return a.upair + ((size_t)u * 2 + b) * cap; };
    auto srb = [&](int b) { return a.dst + ((size_t)u * 2 + b) * cap; };
    int *seg_ofs = a.seg_ofs + (size_t)u * (cap + 2), *seg_cptr = a.seg_cptr + (size_t)u * (cap + 2), *seg_hist = a.seg_hist + (size_t)u * (cap + 2);
    double *seg_score = a.seg_score + (size_t)u * (cap + 2);
    int *__restrict__ slot = a.slot + (size_t)u * a.n_nodes;
    int *hprev = a.hist_prev + (size_t)u * a.Tmax, *hnode = a.hist_node + (size_t)u * a.Tmax;
    const int4 *__restrict__ ninfo = a.node_info;
    const double lpi1 = a.lpi1, lpi2 = a.lpi2;
    double *tab = dyn, *Bsl = dyn + ((a.n_units * TS + 1) & ~1);

#ifdef PCL_DEC_STAMPS
    long long st_acc[PCL_DEC_N_STAMP] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = wall_clock64();
#define STAMP(k)                                  \
    if (u == 0 && tid == 0) {                     \
        const long long now_ = wall_clock64();    \
        st_acc[k] += now_ - st_t;                 \
        st_t = now_;                              \
    }
#else
#define STAMP(k)
#endif

    for (int k = tid; k < a.n_units * TS; k += LW) {               // the left-to-right entries of every unit matrix
        const double *m = a.unit_logtrans + (size_t)(k / TS) * 25;
        const int f = k % TS;
        tab[k] = f < E ? m[(1 + f) * 5 + 1 + f] : f < 2 * E ? m[(1 + f - E) * 5 + 2 + f - E] : m[1];
    }
    for (int k = tid; k < Nb; k += LW) {
        Bsl[k] = B[k];
        if (T > 1) Bsl[NbP + k] = B[(size_t)Nb + k];
    }
    if (tid < 256) occ[tid] = 0u;
    if (tid == 0) s_cn = 0;
    __syncthreads();

    // ---- frame 0: every first-character node starts (D3) and takes its first step
    int cur = 0, n = min(a.n_roots, cap), ovf = a.n_roots > cap, nh = 0;
    for (int i = tid; i < n; i += LW) {
        const int node = a.roots[i];
        const int4 info = ninfo[node];
        double po[NE] = {0, 0, 0, 0, 0, 0}, pn[NE], best;
        int fin;
        lr_step(tab, Bsl, info.w, true, false, lpi1, lpi2, po, pn, best, fin);
#pragma unroll
        for (int j = 0; j < NE; ++j) pb(0)[(size_t)j * cap + i] = pn[j];
        scb(0)[i] = 0.0 + best;
        ndb(0)[i] = node;
        hsb(0)[i] = -1;
        upb(0)[i] = info.w;
        srb(0)[i] = i | FRESH;
        // (the node -> token map is written by every frame's step for the tokens that step in it, stamped with the frame: frame 0 needs none)
    }
    __syncthreads();
    if (tid == 0) a.trace[(size_t)u * a.Tmax] = n;

    for (int t = 1; t < T; ++t) {
        // every token array has two buffers of one parity: this frame READS a token's state where the previous frame left it (buffers
        // `cur`, at the index its src entry names) and WRITES it at the token's dense index into the other buffers; what the end of the
        // frame compacts is only the list of those indices
        const double *__restrict__ sc_o = scb(cur);
        const double *__restrict__ pin = pb(cur);
        const int *__restrict__ nd_o = ndb(cur), *__restrict__ hs_o = hsb(cur), *__restrict__ up_o = upb(cur);
        const int *__restrict__ sr = srb(cur);
        double *__restrict__ sc = scb(cur ^ 1);
        double *__restrict__ p = pb(cur ^ 1);
        int *__restrict__ nd = ndb(cur ^ 1), *__restrict__ hs = hsb(cur ^ 1), *__restrict__ up = upb(cur ^ 1);
        const double *Bs = Bsl + (t & 1) * NbP;
        // the next frame's emission row: on its way now, into LDS at the end of the frame
        double bnext[2] = {0.0, 0.0};
        const bool pre = t + 1 < T && Nb <= 2 * LW;
        if (pre) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
                if (tid + x * LW < Nb) bnext[x] = B[(size_t)(t + 1) * Nb + tid + x * LW];
        }
        const int C = ((n + LW - 1) / LW) * 64, w0 = wave * C;    // this frame's ownership of the old tokens
        // a wave's donor list: entry e in LDS below DLW, in the utterance's seg_* arrays (at the wave's token range) above
        auto put = [&](int e, int tok, int node, int hs, double s) {
            if (e < DLW) {
                const int x = wave * DLW + e;
                L_a[x] = tok;
                L_b[x] = node;
                L_h[x] = hs;
                L_s[x] = s;
            } else {
                seg_ofs[w0 + e] = tok;
                seg_cptr[w0 + e] = node;
                seg_hist[w0 + e] = hs;
                seg_score[w0 + e] = s;
            }
        };
        // ---- A: every live token takes its step; finished tokens (D1) = donors, into the wave's list in token order
        unsigned int finmask = 0u;
        int dcount = 0;
        {
            int srcv[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int i = w0 + k * 64 + lane;
                srcv[k] = (k * 64 < C && i < n) ? sr[i] : 0;
            }
#pragma unroll
            for (int kb = 0; kb < KMAX; kb += G) {
                if (kb * 64 < C) {                                 // (wave-uniform)
                    double po[G][NE], so[G];
                    int mn[G], mh[G], mu[G];                          // node, history, unit pair
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int i = w0 + (kb + g) * 64 + lane, s = srcv[kb + g] & 0x7fffffff;
                        const bool ok = (kb + g) * 64 < C && i < n;
#pragma unroll
                        for (int j = 0; j < NE; ++j) po[g][j] = ok ? pin[(size_t)j * cap + s] : -INFINITY;
                        mn[g] = ok ? nd_o[s] : 0;
                        mh[g] = ok ? hs_o[s] : -1;
                        mu[g] = ok ? up_o[s] : (int)0xffff0000;
                        so[g] = ok ? sc_o[s] : 0.0;
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if ((kb + g) * 64 < C) {                   // (wave-uniform)
                            const int i = w0 + (kb + g) * 64 + lane;
                            const bool ok = i < n;
                            double pn[NE], best;
                            int fin;
                            lr_step(tab, Bs, mu[g], false, srcv[kb + g] < 0, lpi1, lpi2, po[g], pn, best, fin);
                            const double sn = so[g] + best;        // score += max_j p_j (Decoder.py:285)
                            if (ok) {
#pragma unroll
                                for (int j = 0; j < NE; ++j) p[(size_t)j * cap + i] = pn[j];
                                sc[i] = sn;
                                nd[i] = mn[g];
                                hs[i] = mh[g];
                                up[i] = mu[g];
                            }
                            const bool don = ok && fin;
                            // the node -> token map of THIS frame: (frame << SLOT_BITS) | dense index for a token that stepped and goes on,
                            // -1 for one that finished (no live target: a new one for its node may be made by the pairs below).  An entry
                            // of another frame's stamp is stale (its token was pruned, or never stepped again) and reads as "no token":
                            // nothing has to be cleared or re-mapped when the frame's survivors are compacted
                            if (ok) slot[mn[g]] = don ? -1 : ((t << SLOT_BITS) | i);
                            const unsigned long long mask = __ballot(don);
                            if (mask != 0ull) {                    // (a few per cent of the tokens finish in a frame)
                                if (don) {
                                    finmask |= 1u << (kb + g);
                                    put(dcount + __popcll(mask & lt_mask), i, mn[g], mh[g], sn);
                                }
                                dcount += __popcll(mask);
                            }
                        }
                    }
                }
            }
        }
        if (lane == 0) wd_cnt[wave] = dcount;
        __syncthreads();
        STAMP(0)
        // ---- A2: a lane per donor of the wave: its node's children -> child offsets inside the wave; the wave's best finished
        //      word-end token (it re-seeds the first characters, D4; earliest on ties) and its first one (the first characters
        //      are created right behind that donor's children)
        {
            int crun = 0, bw_i = NONE, bw_node = 0, bw_hist = 0, fw = NONE, fw_end = 0;
            double bw = -INFINITY;
            for (int e0 = 0; e0 < dcount; e0 += 64) {
                const int e = e0 + lane;
                const bool valid = e < dcount;
                int tok = 0, node = 0, hs = 0;
                double s = 0.0;
                if (valid) {
                    if (e < DLW) {
                        const int x = wave * DLW + e;
                        tok = L_a[x];
                        node = L_b[x];
                        hs = L_h[x];
                        s = L_s[x];
                    } else {
                        tok = seg_ofs[w0 + e];
                        node = seg_cptr[w0 + e];
                        hs = seg_hist[w0 + e];
                        s = seg_score[w0 + e];
                    }
                }
                int4 info = make_int4(0, 0, 0, 0);
                if (valid) info = ninfo[node];
                const int cnt = info.y;
                int inc = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int x = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += x;
                }
                const int ofs = crun + inc - cnt;
                if (valid) {
                    if (e < DLW) {
                        L_a[wave * DLW + e] = ofs;
                        L_b[wave * DLW + e] = info.x;
                    } else {
                        seg_ofs[w0 + e] = ofs;
                        seg_cptr[w0 + e] = info.x;
                    }
                }
                const bool word = valid && info.z != 0;
                const unsigned long long wmask = __ballot(word);
                if (wmask != 0ull) {
                    if (fw == NONE) {                              // (wave-uniform: the first word-end donor of the wave)
                        const int f = __ffsll((long long)wmask) - 1;
                        fw = __shfl(tok, f, 64);
                        fw_end = __shfl(ofs + cnt, f, 64);
                    }
                    if (word && (bw_i == NONE || s > bw)) {        // (a lane's donors come in ascending token order)
                        bw = s;
                        bw_i = tok;
                        bw_node = node;
                        bw_hist = hs;
                    }
                }
                crun += __shfl(inc, 63, 64);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(bw, o, 64);
                const int oi = __shfl_xor(bw_i, o, 64), on = __shfl_xor(bw_node, o, 64), oh = __shfl_xor(bw_hist, o, 64);
                if (oi != NONE && (bw_i == NONE || ob > bw || (ob == bw && oi < bw_i))) {
                    bw = ob;
                    bw_i = oi;
                    bw_node = on;
                    bw_hist = oh;
                }
            }
            if (lane == 0) {
                wd_ch[wave] = crun;
                wd_fw[wave] = fw;
                wd_fwend[wave] = fw_end;
                wd_bw[wave] = bw;
                wd_bwi[wave] = bw_i;
                wd_bwnode[wave] = bw_node;
                wd_bwhist[wave] = bw_hist;
            }
        }
        __syncthreads();
        STAMP(1)
        // ---- C: the frame's hand-overs as ONE flattened list of (donor, target) pairs in the order the restated rules create
        //      tokens: donors in token order, each donor's children in child order, the first characters (as the children of a
        //      pseudo-donor) right behind the children of the first word-end donor, at pair R0.
        int ch_tot = 0, w_i = NONE, w_node = 0, w_dhist = 0, R0 = -1;
        double w_score = -INFINITY;
#pragma unroll
        for (int w = 0; w < LNW; ++w) {
            if (R0 < 0 && wd_fw[w] != NONE) R0 = ch_tot + wd_fwend[w];
            ch_tot += wd_ch[w];
            if (wd_bwi[w] != NONE && (w_i == NONE || wd_bw[w] > w_score || (wd_bw[w] == w_score && wd_bwi[w] < w_i))) {
                w_score = wd_bw[w];
                w_i = wd_bwi[w];
                w_node = wd_bwnode[w];
                w_dhist = wd_bwhist[w];
            }
        }
        const bool has_w = w_i != NONE;
        const int w_hist = has_w ? nh : -1;
        if (has_w) {
            if (tid == 0 && nh < a.Tmax) {                         // one history entry per frame: the winning donor's word
                hprev[nh] = w_dhist;
                hnode[nh] = w_node;
            }
            ++nh;
        }
        const int Q = ch_tot + (has_w ? a.n_roots : 0);
        //      A target is "live" when its node has a token that did not finish in this frame (finished ones left the map in A):
        //      it keeps its recursion and takes the score if strictly better (passing_in_word, Decoder.py:126-134); otherwise a
        //      new token is made behind the old ones, in pair order, and takes its first step at once (:138-139).
        int created = 0;
        for (int qc = 0; qc < Q; qc += PMAX * LW) {
            int child[PMAX], dh[PMAX], sidx[PMAX], upn[PMAX];
            double ds[PMAX];
            bool val[PMAX];
#pragma unroll
            for (int x = 0; x < PMAX; ++x) {
                const int q = qc + x * LW + tid;
                val[x] = q < Q;
                child[x] = 0;
                dh[x] = 0;
                ds[x] = 0.0;
                if (val[x]) {
                    if (has_w && q >= R0 && q < R0 + a.n_roots) {
                        child[x] = a.roots[q - R0];
                        ds[x] = w_score;
                        dh[x] = w_hist;
                    } else {
                        const int qq = (has_w && q >= R0 + a.n_roots) ? q - a.n_roots : q;
                        int w = 0, cb = 0;                         // the wave whose donors own pair qq
#pragma unroll
                        for (int ww = 0; ww < LNW - 1; ++ww) {
                            const int c = wd_ch[ww];
                            if (w == ww && qq >= cb + c) {
                                cb += c;
                                w = ww + 1;
                            }
                        }
                        const int xl = qq - cb, ne = wd_cnt[w], g0 = w * C;
                        int lo = 0, hi = ne;                       // the last donor of that wave whose pairs start at or before xl
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            const int v = mid < DLW ? L_a[w * DLW + mid] : seg_ofs[g0 + mid];
                            if (v <= xl) lo = mid + 1;
                            else hi = mid;
                        }
                        const int e = lo - 1;
                        int o, cptr;
                        if (e < DLW) {
                            const int y = w * DLW + e;
                            o = L_a[y];
                            cptr = L_b[y];
                            dh[x] = L_h[y];
                            ds[x] = L_s[y];
                        } else {
                            o = seg_ofs[g0 + e];
                            cptr = seg_cptr[g0 + e];
                            dh[x] = seg_hist[g0 + e];
                            ds[x] = seg_score[g0 + e];
                        }
                        child[x] = a.child_idx[cptr + xl - o];
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < PMAX; ++x) {
                const int sv = val[x] ? slot[child[x]] : -1;       // this frame's stamp, or the node has no live token
                sidx[x] = (sv >= 0 && (sv >> SLOT_BITS) == t) ? (sv & ((1 << SLOT_BITS) - 1)) : -1;
                upn[x] = val[x] ? ninfo[child[x]].w : 0;
            }
            unsigned long long nmask[PMAX];
#pragma unroll
            for (int x = 0; x < PMAX; ++x) {
                if (val[x] && sidx[x] >= 0) {
                    if (ds[x] > sc[sidx[x]]) {
                        sc[sidx[x]] = ds[x];
                        hs[sidx[x]] = dh[x];
                    }
                }
                nmask[x] = __ballot(val[x] && sidx[x] < 0);
                if (lane == 0) wp[x][wave] = __popcll(nmask[x]);
            }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < PMAX; ++x) {
                int base = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < LNW; ++w) {
                    const int y = wp[x][w];
                    if (w < wave) base += y;
                    tot += y;
                }
                if (val[x] && sidx[x] < 0) {
                    const int pos = n + created + base + __popcll(nmask[x] & lt_mask);
                    if (pos < cap) {                               // (slots n .. cap-1 of the current buffers)
                        double po[NE] = {0, 0, 0, 0, 0, 0}, pn[NE], best;
                        int fin;
                        lr_step(tab, Bs, upn[x], true, false, lpi1, lpi2, po, pn, best, fin);
#pragma unroll
                        for (int j = 0; j < NE; ++j) p[(size_t)j * cap + pos] = pn[j];
                        sc[pos] = ds[x] + best;
                        nd[pos] = child[x];
                        hs[pos] = dh[x];
                        up[pos] = upn[x];
                    }
                }
                created += tot;
            }
            __syncthreads();
        }
        const int n_new = min(created, cap - n);
        if (created > cap - n) ovf = 1;
        if (Q <= 0) __syncthreads();                               // (the donor lists are read no more: the pool changes tenant)
        STAMP(2)
        // ---- E: pruning over the tokens that were alive before the frame and did not finish (Decoder.py:159-167): nothing
        //      below min_distinct different scores, else the int(width (1 - beam)) lowest go (stable ascending order: ties by
        //      token order).  One pass gives the width, the key range and a hashed occupancy map (different bins => different
        //      scores); then the m-th smallest key: a 12-bit digit below the range's common prefix picks a bin, and the few keys
        //      of that bin are ranked directly.
        for (int k = tid; k < HBINS; k += LW) hist[k] = 0u;
        unsigned long long keys[KMAX];
        int cnt = 0;
        unsigned long long kmn = ~0ull, kmx = 0ull;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = w0 + k * 64 + lane;
            const bool in = k * 64 < C && i < n && !((finmask >> k) & 1u);
            keys[k] = in ? pcl_okey(sc[i]) : NOKEY;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned long long key = keys[k];
            if (key != NOKEY) {
                ++cnt;
                kmn = min(kmn, key);
                kmx = max(kmx, key);
                atomicOr(&occ[(unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 56)], 1u);
            }
        }
        cnt = pcl_wave_sum(cnt);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmn = min(kmn, (unsigned long long)__shfl_xor((long long)kmn, o, 64));
            kmx = max(kmx, (unsigned long long)__shfl_xor((long long)kmx, o, 64));
        }
        if (lane == 0) {
            we_cnt[wave] = cnt;
            red_u[0][wave] = kmn;
            red_u[1][wave] = kmx;
        }
        __syncthreads();
        const int bins = __syncthreads_count(tid < 256 && occ[tid] != 0u);
        if (tid < 256) occ[tid] = 0u;                              // (next read: a frame and many barriers away)
        int n_old = 0;
        unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
        for (int w = 0; w < LNW; ++w) {
            n_old += we_cnt[w];
            kmin = min(kmin, red_u[0][w]);
            kmax = max(kmax, red_u[1][w]);
        }
        STAMP(3)
        const int m_cut = (int)((double)n_old * (1.0 - a.beam));                // int(width * (1 - beam))
        bool prune = m_cut > 0 && n_old >= a.min_distinct;
        if (prune && bins < a.min_distinct) {                                   // few bins: count the distinct scores exactly
            unsigned long long prev = 0ull;
            bool have_prev = false;
            int distinct = 0;
            for (int round = 0; round < a.min_distinct; ++round) {              // the next larger key, min_distinct times
                unsigned long long mn = ~0ull;
                bool any = false;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const unsigned long long key = keys[k];
                    if (key != NOKEY && (!have_prev || key > prev) && (!any || key < mn)) {
                        mn = key;
                        any = true;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long om = (unsigned long long)__shfl_xor((long long)mn, o, 64);
                    const int oa = __shfl_xor((int)any, o, 64);
                    if (oa && (!any || om < mn)) {
                        mn = om;
                        any = true;
                    }
                }
                __syncthreads();
                if (lane == 0) {
                    red_u[0][wave] = mn;
                    red_i[wave] = any;
                }
                __syncthreads();
                unsigned long long g = ~0ull;
                bool gany = false;
                for (int w = 0; w < LNW; ++w)
                    if (red_i[w] && (!gany || red_u[0][w] < g)) {
                        g = red_u[0][w];
                        gany = true;
                    }
                if (!gany) break;
                prev = g;
                have_prev = true;
                ++distinct;
            }
            prune = distinct >= a.min_distinct;
            __syncthreads();
        }
        unsigned int prunemask = 0u;
        if (prune) {
            unsigned long long sel = kmin;
            int rank = m_cut - 1;
            const unsigned long long diff = kmin ^ kmax;
            if (diff != 0ull) {
                const int hb = 63 - __clzll((long long)diff);                  // the highest bit in which the keys differ
                const unsigned long long lowmask = (2ull << hb) - 1ull;        // (hb = 63: all ones)
                int shift = max(hb - (HB - 1), 0);
                unsigned long long pmask = 0ull, pval = 0ull;                   // keys still in play: (key & pmask) == pval
                for (;;) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const unsigned long long key = keys[k];
                        if (key != NOKEY && (key & pmask) == pval) atomicAdd(&hist[(unsigned int)(key >> shift) & (HBINS - 1)], 1u);
                    }
                    __syncthreads();
                    int h[BPT], s4 = 0;                                         // BPT bins per thread: the bin holding rank
#pragma unroll
                    for (int j = 0; j < BPT; ++j) {
                        h[j] = (int)hist[BPT * tid + j];
                        s4 += h[j];
                        hist[BPT * tid + j] = 0u;
                    }
                    int inc = s4;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int x = __shfl_up(inc, o, 64);
                        if (lane >= o) inc += x;
                    }
                    if (lane == 63) ws_scan[wave] = inc;
                    __syncthreads();
                    int base = 0;
#pragma unroll
                    for (int w = 0; w < LNW; ++w)
                        if (w < wave) base += ws_scan[w];
                    const int lo = base + inc - s4;
                    if (rank >= lo && rank < lo + s4) {
                        int acc = lo, j = 0;
                        for (; j < BPT - 1; ++j) {
                            if (acc + h[j] > rank) break;
                            acc += h[j];
                        }
                        s_sel = BPT * tid + j;
                        s_rank = rank - acc;
                        s_cnt = h[j];
                    }
                    __syncthreads();
                    const int c = s_cnt;
                    rank = s_rank;
                    pmask |= (unsigned long long)(HBINS - 1) << shift;
                    pval = (pval & ~((unsigned long long)(HBINS - 1) << shift)) | ((unsigned long long)s_sel << shift);
                    if (shift == 0) {
                        sel = (kmin & ~lowmask) | (pval & lowmask);
                        break;
                    }
                    if (c <= CAND) {                                            // the bin's keys, ranked directly
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) {
                            const unsigned long long key = keys[k];
                            if (key != NOKEY && (key & pmask) == pval) cand[atomicAdd(&s_cn, 1)] = key;
                        }
                        __syncthreads();
                        for (int x = tid; x < c; x += LW) {
                            const unsigned long long kx = cand[x];
                            int less = 0, eq = 0;
                            for (int y = 0; y < c; ++y) {
                                const unsigned long long ky = cand[y];
                                less += ky < kx;
                                eq += ky == kx;
                            }
                            if (less <= rank && rank < less + eq) {             // (equal keys write the same two values)
                                s_key = kx;
                                s_rank = rank - less;
                            }
                        }
                        __syncthreads();
                        sel = s_key;
                        rank = s_rank;
                        if (tid == 0) s_cn = 0;
                        break;
                    }
                    shift = max(shift - HB, 0);
                }
            }
            STAMP(6)
            // everything below the selected key goes, and of the tokens equal to it the first (rank + 1) in token order
            int eq = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) eq += keys[k] == sel;
            eq = pcl_wave_sum(eq);
            if (lane == 0) w_eq[wave] = eq;
            __syncthreads();
            int run = 0;
#pragma unroll
            for (int w = 0; w < LNW; ++w)
                if (w < wave) run += w_eq[w];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const unsigned long long key = keys[k];
                const bool is_eq = key == sel;                                 // (sel is a real key, never NOKEY)
                const unsigned long long mask = __ballot(is_eq);
                if (key != NOKEY && (key < sel || (is_eq && run + __popcll(mask & lt_mask) <= rank))) prunemask |= 1u << k;
                run += __popcll(mask);
            }
        }
        STAMP(4)
        // ---- F: stable compaction of the INDEX LIST: the survivors of the old tokens, then the new ones (the node -> token map is
        //      re-stamped by the next frame's step: nothing to follow here).
        //      Scores, nodes, histories, unit pairs and p stay where this frame wrote them: the next frame's step gathers them through
        //      the list (an ascending sequence with the pruned tokens' gaps: nearly coalesced) and writes them dense again -- round 4
        //      moved score + meta here as well (24 B in, 24 B out per token and frame, and a dependent load round)
        int *__restrict__ srn = srb(cur ^ 1);
        const unsigned int dead = finmask | prunemask;
        int keep_cnt = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) keep_cnt += (k * 64 < C && w0 + k * 64 + lane < n && !((dead >> k) & 1u)) ? 1 : 0;
        keep_cnt = pcl_wave_sum(keep_cnt);
        if (lane == 0) wf_keep[wave] = keep_cnt;
        __syncthreads();
        int krun = 0, n_keep = 0;
#pragma unroll
        for (int w = 0; w < LNW; ++w) {
            const int x = wf_keep[w];
            if (w < wave) krun += x;
            n_keep += x;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k * 64 >= C) continue;                             // (wave-uniform)
            const int i = w0 + k * 64 + lane;
            const bool old = i < n && !((finmask >> k) & 1u), keep = old && !((prunemask >> k) & 1u);
            const unsigned long long mask = __ballot(keep);
            if (keep) srn[krun + __popcll(mask & lt_mask)] = i;    // where the token's state sits in this frame's buffers
            krun += __popcll(mask);
        }
        for (int j = tid; j < n_new; j += LW) srn[n_keep + j] = (n + j) | FRESH;
        if (t + 1 < T) {                                           // the next frame's emission row
            double *Bn = Bsl + ((t + 1) & 1) * NbP;
            if (pre) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
                    if (tid + x * LW < Nb) Bn[tid + x * LW] = bnext[x];
            } else {
                for (int k = tid; k < Nb; k += LW) Bn[k] = B[(size_t)(t + 1) * Nb + k];
            }
        }
        __syncthreads();
        STAMP(5)
        n = n_keep + n_new;
        cur ^= 1;
        if (tid == 0) a.trace[(size_t)u * a.Tmax + t] = n;
    }
    // ---- transfer (Decoder.py:175-187): the `candidate` best tokens, ties in token order (token k's state sits at index src[k] of the
    //      buffers the last frame wrote)
    const double *sc = scb(cur);
    const int *nd = ndb(cur), *hs = hsb(cur), *srf = srb(cur);
    int *taken = srb(cur ^ 1);                                     // (the idle src buffer: 4 = taken)
    for (int i = tid; i < n; i += LW) taken[i] = 0;
    __syncthreads();
    int n_out = 0;
    for (int cc = 0; cc < a.candidate && cc < n; ++cc) {
        double b = -INFINITY;
        int bi = NONE;
        for (int i = tid; i < n; i += LW) {
            const double v = sc[srf[i] & 0x7fffffff];
            if (taken[i] != 4 && (bi == NONE || v > b)) {          // (strictly greater keeps the earliest on ties)
                b = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(b, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != NONE && (bi == NONE || ob > b || (ob == b && oi < bi))) {
                b = ob;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_d[wave] = b;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            double g = -INFINITY;
            int gi = NONE;
            for (int w = 0; w < LNW; ++w)
                if (red_i[w] != NONE && (gi == NONE || red_d[w] > g || (red_d[w] == g && red_i[w] < gi))) {
                    g = red_d[w];
                    gi = red_i[w];
                }
            const int at = srf[gi] & 0x7fffffff;
            a.out_node[(size_t)u * a.candidate + cc] = nd[at];
            a.out_score[(size_t)u * a.candidate + cc] = sc[at];
            a.out_hist[(size_t)u * a.candidate + cc] = hs[at];
            taken[gi] = 4;
        }
        ++n_out;
        __syncthreads();
    }
    if (tid == 0) {
        a.out_n[u] = n_out;
        a.hist_n[u] = min(nh, a.Tmax);
        a.overflow[u] = ovf;
#ifdef PCL_DEC_STAMPS
        if (u == 0 && a.stamps)
            for (int k = 0; k < PCL_DEC_N_STAMP; ++k) a.stamps[k] = st_acc[k];
#endif
    }
}

constexpr size_t LR_STATIC_LDS = POOL_BYTES + 256 * 4 + 2048;      // (what the kernel declares, rounded up)
constexpr size_t LR_LDS_BUDGET = 150u * 1024u;

size_t lr_dyn_bytes(const pcl_ctx *ctx, int n_rows) {
    const int NbP = (n_rows + 1) & ~1;
    return ((size_t)((ctx->n_units * TS + 1) & ~1) + 2 * (size_t)NbP) * sizeof(double);
}

}  // namespace

bool pcl_decode_lr_applicable(const pcl_ctx *ctx, int n_rows, int cap, int t_max) {
    if (ctx->S != 5 || cap > 16 * LW || cap > (1 << SLOT_BITS)) return false;
    if (t_max >= (1 << (31 - SLOT_BITS))) return false;            // (the frame stamp of the node -> token map: utterances of up to 131071 frames)
    if (lr_dyn_bytes(ctx, n_rows) + LR_STATIC_LDS > LR_LDS_BUDGET) return false;
    const double *lt = ctx->unit_logtrans.data();
    for (int u = 0; u < ctx->n_units; ++u)
        for (int r = 0; r <= E; ++r)                                // (the exit row takes no part in a sentence HMM)
            for (int c = 0; c < 5; ++c) {
                const bool may = c == r + 1 || (c == r && r >= 1);
                if (!may && lt[((size_t)u * 5 + r) * 5 + c] != -INFINITY) return false;
            }
    return true;
}

int pcl_decode_lr_launch(pcl_ctx *ctx, const DecArgs &a, int U, int n_rows) {
    const int cap = a.cap, NbP = (n_rows + 1) & ~1;
    const size_t dyn = lr_dyn_bytes(ctx, n_rows);
#define PCL_DECLR_LAUNCH(K)                                                                                                       \
    do {                                                                                                                          \
        if (dyn + LR_STATIC_LDS > 64u * 1024u)                                                                                    \
            HIPCHK(ctx, hipFuncSetAttribute((const void *)hmm_decode_lr_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
        hipLaunchKernelGGL((hmm_decode_lr_kernel<K>), dim3(U), dim3(LW), dyn, ctx->stream, a, NbP);                               \
    } while (0)
    if (cap <= 8 * LW) PCL_DECLR_LAUNCH(8);
    else PCL_DECLR_LAUNCH(16);
#undef PCL_DECLR_LAUNCH
    return PCL_OK;
}

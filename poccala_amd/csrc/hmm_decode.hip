// hmm_decode.hip -- frame-synchronous token passing over the pronunciation tree, batched one workgroup per utterance
// (gfx950).  SURVEY.md section 8(a) row A16 and 8(f) rank 3: the decode half of BASELINE config 5.
//
// What is restated from the reference (all of it dead code there: Decoder.py cannot be imported):
//   Token.viterbi        Decoder.py:250-288   first step p = ln pi + B[:,t]; later p_j = max_i(p_i + ln A_ij) + B_j;
//                                              score += max_j p_j; mark = first argmax
//   Token.__init__       Decoder.py:222-236   the token's HMM = AcousticModel.embedded of the node's units
//                                              (AcousticModel.py:957-1014): uniform pi, entry row 0, exit row -inf
//   token_passing        Decoder.py:91-111    one step of every token per frame, finished tokens hand over and go
//   passing_in_word      Decoder.py:114-143   children of the tree node get the finished token's score; a child that
//                                              already has a token takes the score if strictly better and keeps its p
//   pruning              Decoder.py:159-167   nothing below 8 distinct scores; else the int(width (1 - beam)) lowest go
//   transfer             Decoder.py:175-187   the `candidate` best tokens at the end
// and the gaps D1..D5 that had to be filled because the source cannot run (finished <=> best state is the last emitting
// one; tokens keyed by tree node; all first-character nodes start; a finished word re-seeds every first-character node
// with a uniform language model and one history entry per frame; frame semantics "all step, then all hand over") are
// spelled out next to the CPU restatement the parity tests hold this kernel to, bit for bit (include/poccala_hip.h names it).
// The restatement's recursion, pruning, frame loop and in-word hand-over are pinned by golden G14 (those pieces of Decoder.py run
// with a stand-in for its missing import); D1..D5 are the builder's completion of what the source cannot do.
//
// Mapping.  The emissions are the all-state matrix of a scoring batch (rows entry, 0..J-1, exit: pcl_batch_score), time
// major, so a frame's J values are contiguous; the frame's row and the unit matrices (183 x 25 doubles) are staged in
// LDS.  A token is 8 lanes in its step (N = two units x three emitting states + 2 <= 8: lane j holds p_j, two tokens per
// lane group in flight) and one lane in the bookkeeping phases; its state lives in separate arrays (score, p[8], node,
// history, unit pair), so every pass is coalesced.  All float64: scores reach -1e5.  Per frame the workgroup runs
//   step -> donors (finished tokens), the best word-end donor, and the donors' children as ONE flattened list of
//   (donor, target) pairs -> merge / create, a pair per thread (a node with hundreds of children is spread over the
//   workgroup) -> first step of the new tokens -> prune (one pass for the width, the key range and a hashed occupancy map
//   that settles the "8 distinct scores" rule; radix select below the keys' common prefix; ties by token order) ->
//   stable compaction of the small fields into the other set of arrays (the 64 bytes of p stay where the step wrote them:
//   the next step reads them through a source map and writes the other p buffer in the new order),
// with workgroup barriers between phases; ordered prefix sums are a ballot per 64 tokens plus one exchange of 16 wave
// totals.  A tree node has at most one live token and one parent, so no hand-over needs an atomic and the result does not
// depend on timing.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "hmm_decode_args.h"

namespace {

#ifndef PCL_DEC_DW
#define PCL_DEC_DW 1024
#endif
constexpr int DW = PCL_DEC_DW;      // threads per workgroup
constexpr int NWV = DW / 64;        // wavefronts per workgroup
constexpr int NS = 8;               // lanes per token = states of its HMM at most
constexpr int TG = DW / NS;         // tokens stepped per pass of the workgroup
constexpr int SEG_LDS = 4096;       // donor segments whose offsets are searched in LDS (more: searched in HBM)
constexpr int NONE = 0x7fffffff;
[[maybe_unused]] constexpr int N_STAMP = PCL_DEC_N_STAMP;

// What a lane (state j = sub of its token) needs of ln A of the embedded HMM (AcousticModel.py:979-989), fixed per thread:
// predecessor i reaches j through entry rc[i] of its unit's (S,S) matrix, or not at all.
struct LaneCtx {
    int e, SS, mypos, myk;
    int rc[NS];                     // entry r S + c of the unit matrix, + 256 when the predecessor sits in the second unit; -1: none
    double lpi1, lpi2;
};

// One step of one token by its 8 lanes (they sit in one wave and run in lockstep).  FIRST: p = ln pi + B[:,t]
// (Decoder.py:270); else the max recursion (:278-283).  best = max_j p_j and fin (D1) come back on every lane.
template <bool FIRST>
__device__ __forceinline__ void token_step(const LaneCtx &c, const double *lt, const double *Bs, int up, double pold, int sub, double &pj,
                                           double &best, int &fin) {
    const int u0 = up & 0xffff, u1 = (int)((unsigned int)up >> 16);
    const int nu = (u1 == 0xffff) ? 1 : 2, N = c.e * nu + 2;
    double bj = -INFINITY;
    if (sub == 0) bj = 0.0;                                        // entry VirtualState: ln 1 (AcousticModel.py:218)
    else if (sub < N - 1) bj = Bs[1 + (c.mypos ? u1 : u0) * c.e + c.myk];    // (exit VirtualState: ln 0, :219)
    pj = -INFINITY;
    if (FIRST) {
        if (sub < N) pj = (nu == 1 ? c.lpi1 : c.lpi2) + bj;
    } else {
        double m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NS; ++i) {                             // every lane's old value travels by shuffle
            const double pi = __shfl(pold, i, NS);
            if (i < N - 1 && c.rc[i] >= 0 && sub < N) m = fmax(m, pi + lt[((c.rc[i] & 256) ? u1 : u0) * c.SS + (c.rc[i] & 255)]);
        }
        if (sub < N) pj = m + bj;
    }
    best = pj;
    int arg = (sub < N) ? sub : NS;
#pragma unroll
    for (int o = 1; o < NS; o <<= 1) {                             // max and FIRST argmax over the token's lanes (:263-268)
        const double ob = __shfl_xor(best, o, NS);
        const int oa = __shfl_xor(arg, o, NS);
        if (ob > best || (ob == best && oa < arg)) {
            best = ob;
            arg = oa;
        }
    }
    fin = arg >= N - 2;
}

// tokens [lo, hi) take a step, two per 8-lane group and pass, software-pipelined over the passes: while pass k computes,
// the old p of pass k+1 and the indices of pass k+2 are on their way (the old p of token i sits at src[i] of the other p
// buffer: the compaction at the end of a frame only writes that map).
struct StepIn {
    int up[2], at[2];
    double sc[2];
};
template <bool FIRST>
__device__ __forceinline__ void step_range(const LaneCtx &c, const double *lt, const double *Bs, int lo, int hi, const int *up, const double *pin,
                                           const int *src, double *p, double *sc, int *flag, int tk8, int sub) {
    auto fetch = [&](int i0, StepIn &in) {                         // indices, unit pairs, scores of the pass that starts at i0
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = i0 + tk8 + x * TG;
            const bool ok = i < hi;
            in.up[x] = ok ? up[i] : (int)0xffff0000;
            in.at[x] = (!FIRST && ok) ? src[i] : 0;
            in.sc[x] = (ok && sub == 0) ? sc[i] : 0.0;
        }
    };
    auto fetch_p = [&](int i0, const StepIn &in, double (&po)[2]) {
#pragma unroll
        for (int x = 0; x < 2; ++x) po[x] = (!FIRST && i0 + tk8 + x * TG < hi) ? pin[(size_t)in.at[x] * NS + sub] : 0.0;
    };
    if (lo >= hi) return;
    StepIn in0, in1, in2;
    double po0[2], po1[2];
    fetch(lo, in0);
    fetch(lo + 2 * TG, in1);
    fetch_p(lo, in0, po0);
    for (int i0 = lo; i0 < hi; i0 += 2 * TG) {
        fetch(i0 + 4 * TG, in2);
        fetch_p(i0 + 2 * TG, in1, po1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = i0 + tk8 + x * TG;
            double pj, best;
            int fin;
            token_step<FIRST>(c, lt, Bs, in0.up[x], po0[x], sub, pj, best, fin);
            if (i < hi) {
                p[(size_t)i * NS + sub] = pj;
                if (sub == 0) {
                    sc[i] = in0.sc[x] + best;                      // score += max_j p_j (Decoder.py:285)
                    flag[i] = fin;
                }
            }
        }
        in0 = in1;
        in1 = in2;
        po0[0] = po1[0];
        po0[1] = po1[1];
    }
}

// TLDS: the unit matrices and the frame's emission row are staged in LDS (dynamic: (n_units S S + N) doubles).
// Light phases use a wave-blocked ownership of the tokens: wave w owns [w C, (w+1) C), lane l its tokens w C + 64 k + l --
// coalesced, and an ORDERED prefix over the tokens is a ballot per 64 tokens plus one exchange of 16 wave totals.
// KMAX: a lane owns at most KMAX old tokens (cap <= DW KMAX): their sort keys stay in registers through the pruning phase.
#ifdef PCL_DEC_WAVES
#define PCL_DEC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PCL_DEC_WAVES, PCL_DEC_WAVES)))
#else
#define PCL_DEC_WAVES_ATTR
#endif
template <bool TLDS, int KMAX>
__global__ __launch_bounds__(DW) PCL_DEC_WAVES_ATTR void hmm_decode_kernel(DecArgs a) {
    extern __shared__ double dyn[];
    __shared__ int seg_l[SEG_LDS + 1];
    __shared__ unsigned int hist256[256];
    __shared__ int wsum[2][NWV];
    __shared__ double red_d[NWV];
    __shared__ unsigned long long red_u[2][NWV];
    __shared__ int red_i[2][NWV];
    __shared__ unsigned long long s_sel;
    __shared__ int s_i[4];
    __shared__ double s_d[1];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const UttDesc d = a.utts[u];
    const int T = d.T, Nb = d.N, cap = a.cap;
    const double *B = a.Bt + d.b_off;
    double *scb[2] = {a.score + (size_t)u * 2 * cap, a.score + ((size_t)u * 2 + 1) * cap};
    double *pb[2] = {a.p + (size_t)u * 2 * cap * NS, a.p + ((size_t)u * 2 + 1) * cap * NS};
    int *ndb[2] = {a.node + (size_t)u * 2 * cap, a.node + ((size_t)u * 2 + 1) * cap};
    int *hsb[2] = {a.hist + (size_t)u * 2 * cap, a.hist + ((size_t)u * 2 + 1) * cap};
    int *upb[2] = {a.upair + (size_t)u * 2 * cap, a.upair + ((size_t)u * 2 + 1) * cap};
    int *flag = a.flag + (size_t)u * cap, *src = a.dst + (size_t)u * cap;
    int *seg_ofs = a.seg_ofs + (size_t)u * (cap + 2), *seg_cptr = a.seg_cptr + (size_t)u * (cap + 2), *seg_hist = a.seg_hist + (size_t)u * (cap + 2);
    double *seg_score = a.seg_score + (size_t)u * (cap + 2);
    int *slot = a.slot + (size_t)u * a.n_nodes;
    int *hprev = a.hist_prev + (size_t)u * a.Tmax, *hnode = a.hist_node + (size_t)u * a.Tmax;
    const int tk8 = tid >> 3, sub = tid & 7;                       // token group of 8 lanes

    LaneCtx c;
    c.e = a.S - 2;
    c.SS = a.S * a.S;
    c.mypos = (sub == 0) ? 0 : (sub - 1) / c.e;
    c.myk = (sub == 0) ? 0 : (sub - 1) % c.e;
    c.lpi1 = a.lpi1;
    c.lpi2 = a.lpi2;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int pos = (i == 0) ? 0 : (i - 1) / c.e, r = (i == 0) ? 0 : 1 + (i - 1) % c.e, col = sub - pos * c.e;
        c.rc[i] = (col >= 0 && col < a.S) ? (r * a.S + col) | (pos ? 256 : 0) : -1;
    }
    const double *lt;
    double *Bs_l = nullptr;
    if (TLDS) {
        for (int k = tid; k < a.n_units * c.SS; k += DW) dyn[k] = a.unit_logtrans[k];
        lt = dyn;
        Bs_l = dyn + a.n_units * c.SS;
    } else {
        lt = a.unit_logtrans;
    }
    auto pack_units = [&](int node) -> int {
        const int u0 = a.node_units[2 * node];
        const int u1 = (a.node_nunits[node] == 2) ? a.node_units[2 * node + 1] : 0xffff;
        return u0 | (u1 << 16);
    };
#ifdef PCL_DEC_STAMPS
    long long st_acc[N_STAMP] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = wall_clock64();
#define STAMP(k)                                  \
    if (u == 0 && tid == 0) {                     \
        const long long now_ = wall_clock64();    \
        st_acc[k] += now_ - st_t;                 \
        st_t = now_;                              \
    }
#else
#define STAMP(k)
#endif

    // ---- frame 0: every first-character node starts (D3)
    int cur = 0, n = min(a.n_roots, cap), ovf = a.n_roots > cap, nh = 0;
    for (int i = tid; i < n; i += DW) {
        const int node = a.roots[i];
        ndb[0][i] = node;
        hsb[0][i] = -1;
        scb[0][i] = 0.0;
        upb[0][i] = pack_units(node);
        slot[node] = i;
    }
    if (TLDS)
        for (int k = tid; k < Nb; k += DW) Bs_l[k] = B[k];
    __syncthreads();
    step_range<true>(c, lt, TLDS ? Bs_l : B, 0, n, upb[0], nullptr, nullptr, pb[0], scb[0], flag, tk8, sub);
    for (int i = tid; i < n; i += DW) src[i] = i;
    __syncthreads();
    if (tid == 0) a.trace[(size_t)u * a.Tmax] = n;
    int pcur = 0;                                                  // the p buffer the tokens' values are in (at src[])

    for (int t = 1; t < T; ++t) {
        double *sc = scb[cur], *p = pb[pcur ^ 1];
        const double *pin = pb[pcur];
        int *nd = ndb[cur], *hs = hsb[cur], *up = upb[cur];
        const double *Bf = B + (size_t)t * Nb;
        if (TLDS) {
            for (int k = tid; k < Nb; k += DW) Bs_l[k] = Bf[k];
            __syncthreads();
        }
        const double *Bs = TLDS ? Bs_l : Bf;
        const int C = ((n + DW - 1) / DW) * 64, w0 = wave * C;     // this frame's ownership of the old tokens
        // ---- (1) every live token takes its step
        step_range<false>(c, lt, Bs, 0, n, up, pin, src, p, sc, flag, tk8, sub);
        __syncthreads();
        STAMP(0)
        // ---- (2) donors = finished tokens.  The best finished word-end token (earliest on ties) re-seeds the first
        //      characters (D4); first_w = the first finished word-end token: the roots are created right after its children
        int nd_cnt = 0, ch_cnt = 0, bw_i = NONE, fw = NONE;
        double bw = -INFINITY;
        for (int k = 0; k < C; k += 64) {
            const int i = w0 + k + lane;
            if (i < n && (flag[i] & 1)) {
                const int node = nd[i];
                ++nd_cnt;
                ch_cnt += a.child_ptr[node + 1] - a.child_ptr[node];
                if (a.node_word[node]) {
                    fw = min(fw, i);
                    const double s = sc[i];
                    if (bw_i == NONE || s > bw) {                  // (a thread's tokens come in ascending order)
                        bw = s;
                        bw_i = i;
                    }
                }
            }
        }
        nd_cnt = pcl_wave_sum(nd_cnt);
        ch_cnt = pcl_wave_sum(ch_cnt);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(bw, o, 64);
            const int oi = __shfl_xor(bw_i, o, 64);
            if (oi != NONE && (bw_i == NONE || ob > bw || (ob == bw && oi < bw_i))) {
                bw = ob;
                bw_i = oi;
            }
            fw = min(fw, __shfl_xor(fw, o, 64));
        }
        if (lane == 0) {
            wsum[0][wave] = nd_cnt;
            wsum[1][wave] = ch_cnt;
            red_d[wave] = bw;
            red_i[0][wave] = bw_i;
            red_i[1][wave] = fw;
        }
        __syncthreads();
        if (tid == 0) {
            double b = -INFINITY;
            int bi = NONE, f = NONE;
            for (int w = 0; w < NWV; ++w) {
                if (red_i[0][w] != NONE && (bi == NONE || red_d[w] > b || (red_d[w] == b && red_i[0][w] < bi))) {
                    b = red_d[w];
                    bi = red_i[0][w];
                }
                f = min(f, red_i[1][w]);
            }
            s_d[0] = b;
            s_i[0] = bi;
            s_i[1] = f;
            s_i[2] = -1;
            if (bi != NONE) {                                      // one history entry per frame: the winning donor's word
                if (nh < a.Tmax) {
                    hprev[nh] = hs[bi];
                    hnode[nh] = nd[bi];
                }
                s_i[2] = nh;
            }
        }
        int dbase = 0, cbase = 0, nd_tot = 0, ch_tot = 0;
        for (int w = 0; w < NWV; ++w) {
            const int x = wsum[0][w], y = wsum[1][w];
            if (w < wave) {
                dbase += x;
                cbase += y;
            }
            nd_tot += x;
            ch_tot += y;
        }
        __syncthreads();
        const double w_score = s_d[0];
        const int w_i = s_i[0], first_w = s_i[1], w_hist = s_i[2];
        const bool has_w = w_i != NONE;
        if (has_w) ++nh;
        // the frame's hand-overs as ONE flattened list of (donor, target) pairs in the order the restated rules create
        // tokens: donors in token order, each donor's children in child order, the first characters as the children of a
        // pseudo-donor right behind the first word-end donor.  seg_ofs = where each donor's pairs start.
        const int nseg = nd_tot + (has_w ? 1 : 0), Q = ch_tot + (has_w ? a.n_roots : 0);
        const bool use_l = nseg <= SEG_LDS;
        {
            int drun = dbase, crun = cbase;
            for (int k = 0; k < C; k += 64) {
                const int i = w0 + k + lane;
                const bool fin = i < n && (flag[i] & 1);
                int cptr = 0, cnt = 0;
                if (fin) {
                    const int node = nd[i];
                    cptr = a.child_ptr[node];
                    cnt = a.child_ptr[node + 1] - cptr;
                }
                const unsigned long long mask = __ballot(fin);
                const int r = drun + __popcll(mask & lt_mask);
                int inc = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int x = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += x;
                }
                if (fin) {
                    const int after = (has_w && i > first_w) ? 1 : 0, seg = r + after;
                    const int ofs = crun + inc - cnt + (after ? a.n_roots : 0);
                    seg_ofs[seg] = ofs;
                    if (use_l) seg_l[seg] = ofs;
                    seg_cptr[seg] = cptr;
                    seg_hist[seg] = hs[i];
                    seg_score[seg] = sc[i];
                    if (has_w && i == first_w) {
                        seg_ofs[seg + 1] = ofs + cnt;
                        if (use_l) seg_l[seg + 1] = ofs + cnt;
                        seg_cptr[seg + 1] = -1;
                        seg_hist[seg + 1] = w_hist;
                        seg_score[seg + 1] = w_score;
                    }
                }
                drun += __popcll(mask);
                crun += __shfl(inc, 63, 64);
            }
        }
        __syncthreads();
        STAMP(1)
        // ---- (3) merges and creations, one pair per thread (passing_in_word, Decoder.py:114-143).  A target is "live"
        //      when its node has a token that did not finish in this frame: it keeps its recursion and takes the score if
        //      strictly better (:126-134); otherwise a new token is made behind the old ones, in pair order.
        int created = 0;
        for (int q0 = 0; q0 < Q; q0 += DW) {
            const int q = q0 + tid;
            bool isnew = false;
            int child = 0, dh = 0;
            double ds = 0.0;
            if (q < Q) {
                int lo = 0, hi = nseg;                             // the last segment that starts at or before q
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const int v = use_l ? seg_l[mid] : seg_ofs[mid];
                    if (v <= q) lo = mid + 1;
                    else hi = mid;
                }
                const int seg = lo - 1, o = use_l ? seg_l[seg] : seg_ofs[seg], cptr = seg_cptr[seg];
                ds = seg_score[seg];
                dh = seg_hist[seg];
                child = (cptr < 0) ? a.roots[q - o] : a.child_idx[cptr + q - o];
                const int s = slot[child];
                if (s >= 0 && !(flag[s] & 1)) {
                    if (ds > sc[s]) {
                        sc[s] = ds;
                        hs[s] = dh;
                    }
                } else {
                    isnew = true;
                }
            }
            const unsigned long long mask = __ballot(isnew);
            if (lane == 0) wsum[0][wave] = __popcll(mask);
            __syncthreads();
            int base = 0, tot = 0;
            for (int w = 0; w < NWV; ++w) {
                const int x = wsum[0][w];
                if (w < wave) base += x;
                tot += x;
            }
            if (isnew) {
                const int pos = n + created + base + __popcll(mask & lt_mask);
                if (pos < cap) {                                   // (slots n .. cap-1 of the current buffer)
                    nd[pos] = child;
                    hs[pos] = dh;
                    sc[pos] = ds;
                    up[pos] = pack_units(child);
                }
            }
            created += tot;
            __syncthreads();
        }
        const int n_new = min(created, cap - n);
        if (created > cap - n) ovf = 1;
        STAMP(2)
        // the new tokens take their first step at once (Decoder.py:138-139)
        step_range<true>(c, lt, Bs, n, n + n_new, up, nullptr, nullptr, p, sc, flag, tk8, sub);
        STAMP(3)
        // ---- (4) pruning over the tokens that were alive before the frame and did not finish (Decoder.py:159-167):
        //      nothing below min_distinct different scores, else the int(width (1 - beam)) lowest go (stable ascending
        //      order: ties by token order).  One pass gives the width, the key range and a hashed occupancy map (different
        //      bins => different scores); the m-th smallest key by radix select below the range's common prefix.
        if (tid < 256) hist256[tid] = 0u;
        __syncthreads();
        constexpr unsigned long long NOKEY = ~0ull;                             // not an old unfinished token
        unsigned long long keys[KMAX];
        int cnt = 0;
        unsigned long long kmn = ~0ull, kmx = 0ull;
#pragma unroll
        for (int kk = 0; kk < KMAX; ++kk) {
            const int i = w0 + kk * 64 + lane;
            keys[kk] = NOKEY;
            if (kk * 64 < C && i < n && !(flag[i] & 1)) {
                const unsigned long long key = pcl_okey(sc[i]);
                keys[kk] = key;
                ++cnt;
                kmn = min(kmn, key);
                kmx = max(kmx, key);
                atomicOr(&hist256[(unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 56)], 1u);
            }
        }
        cnt = pcl_wave_sum(cnt);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmn = min(kmn, (unsigned long long)__shfl_xor((long long)kmn, o, 64));
            kmx = max(kmx, (unsigned long long)__shfl_xor((long long)kmx, o, 64));
        }
        if (lane == 0) {
            wsum[0][wave] = cnt;
            red_u[0][wave] = kmn;
            red_u[1][wave] = kmx;
        }
        __syncthreads();
        const int bins = __syncthreads_count(tid < 256 && hist256[tid] != 0u);
        int n_old = 0;
        unsigned long long kmin = ~0ull, kmax = 0ull;
        for (int w = 0; w < NWV; ++w) {
            n_old += wsum[0][w];
            kmin = min(kmin, red_u[0][w]);
            kmax = max(kmax, red_u[1][w]);
        }
        const int m = (int)((double)n_old * (1.0 - a.beam));                   // int(width * (1 - beam))
        bool prune = m > 0 && n_old >= a.min_distinct;
        if (prune && bins < a.min_distinct) {                                   // few bins: count the distinct scores exactly
            unsigned long long prev = 0ull;
            bool have_prev = false;
            int distinct = 0;
            for (int round = 0; round < a.min_distinct; ++round) {              // the next larger key, min_distinct times
                unsigned long long mn = ~0ull;
                bool any = false;
#pragma unroll
                for (int kk = 0; kk < KMAX; ++kk) {
                    const unsigned long long key = keys[kk];
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
                    red_i[0][wave] = any;
                }
                __syncthreads();
                unsigned long long g = ~0ull;
                bool gany = false;
                for (int w = 0; w < NWV; ++w)
                    if (red_i[0][w] && (!gany || red_u[0][w] < g)) {
                        g = red_u[0][w];
                        gany = true;
                    }
                if (!gany) break;
                prev = g;
                have_prev = true;
                ++distinct;
            }
            prune = distinct >= a.min_distinct;
        }
        if (prune) {
            unsigned long long sel = kmin;
            int rank = m - 1;
            const unsigned long long diff = kmin ^ kmax;
            if (diff != 0ull) {
                const int b0 = (63 - __clzll((long long)diff)) >> 3;           // the first byte in which the keys differ
                unsigned long long prefix = (b0 == 7) ? 0ull : (kmin & (~0ull << (8 * (b0 + 1))));
                for (int byte = b0; byte >= 0; --byte) {
                    if (tid < 256) hist256[tid] = 0u;
                    __syncthreads();
                    const unsigned long long hi_mask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
#pragma unroll
                    for (int kk = 0; kk < KMAX; ++kk) {
                        const unsigned long long key = keys[kk];
                        if (key != NOKEY && (key & hi_mask) == prefix) atomicAdd(&hist256[(unsigned int)(key >> (8 * byte)) & 255u], 1u);
                    }
                    __syncthreads();
                    if (wave == 0) {                                            // 4 bins per lane: the bin holding rank
                        int h[4], s4 = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            h[j] = (int)hist256[4 * lane + j];
                            s4 += h[j];
                        }
                        int inc = s4;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const int x = __shfl_up(inc, o, 64);
                            if (lane >= o) inc += x;
                        }
                        const unsigned long long hit = __ballot(inc > rank);
                        if (hit != 0ull && lane == __ffsll((long long)hit) - 1) {
                            int acc = inc - s4, j = 0;
                            for (; j < 3; ++j) {
                                if (acc + h[j] > rank) break;
                                acc += h[j];
                            }
                            s_sel = (unsigned long long)(4 * lane + j);
                            s_i[3] = rank - acc;
                        }
                    }
                    __syncthreads();
                    prefix |= s_sel << (8 * byte);
                    rank = s_i[3];
                }
                sel = prefix;
            }
            // everything below the selected key goes, and of the tokens equal to it the first (rank + 1) in token order
            int eq = 0;
#pragma unroll
            for (int kk = 0; kk < KMAX; ++kk) eq += keys[kk] == sel;
            eq = pcl_wave_sum(eq);
            __syncthreads();
            if (lane == 0) wsum[0][wave] = eq;
            __syncthreads();
            int run = 0;
            for (int w = 0; w < wave; ++w) run += wsum[0][w];
#pragma unroll
            for (int kk = 0; kk < KMAX; ++kk) {
                const unsigned long long key = keys[kk];
                const bool is_eq = key == sel;                                 // (sel is a real key, never NOKEY)
                const unsigned long long mask = __ballot(is_eq);
                if (key != NOKEY && (key < sel || (is_eq && run + __popcll(mask & lt_mask) <= rank))) flag[w0 + kk * 64 + lane] |= 2;
                run += __popcll(mask);
            }
        }
        __syncthreads();
        STAMP(4)
        // ---- (5) stable compaction: the survivors of the old tokens, then the new ones; the node -> token map follows
        double *scn = scb[cur ^ 1];
        int *ndn = ndb[cur ^ 1], *hsn = hsb[cur ^ 1], *upn = upb[cur ^ 1];
        int keep_cnt = 0;
        for (int k = 0; k < C; k += 64) {
            const int i = w0 + k + lane;
            if (i < n) {
                if (flag[i] & 3) {
                    const int node = nd[i];
                    if (slot[node] == i) slot[node] = -1;
                } else {
                    ++keep_cnt;
                }
            }
        }
        keep_cnt = pcl_wave_sum(keep_cnt);
        if (lane == 0) wsum[1][wave] = keep_cnt;
        __syncthreads();
        int krun = 0, n_keep = 0;
        for (int w = 0; w < NWV; ++w) {
            const int x = wsum[1][w];
            if (w < wave) krun += x;
            n_keep += x;
        }
        for (int k = 0; k < C; k += 64) {
            const int i = w0 + k + lane;
            const bool keep = i < n && !(flag[i] & 3);
            const unsigned long long mask = __ballot(keep);
            {
                if (keep) {
                    const int to = krun + __popcll(mask & lt_mask);
                    const int node = nd[i];
                    scn[to] = sc[i];
                    ndn[to] = node;
                    hsn[to] = hs[i];
                    upn[to] = up[i];
                    src[to] = i;                                   // where the token's p sits in this frame's p buffer
                    slot[node] = to;
                }
            }
            krun += __popcll(mask);
        }
        for (int j = tid; j < n_new; j += DW) {
            const int i = n + j, to = n_keep + j, node = nd[i];
            scn[to] = sc[i];
            ndn[to] = node;
            hsn[to] = hs[i];
            upn[to] = up[i];
            src[to] = i;
            slot[node] = to;
        }
        __syncthreads();
        STAMP(5)
        n = n_keep + n_new;
        cur ^= 1;
        pcur ^= 1;
        if (tid == 0) a.trace[(size_t)u * a.Tmax + t] = n;
    }
    // ---- transfer (Decoder.py:175-187): the `candidate` best tokens, ties in token order
    const double *sc = scb[cur];
    for (int i = tid; i < n; i += DW) flag[i] = 0;                 // 4 = taken
    __syncthreads();
    int n_out = 0;
    for (int cc = 0; cc < a.candidate && cc < n; ++cc) {
        double b = -INFINITY;
        int bi = NONE;
        for (int i = tid; i < n; i += DW)
            if (flag[i] != 4 && (bi == NONE || sc[i] > b)) {        // (strictly greater keeps the earliest on ties)
                b = sc[i];
                bi = i;
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
            red_i[0][wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            double g = -INFINITY;
            int gi = NONE;
            for (int w = 0; w < NWV; ++w)
                if (red_i[0][w] != NONE && (gi == NONE || red_d[w] > g || (red_d[w] == g && red_i[0][w] < gi))) {
                    g = red_d[w];
                    gi = red_i[0][w];
                }
            a.out_node[(size_t)u * a.candidate + cc] = ndb[cur][gi];
            a.out_score[(size_t)u * a.candidate + cc] = sc[gi];
            a.out_hist[(size_t)u * a.candidate + cc] = hsb[cur][gi];
            flag[gi] = 4;
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
            for (int k = 0; k < N_STAMP; ++k) a.stamps[k] = st_acc[k];
#endif
    }
}

}  // namespace

void pcl_lexicon_release(pcl_ctx *ctx) {
    dev_free(ctx->lex_units);
    dev_free(ctx->lex_nunits);
    dev_free(ctx->lex_child_ptr);
    dev_free(ctx->lex_child_idx);
    dev_free(ctx->lex_word);
    dev_free(ctx->lex_roots);
    dev_free(ctx->lex_info);
    dev_free(ctx->d_unit_logtrans);
    ctx->lex_nodes = ctx->lex_nroots = 0;
}

void pcl_batch_decode_release(pcl_batch *b) {
    dev_free(b->dec_f64);
    dev_free(b->dec_slot);
    dev_free(b->dec_work);
    dev_free(b->dec_int);
    dev_free(b->dec_score);
    b->dec_cap = b->dec_cand = 0;
}

extern "C" {

int pcl_lexicon_upload(pcl_ctx *ctx, int n_nodes, const int32_t *node_units, const int32_t *node_nunits, const int32_t *child_ptr,
                       const int32_t *child_idx, const int32_t *node_word, int n_roots, const int32_t *roots) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->n_units) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_lexicon_upload: pcl_units_upload first");
    if (n_nodes <= 0 || n_roots <= 0 || !node_units || !node_nunits || !child_ptr || !node_word || !roots)
        PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: bad arguments (n_nodes=%d, n_roots=%d)", n_nodes, n_roots);
    const int e = ctx->S - 2;
    if (child_ptr[0] != 0) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: child_ptr[0] = %d, must be 0", child_ptr[0]);
    for (int i = 0; i < n_nodes; ++i) {
        const int nu = node_nunits[i];
        if (ctx->n_units >= 0xffff) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: %d units (the decoder packs unit ids into 16 bits)", ctx->n_units);
        if (nu < 1 || nu > 2 || e * nu + 2 > NS) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: node %d has %d units (1 or 2; %d-state units need %d <= %d HMM states)", i, nu, ctx->S, e * nu + 2, NS);
        for (int k = 0; k < nu; ++k)
            if (node_units[2 * i + k] < 0 || node_units[2 * i + k] >= ctx->n_units) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: node %d unit %d outside [0,%d)", i, node_units[2 * i + k], ctx->n_units);
        if (child_ptr[i + 1] < child_ptr[i]) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: child_ptr is not monotone at node %d", i);
    }
    const int nc = child_ptr[n_nodes];
    if (nc > 0 && !child_idx) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: child_idx is NULL");
    std::vector<char> has_parent(n_nodes, 0);
    for (int k = 0; k < nc; ++k) {
        if (child_idx[k] < 0 || child_idx[k] >= n_nodes) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: child index %d outside [0,%d)", child_idx[k], n_nodes);
        if (has_parent[child_idx[k]]) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: node %d has two parents (not a tree)", child_idx[k]);
        has_parent[child_idx[k]] = 1;
    }
    for (int r = 0; r < n_roots; ++r) {
        if (roots[r] < 0 || roots[r] >= n_nodes || has_parent[roots[r]] == 1) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: root %d is not a parentless node", roots[r]);
        if (has_parent[roots[r]] == 2) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: root %d is listed twice", roots[r]);
        has_parent[roots[r]] = 2;                    // (seen as a root)
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_lexicon_release(ctx);
    TRY(dev_alloc(ctx, &ctx->lex_units, (size_t)2 * n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_nunits, (size_t)n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_child_ptr, (size_t)n_nodes + 1));
    TRY(dev_alloc(ctx, &ctx->lex_child_idx, (size_t)std::max(nc, 1)));
    TRY(dev_alloc(ctx, &ctx->lex_word, (size_t)n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_roots, (size_t)n_roots));
    TRY(dev_alloc(ctx, &ctx->lex_info, (size_t)n_nodes));
    {   // (first child, children, words end here, unit pair) of a node in one 16-byte record
        std::vector<int4> info(n_nodes);
        for (int i = 0; i < n_nodes; ++i) {
            const int u0 = node_units[2 * i], u1 = node_nunits[i] == 2 ? node_units[2 * i + 1] : 0xffff;
            info[i] = make_int4(child_ptr[i], child_ptr[i + 1] - child_ptr[i], node_word[i] ? 1 : 0, u0 | (u1 << 16));
        }
        HIPCHK(ctx, hipMemcpy(ctx->lex_info, info.data(), (size_t)n_nodes * sizeof(int4), hipMemcpyHostToDevice));
    }
    TRY(dev_alloc(ctx, &ctx->d_unit_logtrans, ctx->unit_logtrans.size()));
    HIPCHK(ctx, hipMemcpy(ctx->lex_units, node_units, (size_t)2 * n_nodes * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lex_nunits, node_nunits, (size_t)n_nodes * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lex_child_ptr, child_ptr, ((size_t)n_nodes + 1) * 4, hipMemcpyHostToDevice));
    if (nc) HIPCHK(ctx, hipMemcpy(ctx->lex_child_idx, child_idx, (size_t)nc * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lex_word, node_word, (size_t)n_nodes * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lex_roots, roots, (size_t)n_roots * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_unit_logtrans, ctx->unit_logtrans.data(), ctx->unit_logtrans.size() * 8, hipMemcpyHostToDevice));
    ctx->lex_nodes = n_nodes;
    ctx->lex_nroots = n_roots;
    return PCL_OK;
}

int pcl_batch_decode(pcl_batch *b, double beam, int min_distinct, int candidate, int max_tokens, double logpi_one_unit, double logpi_two_units) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!ctx->lex_nodes) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode: pcl_lexicon_upload first");
    if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode: no emissions (pcl_batch_score first)");
    if (!(beam > 0.0 && beam <= 1.0) || min_distinct < 1 || candidate < 1 || max_tokens < 1)
        PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_decode: beam %g, min_distinct %d, candidate %d, max_tokens %d", beam, min_distinct, candidate, max_tokens);
    if (max_tokens > 16 * DW) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_decode: max_tokens %d > %d (a lane keeps the sort keys of its tokens in registers)", max_tokens, 16 * DW);
    // the emission rows must be [entry, state 0 .. J-1, exit]: the all-state matrix
    const int J = ctx->n_units * (ctx->S - 2);
    for (int u = 0; u < b->U; ++u) {
        const UttDesc &d = b->utt[u];
        bool ok = d.N == J + 2 && (int)b->row_state.size() >= d.vec_off + d.N;
        for (int n = 0; ok && n < d.N; ++n) ok = b->row_state[d.vec_off + n] == (n == 0 ? PCL_ROW_ENTRY : n == d.N - 1 ? PCL_ROW_EXIT : n - 1);
        if (!ok) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode: utterance %d is not an all-state batch (rows entry, 0..%d, exit)", u, J - 1);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (b->dp_pending) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b->ev_dp, 0));
        b->dp_pending = false;
    }
    // Like the forward-backward recursion, the decoder does no matrix work and is latency / bandwidth bound: it runs on
    // the second stream, beside the scoring of the next chunk on the main one (every later call on this batch joins it).
    hipStream_t main_stream = ctx->stream;
    if (ctx->dp_async) {
        if (!b->ev_dp) HIPCHK(ctx, hipEventCreateWithFlags(&b->ev_dp, hipEventDisableTiming));
        HIPCHK(ctx, pcl_dp_follows_main(b));
    }
    struct StreamSwap {                                                        // the launches below and their timer use ctx->stream
        pcl_ctx *c;
        hipStream_t keep;
        ~StreamSwap() { c->stream = keep; }
    } swap_back{ctx, main_stream};
    if (ctx->dp_async) ctx->stream = ctx->stream_dp;
    const int cap = max_tokens, U = b->U, Tm = b->Tmax;
    // left-to-right units (every model the reference builds): one lane per token, hmm_decode_lr.hip; PCL_DEC_GENERAL=1 keeps
    // the general kernel (the parity tests run both against the restatement)
    const char *force_general = getenv("PCL_DEC_GENERAL");
    const bool use_lr = !(force_general && atoi(force_general)) && pcl_decode_lr_applicable(ctx, J + 2, cap, Tm);
    if (b->dec_cap != cap || b->dec_cand != candidate || b->dec_nodes != ctx->lex_nodes) {
        pcl_batch_decode_release(b);
        // doubles per utterance: score 2 cap | p 2 cap NS | seg_score cap + 2;  ints: node, hist, upair 2 cap each | flag, dst cap each |
        // seg_ofs, seg_cptr, seg_hist cap + 2 each (general kernel); the left-to-right kernel lays meta 8 cap | src 2 cap over the same words
        TRY(dev_alloc(ctx, &b->dec_f64, (size_t)U * (2 * (size_t)cap * (1 + NS) + cap + 2)));
        TRY(dev_alloc(ctx, &b->dec_work, (size_t)U * (10 * (size_t)cap + 3 * ((size_t)cap + 2))));
        TRY(dev_alloc(ctx, &b->dec_slot, (size_t)U * ctx->lex_nodes));
        // ints: out_n U | out_node U*cand | out_hist U*cand | hist_n U | hist_prev U*Tm | hist_node U*Tm | trace U*Tm | overflow U
        TRY(dev_alloc(ctx, &b->dec_int, (size_t)U * (3 + 2 * candidate + 3 * Tm)));
        TRY(dev_alloc(ctx, &b->dec_score, (size_t)U * candidate));
        b->dec_cap = cap;
        b->dec_cand = candidate;
        b->dec_nodes = ctx->lex_nodes;
    }
    HIPCHK(ctx, hipMemsetAsync(b->dec_slot, 0xff, (size_t)U * ctx->lex_nodes * sizeof(int), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(b->dec_int, 0, (size_t)U * (3 + 2 * candidate + 3 * Tm) * sizeof(int), ctx->stream));
    DecArgs a;
    a.utts = b->d_utt;
    a.Bt = b->Bt;
    a.unit_logtrans = ctx->d_unit_logtrans;
    a.node_units = ctx->lex_units; a.node_nunits = ctx->lex_nunits; a.child_ptr = ctx->lex_child_ptr; a.child_idx = ctx->lex_child_idx;
    a.node_word = ctx->lex_word; a.roots = ctx->lex_roots; a.node_info = ctx->lex_info;
    a.n_nodes = ctx->lex_nodes; a.n_roots = ctx->lex_nroots; a.n_units = ctx->n_units; a.S = ctx->S; a.cap = cap; a.candidate = candidate;
    a.min_distinct = min_distinct; a.Tmax = Tm;
    a.beam = beam; a.lpi1 = logpi_one_unit; a.lpi2 = logpi_two_units;
    {
        double *q = b->dec_f64;
        a.score = q; q += (size_t)U * 2 * cap;
        a.p = q; q += (size_t)U * 2 * cap * NS;
        a.seg_score = q;
        int *w = b->dec_work;
        a.meta = (int4 *)w;                          // (left-to-right kernel: 8 cap ints per utterance)
        a.node = w; w += (size_t)U * 2 * cap;
        a.hist = w; w += (size_t)U * 2 * cap;
        a.upair = w; w += (size_t)U * 2 * cap;
        a.flag = w; w += (size_t)U * cap;
        a.dst = w; w += (size_t)U * cap;
        if (use_lr) {                                // node, hist, upair [U][2][cap] each (as above) | 2 U cap spare | src [U][2][cap] | the seg_* arrays
            a.dst = b->dec_work + (size_t)U * 8 * cap;
            a.flag = nullptr;
            w = a.dst + (size_t)U * 2 * cap;
        }
        a.seg_ofs = w; w += (size_t)U * (cap + 2);
        a.seg_cptr = w; w += (size_t)U * (cap + 2);
        a.seg_hist = w;
    }
    a.slot = b->dec_slot;
    int *p = b->dec_int;
    a.out_n = p; p += U;
    a.out_node = p; p += (size_t)U * candidate;
    a.out_hist = p; p += (size_t)U * candidate;
    a.hist_n = p; p += U;
    a.hist_prev = p; p += (size_t)U * Tm;
    a.hist_node = p; p += (size_t)U * Tm;
    a.trace = p; p += (size_t)U * Tm;
    a.overflow = p;
    a.out_score = b->dec_score;
    a.stamps = nullptr;
#ifdef PCL_DEC_STAMPS
    long long *d_stamps = nullptr;
    TRY(dev_alloc(ctx, &d_stamps, (size_t)N_STAMP));
    HIPCHK(ctx, hipMemsetAsync(d_stamps, 0, N_STAMP * sizeof(long long), ctx->stream));
    a.stamps = d_stamps;
#endif
    if (use_lr) {
        pcl_timer_begin(ctx, "decode");
        const int rc = pcl_decode_lr_launch(ctx, a, U, J + 2);
        pcl_timer_end(ctx, "decode");
        if (rc != PCL_OK) return rc;
        HIPCHK(ctx, hipGetLastError());
    } else {
    // the unit matrices and one emission row in LDS when they fit beside the kernel's static 18 KB
    const size_t table_bytes = ((size_t)ctx->n_units * ctx->S * ctx->S + (size_t)(J + 2)) * sizeof(double);
    const bool tlds = table_bytes <= 44u * 1024u;
    pcl_timer_begin(ctx, "decode");
#define PCL_DEC_LAUNCH(K)                                                                                                  \
    do {                                                                                                                   \
        if (tlds) hipLaunchKernelGGL((hmm_decode_kernel<true, K>), dim3(U), dim3(DW), table_bytes, ctx->stream, a);       \
        else hipLaunchKernelGGL((hmm_decode_kernel<false, K>), dim3(U), dim3(DW), 0, ctx->stream, a);                      \
    } while (0)
    if (cap <= DW) PCL_DEC_LAUNCH(1);
    else if (cap <= 2 * DW) PCL_DEC_LAUNCH(2);
    else if (cap <= 4 * DW) PCL_DEC_LAUNCH(4);
    else if (cap <= 8 * DW) PCL_DEC_LAUNCH(8);
    else PCL_DEC_LAUNCH(16);
#undef PCL_DEC_LAUNCH
    pcl_timer_end(ctx, "decode");
    HIPCHK(ctx, hipGetLastError());
    }
#ifdef PCL_DEC_STAMPS
    {
        long long h[N_STAMP];
        HIPCHK(ctx, hipMemcpyAsync(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "decode stamps (utterance 0, 100 MHz ticks): step %lld donors %lld pairs %lld first|keys %lld prune %lld compact %lld select %lld\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
        dev_free(d_stamps);
    }
#endif
    if (ctx->dp_async) {
        HIPCHK(ctx, hipEventRecord(b->ev_dp, ctx->stream_dp));
        b->dp_pending = true;
    } else {
        HIPCHK(ctx, pcl_batch_mark(b));
    }
    b->have_dec = true;
    return PCL_OK;
}

int pcl_batch_decode_get(pcl_batch *b, int32_t *n_final, int32_t *node, double *score, int32_t *hist, int32_t *hist_n, int32_t *hist_prev,
                         int32_t *hist_node, int32_t *n_tokens, int32_t *overflow) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!b->have_dec) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode_get: run pcl_batch_decode first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int U = b->U, c = b->dec_cand, Tm = b->Tmax;
    const int *p = b->dec_int;
    // the results come down on the stream the decoder ran on, and only that stream is waited for: the scoring of the next
    // chunk, queued on the main stream meanwhile, keeps running
    hipStream_t st = b->dp_pending ? ctx->stream_dp : ctx->stream;
    auto get = [&](void *dst, const void *src, size_t bytes) -> int {
        if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        return PCL_OK;
    };
    TRY(get(n_final, p, (size_t)U * 4)); p += U;
    TRY(get(node, p, (size_t)U * c * 4)); p += (size_t)U * c;
    TRY(get(hist, p, (size_t)U * c * 4)); p += (size_t)U * c;
    TRY(get(hist_n, p, (size_t)U * 4)); p += U;
    TRY(get(hist_prev, p, (size_t)U * Tm * 4)); p += (size_t)U * Tm;
    TRY(get(hist_node, p, (size_t)U * Tm * 4)); p += (size_t)U * Tm;
    TRY(get(n_tokens, p, (size_t)U * Tm * 4)); p += (size_t)U * Tm;
    TRY(get(overflow, p, (size_t)U * 4));
    TRY(get(score, b->dec_score, (size_t)U * c * 8));
    HIPCHK(ctx, hipStreamSynchronize(st));
    b->dp_pending = false;                                                     // (complete: nothing left to join)
    return PCL_OK;
}

}  // extern "C"

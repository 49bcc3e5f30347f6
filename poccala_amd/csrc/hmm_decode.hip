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
// spelled out next to the CPU restatement the parity tests hold this kernel to, bit for bit (include/poccala_hip.h names it).  PARITY UNPINNED
// against the reference itself -- nothing executable exists there.
//
// Mapping.  The emissions are the all-state matrix of a scoring batch (rows entry, 0..J-1, exit: pcl_batch_score), time
// major, so a frame's J values are contiguous.  A token is 8 lanes (N = two units x three emitting states + 2 <= 8):
// lane j holds p_j; ln A is built on the fly from the unit matrices (183 units x 25 doubles: cache resident).  All
// float64: scores reach -1e5.  Per frame the workgroup runs
//   step -> donors (finished tokens) and the best word-end donor -> prefix sum of the creations -> merge / create ->
//   first step of the new tokens -> prune (radix select on the order-preserving bits of the scores, ties by token
//   order) -> stable compaction into the other token buffer,
// with workgroup barriers between phases; a tree node has at most one live token and one parent, so no hand-over needs
// an atomic and the result does not depend on timing.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pcl_internal.h"

namespace {

constexpr int DW = 1024;            // threads per workgroup
constexpr int NS = 8;               // lanes per token = states of its HMM at most

struct DecTok {
    double score;
    double p[NS];
    int node, hist, fin, drop;      // fin: finished this frame; drop: pruned this frame
};

struct DecArgs {
    const UttDesc *utts;
    const double *Bt;
    const double *unit_logtrans;    // [n_units][S][S]
    const int *node_units, *node_nunits, *child_ptr, *child_idx, *node_word, *roots;
    int n_nodes, n_roots, S, cap, candidate, min_distinct, Tmax;
    double beam, lpi1, lpi2;        // ln(1/N) for one- and two-unit nodes, from the caller's np.log
    DecTok *tok;                    // [U][2][cap]
    int *slot;                      // [U][n_nodes]: live token of a node, or -1
    int *work;                      // [U][cap + n_roots]: creation counts / prefix sums
    int *out_n, *out_node, *out_hist, *hist_n, *hist_prev, *hist_node, *trace, *overflow;
    double *out_score;
};

__device__ __forceinline__ unsigned long long okey(double s) {       // order-preserving bits
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ln A[i][j] of the node's embedded HMM (AcousticModel.py:979-989)
__device__ __forceinline__ double log_a(const double *lt, const int *units, int nu, int S, int i, int j) {
    const int e = S - 2, N = e * nu + 2;
    if (i >= N - 1) return -INFINITY;
    const int pos = (i == 0) ? 0 : (i - 1) / e, r = (i == 0) ? 0 : 1 + (i - 1) % e, c = j - pos * e;
    if (c < 0 || c >= S) return -INFINITY;
    return lt[((size_t)units[pos] * S + r) * S + c];
}

// One step of one token on the frame whose emissions start at Bf, by the token's 8 lanes (sub = state j; the 8 lanes sit
// in one wave and run in lockstep: every lane has read the old p before any lane stores the new one).
// first: p = ln pi + B[:,t] (Decoder.py:270); else the max recursion (:278-283).  score += max_j p_j (:285); fin (D1).
__device__ __forceinline__ void token_step(const DecArgs &a, DecTok *tk, const double *Bf, int sub, bool first) {
    const int *units = a.node_units + (size_t)tk->node * 2;
    const int nu = a.node_nunits[tk->node], e = a.S - 2, N = e * nu + 2;
    double bj = -INFINITY;
    if (sub == 0) bj = 0.0;                                        // entry VirtualState: ln 1 (AcousticModel.py:218)
    else if (sub < N - 1) bj = Bf[1 + units[(sub - 1) / e] * e + (sub - 1) % e];   // (exit VirtualState: ln 0, :219)
    double pj = -INFINITY;
    if (first) {
        if (sub < N) pj = (nu == 1 ? a.lpi1 : a.lpi2) + bj;
    } else {
        const double pold = tk->p[sub];                            // every lane's old value travels by shuffle: no lane reads
        double m = -INFINITY;                                      // p from memory after another lane has stored its new one
        for (int i = 0; i < N; ++i) {
            const double pi = __shfl(pold, i, NS);
            if (sub < N) m = fmax(m, pi + log_a(a.unit_logtrans, units, nu, a.S, i, sub));
        }
        if (sub < N) pj = m + bj;
    }
    double best = pj;
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
    tk->p[sub] = pj;
    if (sub == 0) {
        tk->score += best;
        tk->fin = arg >= N - 2;
        tk->drop = 0;
    }
}

// exclusive prefix sum of v[0..n) in place (workgroup-wide); returns the total.  sh: DW + 1 ints of LDS.
__device__ int block_scan(int *v, int n, int *sh) {
    const int tid = threadIdx.x, per = (n + DW - 1) / DW, lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += v[i];
    sh[tid] = s;
    __syncthreads();
    if (tid < 64) {                                                // one wave scans the DW partials, 16 per lane
        constexpr int PER = DW / 64;
        int loc[PER], run = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            loc[k] = run;
            run += sh[tid * PER + k];
        }
        int inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int w = __shfl_up(inc, o, 64);
            if (tid >= o) inc += w;
        }
        const int base = inc - run;
#pragma unroll
        for (int k = 0; k < PER; ++k) sh[tid * PER + k] = base + loc[k];
        if (tid == 63) sh[DW] = inc;
    }
    __syncthreads();
    int run = sh[tid];
    for (int i = lo; i < hi; ++i) {
        const int x = v[i];
        v[i] = run;
        run += x;
    }
    const int total = sh[DW];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(DW) void hmm_decode_kernel(DecArgs a) {
    __shared__ int sh[DW + 1];
    __shared__ unsigned int hist256[256];
    __shared__ double red_d[DW / 64];
    __shared__ int red_i[DW / 64];
    __shared__ unsigned long long s_sel;
    __shared__ int s_i[4];
    __shared__ double s_d[2];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const UttDesc d = a.utts[u];
    const int T = d.T, Nb = d.N, cap = a.cap;
    const double *B = a.Bt + d.b_off;
    DecTok *buf[2] = {a.tok + (size_t)u * 2 * cap, a.tok + ((size_t)u * 2 + 1) * cap};
    int *slot = a.slot + (size_t)u * a.n_nodes;
    int *work = a.work + (size_t)u * (cap + a.n_roots);
    int *hprev = a.hist_prev + (size_t)u * a.Tmax, *hnode = a.hist_node + (size_t)u * a.Tmax;
    const int tk8 = tid >> 3, sub = tid & 7;                       // token group of 8 lanes

    // ---- frame 0: every first-character node starts (D3)
    int cur = 0, n = min(a.n_roots, cap), ovf = a.n_roots > cap, nh = 0;
    for (int i0 = 0; i0 < n; i0 += DW / NS) {
        const int i = i0 + tk8;
        if (i < n) {
            DecTok *tk = &buf[0][i];
            if (sub == 0) {
                tk->node = a.roots[i];
                tk->hist = -1;
                tk->score = 0.0;
                slot[a.roots[i]] = i;
            }
        }
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += DW / NS) {
        const int i = i0 + tk8;
        if (i < n) token_step(a, &buf[0][i], B, sub, true);
    }
    __syncthreads();
    if (tid == 0) a.trace[(size_t)u * a.Tmax] = n;

    for (int t = 1; t < T; ++t) {
        DecTok *tok = buf[cur], *nxt = buf[cur ^ 1];
        const double *Bf = B + (size_t)t * Nb;
        // ---- (1) every live token takes its step
        for (int i0 = 0; i0 < n; i0 += DW / NS) {
            const int i = i0 + tk8;
            if (i < n) token_step(a, &tok[i], Bf, sub, false);
        }
        __syncthreads();
        // ---- (2) donors.  The best finished word-end token (earliest on ties) re-seeds the first characters (D4);
        //      first_w = the first finished word-end token: the roots it makes are created right after its children
        double bw = -INFINITY;
        int bw_i = 0x7fffffff, fw = 0x7fffffff;
        for (int i = tid; i < n; i += DW)
            if (tok[i].fin && a.node_word[tok[i].node]) {
                fw = min(fw, i);
                if (tok[i].score > bw || (tok[i].score == bw && i < bw_i) || bw_i == 0x7fffffff) {
                    bw = tok[i].score;
                    bw_i = i;
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(bw, o, 64);
            const int oi = __shfl_xor(bw_i, o, 64);
            if (oi != 0x7fffffff && (bw_i == 0x7fffffff || ob > bw || (ob == bw && oi < bw_i))) {
                bw = ob;
                bw_i = oi;
            }
            fw = min(fw, __shfl_xor(fw, o, 64));
        }
        if (lane == 0) {
            red_d[wave] = bw;
            red_i[wave] = bw_i;
            sh[wave] = fw;
        }
        __syncthreads();
        if (tid == 0) {
            double b = -INFINITY;
            int bi = 0x7fffffff, f = 0x7fffffff;
            for (int w = 0; w < DW / 64; ++w) {
                if (red_i[w] != 0x7fffffff && (bi == 0x7fffffff || red_d[w] > b || (red_d[w] == b && red_i[w] < bi))) {
                    b = red_d[w];
                    bi = red_i[w];
                }
                f = min(f, sh[w]);
            }
            s_d[0] = b;
            s_i[0] = bi;
            s_i[1] = f;
            s_i[2] = -1;
            if (bi != 0x7fffffff) {                                // one history entry per frame: the winning donor's word
                if (nh < a.Tmax) {
                    hprev[nh] = tok[bi].hist;
                    hnode[nh] = tok[bi].node;
                }
                s_i[2] = nh;
            }
        }
        __syncthreads();
        const double w_score = s_d[0];
        const int w_i = s_i[0], first_w = s_i[1], w_hist = s_i[2];
        if (w_i != 0x7fffffff) ++nh;
        // a target is "live" when its node has a token that did not finish in this frame
        auto live_slot = [&](int node) -> int {
            const int s = slot[node];
            return (s >= 0 && !tok[s].fin) ? s : -1;
        };
        // creation counts: work[i] for donor i = its children without a live token; work[n + r] for root r (only if a
        // word ended), laid out so that the prefix sum gives the reference's creation order: donors in token order, each
        // donor's children in child order, the roots right after the children of the first word-end donor
        for (int i = tid; i < n; i += DW) {
            int c = 0;
            if (tok[i].fin) {
                const int nd = tok[i].node;
                for (int k = a.child_ptr[nd]; k < a.child_ptr[nd + 1]; ++k) c += live_slot(a.child_idx[k]) < 0;
            }
            work[i] = c;
        }
        for (int r = tid; r < a.n_roots; r += DW) work[n + r] = (w_i != 0x7fffffff) ? (live_slot(a.roots[r]) < 0) : 0;
        __syncthreads();
        const int n_child_new = block_scan(work, n, sh);
        const int n_root_new = block_scan(work + n, a.n_roots, sh);
        // position of donor i's first creation: its prefix, plus the roots if the first word-end donor comes before it
        // ---- (3) merges and creations
        for (int i = tid; i < n; i += DW) {
            if (!tok[i].fin) continue;
            const int nd = tok[i].node;
            int pos = n + work[i] + ((first_w < i) ? n_root_new : 0);
            for (int k = a.child_ptr[nd]; k < a.child_ptr[nd + 1]; ++k) {      // passing_in_word (Decoder.py:114-143)
                const int c = a.child_idx[k], s = live_slot(c);
                if (s >= 0) {
                    if (tok[i].score > tok[s].score) {                          // :126-134: the recursion state is kept
                        tok[s].score = tok[i].score;
                        tok[s].hist = tok[i].hist;
                    }
                } else {
                    if (pos < cap) {
                        DecTok *nt = &tok[pos];                                 // (slots n .. cap-1 of the current buffer)
                        nt->node = c;
                        nt->hist = tok[i].hist;
                        nt->score = tok[i].score;
                    }
                    ++pos;
                }
            }
        }
        if (w_i != 0x7fffffff) {
            const int base = n + work[first_w] + ((first_w + 1 < n) ? (work[first_w + 1] - work[first_w]) : (n_child_new - work[first_w]));
            for (int r = tid; r < a.n_roots; r += DW) {
                const int node = a.roots[r], s = live_slot(node);
                if (s >= 0) {
                    if (w_score > tok[s].score) {
                        tok[s].score = w_score;
                        tok[s].hist = w_hist;
                    }
                } else {
                    const int pos = base + work[n + r];
                    if (pos < cap) {
                        DecTok *nt = &tok[pos];
                        nt->node = node;
                        nt->hist = w_hist;
                        nt->score = w_score;
                    }
                }
            }
        }
        const int n_new_all = n_child_new + n_root_new;
        const int n_new = min(n_new_all, cap - n);
        if (n_new_all > cap - n) ovf = 1;
        __syncthreads();
        // the new tokens take their first step at once (Decoder.py:138-139)
        for (int i0 = 0; i0 < n_new; i0 += DW / NS) {
            const int i = n + i0 + tk8;
            if (i < n + n_new) token_step(a, &tok[i], Bf, sub, true);
        }
        // ---- (4) pruning over the tokens that were alive before the frame and did not finish (Decoder.py:159-167)
        //      n_old, the number of distinct scores (up to min_distinct), then the m-th smallest by radix select
        int cnt = 0;
        for (int i = tid; i < n; i += DW) cnt += !tok[i].fin;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) red_i[wave] = cnt;
        __syncthreads();
        int n_old = 0;
        for (int w = 0; w < DW / 64; ++w) n_old += red_i[w];
        __syncthreads();
        const int m = (int)((double)n_old * (1.0 - a.beam));                   // int(width * (1 - beam))
        bool prune = m > 0 && n_old >= a.min_distinct;
        if (prune) {                                                            // at least min_distinct different scores?
            unsigned long long prev = 0ull;
            bool have_prev = false;
            int distinct = 0;
            for (int round = 0; round < a.min_distinct; ++round) {
                unsigned long long mn = ~0ull;
                bool any = false;
                for (int i = tid; i < n; i += DW)
                    if (!tok[i].fin) {
                        const unsigned long long k = okey(tok[i].score);
                        if ((!have_prev || k > prev) && (!any || k < mn)) {
                            mn = k;
                            any = true;
                        }
                    }
                // reduce (min over the lanes that found something)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long om = __shfl_xor(mn, o, 64);
                    const int oa = __shfl_xor((int)any, o, 64);
                    if (oa && (!any || om < mn)) {
                        mn = om;
                        any = true;
                    }
                }
                if (lane == 0) {
                    reinterpret_cast<unsigned long long *>(red_d)[wave] = mn;
                    red_i[wave] = any;
                }
                __syncthreads();
                unsigned long long g = ~0ull;
                bool gany = false;
                for (int w = 0; w < DW / 64; ++w)
                    if (red_i[w] && (!gany || reinterpret_cast<unsigned long long *>(red_d)[w] < g)) {
                        g = reinterpret_cast<unsigned long long *>(red_d)[w];
                        gany = true;
                    }
                __syncthreads();
                if (!gany) break;
                prev = g;
                have_prev = true;
                ++distinct;
            }
            prune = distinct >= a.min_distinct;
        }
        if (prune) {
            // radix select, most significant byte first: the key of rank m-1 (0-based) among the old unfinished tokens
            unsigned long long prefix = 0ull;
            int rank = m - 1;
            for (int byte = 7; byte >= 0; --byte) {
                for (int k = tid; k < 256; k += DW) hist256[k] = 0u;
                __syncthreads();
                const unsigned long long hi_mask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
                for (int i = tid; i < n; i += DW)
                    if (!tok[i].fin) {
                        const unsigned long long k = okey(tok[i].score);
                        if ((k & hi_mask) == prefix) atomicAdd(&hist256[(k >> (8 * byte)) & 255u], 1u);
                    }
                __syncthreads();
                if (tid == 0) {
                    int acc = 0, b = 0;
                    for (; b < 256; ++b) {
                        if (acc + (int)hist256[b] > rank) break;
                        acc += hist256[b];
                    }
                    s_sel = (unsigned long long)b;
                    s_i[3] = rank - acc;
                }
                __syncthreads();
                prefix |= s_sel << (8 * byte);
                rank = s_i[3];
                __syncthreads();
            }
            // everything below the selected key goes, and of the tokens equal to it the first (rank + 1) in token order
            const unsigned long long sel = prefix;
            for (int i = tid; i < n; i += DW) work[i] = (!tok[i].fin && okey(tok[i].score) == sel) ? 1 : 0;
            __syncthreads();
            block_scan(work, n, sh);
            for (int i = tid; i < n; i += DW)
                if (!tok[i].fin) {
                    const unsigned long long k = okey(tok[i].score);
                    if (k < sel || (k == sel && work[i] <= rank)) tok[i].drop = 1;
                }
            __syncthreads();
        }
        // ---- (5) stable compaction: the survivors of the old tokens, then the new ones; the node -> token map follows
        for (int i = tid; i < n; i += DW) {
            const int keep = !tok[i].fin && !tok[i].drop;
            work[i] = keep;
            if (!keep && slot[tok[i].node] == i) slot[tok[i].node] = -1;
        }
        __syncthreads();
        const int n_keep = block_scan(work, n, sh);
        for (int i = tid; i < n + n_new; i += DW) {
            int dst = -1;
            if (i < n) {
                if (!tok[i].fin && !tok[i].drop) dst = work[i];
            } else {
                dst = n_keep + (i - n);
            }
            if (dst >= 0) {
                nxt[dst] = tok[i];
                slot[tok[i].node] = dst;
            }
        }
        __syncthreads();
        n = n_keep + n_new;
        cur ^= 1;
        if (tid == 0) a.trace[(size_t)u * a.Tmax + t] = n;
    }
    // ---- transfer (Decoder.py:175-187): the `candidate` best tokens, ties in token order
    DecTok *tok = buf[cur];
    int n_out = 0;
    for (int c = 0; c < a.candidate && c < n; ++c) {
        double b = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += DW)
            if (tok[i].drop != 2 && (bi == 0x7fffffff || tok[i].score > b)) {   // (strictly greater keeps the earliest on ties)
                b = tok[i].score;
                bi = i;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(b, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > b || (ob == b && oi < bi))) {
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
            int gi = 0x7fffffff;
            for (int w = 0; w < DW / 64; ++w)
                if (red_i[w] != 0x7fffffff && (gi == 0x7fffffff || red_d[w] > g || (red_d[w] == g && red_i[w] < gi))) {
                    g = red_d[w];
                    gi = red_i[w];
                }
            a.out_node[(size_t)u * a.candidate + c] = tok[gi].node;
            a.out_score[(size_t)u * a.candidate + c] = tok[gi].score;
            a.out_hist[(size_t)u * a.candidate + c] = tok[gi].hist;
            tok[gi].drop = 2;                                                   // taken
        }
        ++n_out;
        __syncthreads();
    }
    if (tid == 0) {
        a.out_n[u] = n_out;
        a.hist_n[u] = min(nh, a.Tmax);
        a.overflow[u] = ovf;
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
    dev_free(ctx->d_unit_logtrans);
    ctx->lex_nodes = ctx->lex_nroots = 0;
}

void pcl_batch_decode_release(pcl_batch *b) {
    dev_free(b->dec_tok);
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
    for (int i = 0; i < n_nodes; ++i) {
        const int nu = node_nunits[i];
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
    for (int r = 0; r < n_roots; ++r)
        if (roots[r] < 0 || roots[r] >= n_nodes || has_parent[roots[r]]) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_lexicon_upload: root %d is not a parentless node", roots[r]);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_lexicon_release(ctx);
    TRY(dev_alloc(ctx, &ctx->lex_units, (size_t)2 * n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_nunits, (size_t)n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_child_ptr, (size_t)n_nodes + 1));
    TRY(dev_alloc(ctx, &ctx->lex_child_idx, (size_t)std::max(nc, 1)));
    TRY(dev_alloc(ctx, &ctx->lex_word, (size_t)n_nodes));
    TRY(dev_alloc(ctx, &ctx->lex_roots, (size_t)n_roots));
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
    ctx->lex_units_gen = ctx->n_units;
    return PCL_OK;
}

int pcl_batch_decode(pcl_batch *b, double beam, int min_distinct, int candidate, int max_tokens, double logpi_one_unit, double logpi_two_units) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!ctx->lex_nodes) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode: pcl_lexicon_upload first");
    if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_decode: no emissions (pcl_batch_score first)");
    if (!(beam > 0.0 && beam <= 1.0) || min_distinct < 1 || candidate < 1 || max_tokens < 1)
        PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_decode: beam %g, min_distinct %d, candidate %d, max_tokens %d", beam, min_distinct, candidate, max_tokens);
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
    const int cap = max_tokens, U = b->U, Tm = b->Tmax;
    if (b->dec_cap != cap || b->dec_cand != candidate || b->dec_nodes != ctx->lex_nodes) {
        pcl_batch_decode_release(b);
        TRY(dev_alloc(ctx, &b->dec_tok, (size_t)U * 2 * cap * sizeof(DecTok)));
        TRY(dev_alloc(ctx, &b->dec_slot, (size_t)U * ctx->lex_nodes));
        TRY(dev_alloc(ctx, &b->dec_work, (size_t)U * (cap + ctx->lex_nroots)));
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
    a.node_word = ctx->lex_word; a.roots = ctx->lex_roots;
    a.n_nodes = ctx->lex_nodes; a.n_roots = ctx->lex_nroots; a.S = ctx->S; a.cap = cap; a.candidate = candidate; a.min_distinct = min_distinct; a.Tmax = Tm;
    a.beam = beam; a.lpi1 = logpi_one_unit; a.lpi2 = logpi_two_units;
    a.tok = reinterpret_cast<DecTok *>(b->dec_tok);
    a.slot = b->dec_slot;
    a.work = b->dec_work;
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
    pcl_timer_begin(ctx, "decode");
    hipLaunchKernelGGL(hmm_decode_kernel, dim3(U), dim3(DW), 0, ctx->stream, a);
    pcl_timer_end(ctx, "decode");
    HIPCHK(ctx, hipGetLastError());
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
    auto get = [&](void *dst, const void *src, size_t bytes) -> int {
        if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

}  // extern "C"

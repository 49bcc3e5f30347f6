// hmm_decode_args.h -- what the two token-passing kernels (hmm_decode.hip: any unit matrices; hmm_decode_lr.hip: left-to-right
// units, the shape every model of the reference has) share: the launch arguments and two device helpers.
#pragma once
#include "pcl_internal.h"

// Token state is kept as separate arrays (coalesced passes), two buffers of each: a frame ends with a stable compaction
// from one into the other.  upair = the node's units, u0 | u1 << 16 (u1 = 0xffff: a one-unit node).
struct DecArgs {
    const UttDesc *utts;
    const double *Bt;
    const double *unit_logtrans;    // [n_units][S][S]
    const int *node_units, *node_nunits, *child_ptr, *child_idx, *node_word, *roots;
    const int4 *node_info;          // [n_nodes] (first child, children, words end here, upair): one gather instead of four
    int n_nodes, n_roots, n_units, S, cap, candidate, min_distinct, Tmax;
    double beam, lpi1, lpi2;        // ln(1/N) for one- and two-unit nodes, from the caller's np.log
    double *score, *p;              // [U][2][cap], [U][2][cap][8] (general kernel: 8 per token; left-to-right kernel: [6][cap])
    int *node, *hist, *upair;       // [U][2][cap]          (both kernels)
    int4 *meta;                     // (round 3-4: the left-to-right kernel's [U][2][cap] (node, history, upair, -) record; unused since round 5)
    int *flag, *dst;                // [U][cap]: bit 0 finished, bit 1 pruned (this frame); where a token's p sits in the other p buffer
                                    // (left-to-right kernel: dst = src [U][2][cap]: where token i's state sits in the other buffers, | fresh << 31; no flags)
    int *seg_ofs, *seg_cptr, *seg_hist;   // [U][cap + 2]: the frame's donors as segments of the flattened (donor, child) list
    double *seg_score;
    int *slot;                      // [U][n_nodes]: live token of a node, or -1
    int *out_n, *out_node, *out_hist, *hist_n, *hist_prev, *hist_node, *trace, *overflow;
    double *out_score;
    long long *stamps;              // PCL_DEC_STAMPS: clock ticks per phase, utterance 0
};

constexpr int PCL_DEC_N_STAMP = 8;

__device__ __forceinline__ unsigned long long pcl_okey(double s) {       // order-preserving bits of a float64
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ int pcl_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hmm_decode_lr.hip: true when every unit matrix is left-to-right (row 0 reaches state 1 only, an emitting state itself and
// its successor only) and S = 5 -- then the fast kernel gives the general kernel's bits and pcl_decode_lr_launch runs it.
bool pcl_decode_lr_applicable(const pcl_ctx *ctx, int n_rows, int cap, int t_max);
int pcl_decode_lr_launch(pcl_ctx *ctx, const DecArgs &a, int U, int n_rows);

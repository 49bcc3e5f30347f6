// Internal structures shared by the C-ABI host code and the HIP kernels of libpoccala_hip.so.
// Target: gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <map>
#include <string>
#include <vector>

#include "../../include/poccala_hip.h"

// ---------------------------------------------------------------- device-side descriptors
// One scoring segment = one (utterance, emitting row): `len` consecutive frames starting at frame
// row `frame0`, written to out[out0 + t*out_stride].  Segments are sorted by GMM state; `vstart`
// is the segment's offset inside the state's virtual concatenation of frames (H4, state-major).
struct ScoreSeg {
    long long frame0;
    long long out0;
    int len;
    int out_stride;
    int vstart;
    int pad;
};
// An emission row that repeats another row of its utterance (the label names the unit twice): T values at stride N.
struct DupRow {
    long long src, dst;
    int T, N;
};
// One workgroup of the scoring kernel: frames [vstart, vstart+tile) of `state`'s concatenation.
struct ScoreTile {
    int state;
    int seg_lo, seg_hi;  // the state's segments are segs[seg_lo, seg_hi)
    int vstart;
    int seg0;            // the segment that contains vstart: a frame of the tile is almost always in seg0 or seg0 + 1
};
// One sentence HMM (time-major device matrices: element (t, n) at b_off + t*N + n).
struct UttDesc {
    long long b_off;    // into Bt / alpha / beta / lgam
    long long mat_off;  // into the ragged dense (N,N) xi output
    long long frame0;   // first frame row (or -1)
    int T, N;
    int vec_off;   // into ragged (N,) vectors
    int ptr_off;   // into row_ptr / col_ptr (N+1 entries per utterance)
    int nnz_off;   // into CSR / CSC entry arrays
    int path_off;  // into ragged (T,) vectors
};

// One surviving (frame, state) pair of the accumulate pass.
struct ActiveFrame {
    long long frame;  // row of the frame matrix
    double coef;      // ln gamma_t(j) - ln b_j(o_t)
    double lg;        // gamma_t(j) = exp(ln gamma_t(j)), taken once when the list is built (alpha_acc sums it)
};

// ---------------------------------------------------------------- host-side objects
struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

// Scratch of the accumulate pass (gmm_accumulate.hip, gmm_accumulate_f16.hip): everything here is rebuilt by every pass and dead when the
// pass's last kernel has run, and the passes of a context are serialised on its main stream (the auxiliary stream's producers are
// joined by the consumers) -- so it belongs to the CONTEXT, grow-only, not to the batch: rounds 2-4 gave every batch its own (4 GB of
// tile images + up to 442 MB of active-frame lists: 35 GB for the 8 resident batches of config 4, and a corpus sweep that makes a
// batch per step allocated them per step).
struct AccScratch {
    // work lists: per-segment counts / offsets, per-state active-frame lists, the accumulate order of the states
    int *acc_cnt = nullptr;
    long long *acc_off = nullptr;
    ActiveFrame *acc_list = nullptr;
    int *d_work_states = nullptr, *d_seg_lo = nullptr, *d_seg_hi = nullptr, *d_split_flag = nullptr;   // (split_flag: 1 = a split state, in accumulate order)
    size_t acc_cap_list = 0, acc_cap_segs = 0, acc_cap_states = 0;
    // producer / consumer: tile images in LDS order, per-state tile offsets, outlier masks; two sets: the producer of state group g + 1
    // runs on the auxiliary stream beside the consumer of group g
    void *acc16_images[2] = {nullptr, nullptr};
    int *acc16_tile_off[2] = {nullptr, nullptr}, *acc16_state_flag[2] = {nullptr, nullptr};
    unsigned int *acc16_tile_mask[2] = {nullptr, nullptr};
    hipEvent_t acc16_ev_prod[2] = {nullptr, nullptr}, acc16_ev_cons[2] = {nullptr, nullptr}, acc16_ev_start = nullptr;
    size_t acc16_cap_tiles = 0, acc16_cap_states = 0;
};

struct pcl_ctx {
    int device = 0;
    AccScratch acc;
    hipStream_t stream = nullptr;
    std::string err;
    int cus = 0;
    // model (device)
    int J = 0, M = 0, Mpad = 0, D = 0, Dhost = 0, row = 0;  // D = device (padded) feature dimension
    int model_flags = 0;
    float *params32 = nullptr;   // J * Mpad * row : [s_0 c_0 s_1 c_1 ... const2]  (log2 domain)
    double *params64 = nullptr;  // same layout, float64
    float *pm32 = nullptr;       // MFMA scoring layout: [J][Mpad32/32][KS4][64 lanes][4], see gmm_score_mfma.hip
    float *centers32 = nullptr;  // J * D per-state expansion centres c_j
    unsigned short *pm16f = nullptr;  // split-f16 layout with the constants folded into the spare K slot: [J][Mpad32/32][2][KS8f][64 lanes][8], see gmm_score_split.hip
    double *kzero = nullptr;          // [J] K0_j = max_m k'_m of that layout
    float *fscale = nullptr;          // [J][2][KS8f*8] power-of-two feature scales of that layout
    int Mpad32 = 0;              // M rounded up to a multiple of 32
    int layouts_valid = 0;             // PCL_LAYOUT_* derived for the current model (the f64 rows are derived on first use)
    hipStream_t stream_dp = nullptr;   // forward-backward runs here, beside the next batch's scoring on `stream`
    hipStream_t stream_aux = nullptr;  // the accumulate pass's tile-image producer runs here, beside its consumer on `stream`
    hipStream_t stream_desc = nullptr; // descriptor uploads of batches that have nothing in flight (pcl_h2d_fresh)
    hipStream_t stream_d2h = nullptr;  // pcl_batch_fetch_async: results travel to the host beside the next step's kernels
    bool dp_async = true;              // env PCL_DP_STREAM=0: everything on one stream
    int score_wgs_per_cu = 0;    // pcl_score_occupancy: 2 = the matrix-pipe scoring kernel leaves a third of each CU to kernels beside it
    int score_variant = 0;       // 7 = two-piece f16 split on the matrix pipe (default), 3 = f32-input MFMA (strict f32), 1 = direct form on the VALU
    // conditioning of the centred expansion the MFMA kernels use: cond[j] = max_m log2e sum_d (mu - c_j)^2 / (2 var),
    // the magnitude of the terms that cancel in it.  States above cond_max are scored / accumulated by the
    // direct-form VALU kernels instead (f32 error of the expansion ~ 5e-7 * cond nats).
    float *d_cond = nullptr;
    std::vector<float> cond;
    float cond_max = 96.f;
    // Split states.  The limit is a property of single MIXTURES (one tight mixture far from the state's centre), and after an M-step most states
    // have a few of them (tools/cond_probe.py: 7 % of the mixtures, every state).  A mixture with cond_m > cond_max is therefore taken OUT of
    // the matrix-pipe layouts (written like a zero-weight mixture, left out of the state's feature scales, K0 and cond) and evaluated by the
    // direct-form kernels over the state's compacted list of such mixtures, merged by a log-add (scoring) / added to the statistics
    // (accumulate).  A state goes to the direct-form kernels as a whole only when more than split_max of its mixtures are out.
    unsigned char *d_bad = nullptr;    // [J][Mpad] 1 = the mixture is off the matrix pipe
    int *d_bad_idx = nullptr;          // [J][Mpad] the state's off-pipe mixtures in ascending order (first nbad[j] entries)
    int *d_nbad = nullptr;             // [J]
    int *d_non = nullptr;              // [J] mixtures on the pipe (M - nbad)
    int *d_good_idx = nullptr;         // [J][Mpad] the state's ON-pipe mixtures in ascending order (first M - nbad[j] entries): row r of a split state's matrix-pipe layout is mixture good_idx[r]
    bool compact_main = true;          // env PCL_COMPACT_MAIN=0 (read when the context is made): every state keeps derive_kernel's tiles in mixture order, all of them walked (A/B)
    int *d_npt = nullptr;              // [J] 32-mixture tiles of the matrix-pipe layout in use: ceil((M - nbad) / 32)
    std::vector<int> nbad;
    float split_frac = 0.5f;           // env PCL_SPLIT_MAX: the share of a state's mixtures that may be off the pipe (0 = no splitting)
    bool split_frac_set = false;       // PCL_SPLIT_MAX was given (otherwise: 1.0 for scoring when the coarse pass is available, 0.5 for the accumulate pass)
    int split_max = 0;                 // = split_frac * M: mixtures per state that may be off the pipe (0: no splitting -- whole states, as before round 4)
    int model_gen = 0;           // bumped whenever the layouts (and cond) are re-derived
    // Coarse layout of the off-pipe mixtures (gmm_score_coarse.hip): their bound v_up on the matrix pipe, exact evaluation of what it
    // cannot rule out.  Derived on first use after the model changed (coarse_gen != model_gen).
    unsigned short *pmc = nullptr;     // [J][Mpad32/32][2][KS8f][64 lanes][8]: the state's bad_idx list, 32 per tile
    float *fscale_c = nullptr;         // [J][2][KS8f*8]
    double *kzero_c = nullptr;         // [J]
    float *rows_c = nullptr;           // [J][Mpad][D][2]: the tight mixtures' direct-form rows (s_d, c_d) in bad_idx order, for the pairs the coarse pass cannot rule out
    float *kgap_c = nullptr;           // [J]: (the state's largest tight k2) - K0, rounded up (the one-product pass: see EPS1 in gmm_score_coarse.hip)
    int coarse_np = 1;                 // env PCL_COARSE_PASSES (read when the context is made): 1 = one f16 product per term, 3 = the two-piece operands' three
    double *k2c = nullptr;             // [J][Mpad] exact log2-domain constant of the idx-th off-pipe mixture
    int *d_nct = nullptr;              // [J] coarse tiles in use
    unsigned long long *d_coarse_counter = nullptr;   // PCL_COARSE_STATS=1: pairs evaluated exactly
    int coarse_gen = -1;
    float coarse_split_frac = 0.99f;   // env PCL_COARSE_SPLIT_MAX: the share of a state's mixtures that may be off the pipe with the coarse pass (see pcl_model_upload)
    bool coarse_stats = false;         // env PCL_COARSE_STATS=1 (read when the context is made): count the pairs evaluated exactly (pcl_coarse_counter)
    bool coarse_on = true;             // env PCL_COARSE=0 (read when the context is made): the direct-form subset launch of rounds 4-5 instead (A/B)
    // the accumulate pass keeps the round 4-5 rule (whole states in direct form above acc_split_max off-pipe mixtures): its subset launch has
    // no coarse pass, and at a high share of off-pipe mixtures the whole-state kernel is the cheaper of its two routes
    int acc_split_max = 0;
    float *mean32 = nullptr;     // J * Mpad * D raw means (accumulate kernel)
    double *mean64 = nullptr;    // float64 master copy of the model: mean, var (J*Mpad*D), weight (J*Mpad)
    double *var64 = nullptr, *w64 = nullptr;
    // frames (device)
    int64_t F = 0;
    int FD = 0, FDhost = 0;
    float *frames32 = nullptr;
    double *frames64 = nullptr;
    // streaming: two frame slots; frames32 points into slot frames_front (or is a plain upload when frames_front < 0);
    // pcl_frames_stage copies the next chunk into the other slot on stream_aux, pcl_frames_swap makes it current
    float *frames_slot[2] = {nullptr, nullptr};
    size_t frames_slot_cap[2] = {0, 0};
    int frames_front = -1, staged_slot = -1, staged_D = 0;
    int64_t staged_F = 0;
    hipEvent_t ev_stage = nullptr, ev_slot_free = nullptr;
    bool have_slot_free = false;
    // E-step statistics (device, float64, linear domain)
    double *stats = nullptr;  // one allocation: [acc J*Mpad | alpha J | mean J*Mpad*D | cov J*Mpad*D]
    size_t stats_len = 0;
    double *st_acc = nullptr, *st_alpha = nullptr, *st_mean = nullptr, *st_cov = nullptr;
    double *d_softplus = nullptr;   // table of log1p(exp(-d)) for the forward-backward recursion (hmm_dp.hip)
    bool stats_fresh = false;
    // pcl_stats_zero clears the block on the auxiliary stream (3.9 GB: 0.58 ms on the main stream in front of every E-step's scoring,
    // which does not touch it); whoever touches the statistics next on the main stream waits for it first (pcl_stats_join)
    hipEvent_t ev_zero = nullptr, ev_zero_src = nullptr;
    bool zero_pending = false;
    double acc_prune_log2 = -1e300;   // pcl_accumulate_prune: frames with gamma_t(j) below 2^this are left out (default: only exact zeros)    // all zero since pcl_stats_zero: the first accumulate pass may store instead of read-modify-write
    // unit inventory (hmm_units.hip): n_units HMMs of S states, unit i owns GMM states i*(S-2) .. i*(S-2)+S-3
    int n_units = 0, S = 0;
    std::vector<double> unit_trans, unit_logtrans;   // host copies [n_units][S][S]: transmat and np.log(transmat)
    double *d_unit_trans = nullptr;                  // device copy of unit_trans (the transition M-step writes it)
    double *hmm_ksai = nullptr, *hmm_gamma = nullptr;   // per-unit accumulators, LOG domain: [n_units][S-2][S], [n_units][S-2]
    // pronunciation tree for the decoder (hmm_decode.hip)
    int *lex_units = nullptr, *lex_nunits = nullptr, *lex_child_ptr = nullptr, *lex_child_idx = nullptr, *lex_word = nullptr, *lex_roots = nullptr;
    int4 *lex_info = nullptr;                        // per node (first child, children, words end here, unit pair)
    double *d_unit_logtrans = nullptr;
    int lex_nodes = 0, lex_nroots = 0;
    // multi-GPU (pcl_comm.hip): RCCL communicator, or the host-callback rehearsal transport
    void *comm = nullptr;
    int rank = 0, nranks = 1;
    int transport = 0;                               // 0 none, 1 RCCL, 2 host callback (several ranks on ONE device)
    pcl_allgather_fn host_allgather = nullptr;
    void *host_user = nullptr;
    float *payload32 = nullptr;                      // f32 staging of the statistics / parameters (payload = PCL_F32)
    size_t payload32_len = 0;
    // pipelined exchange (pcl_comm.hip): state chunks go through reduce-scatter -> M-step -> all-gather -> derive on stream_comm
    // as soon as the accumulate pass has queued the last kernel that touches them
    hipStream_t stream_comm = nullptr;
    std::vector<hipEvent_t> pipe_ev;
    hipEvent_t pipe_done = nullptr;
    bool pipe_active = false;
    int pipe_early = 0;                                            // chunks of the last pipelined call that left while the pass was still running
    int pipe_K = 0, pipe_next = 0, pipe_payload = 0, pipe_mode = 1;   // mode 1: only the reduce-scatter leaves early; 0: the whole chain
    double pipe_c_cov = 0.0;
    std::map<std::string, KernelTimer> timers;
    bool timing = false;         // pcl_timing_enable / env PCL_TIMERS: record HIP events around every launch
    // Batches the caller has destroyed while the GPU was still working on them (a corpus sweep drops the batch of step k - 2 while
    // step k runs): pcl_batch_destroy returns at once; the memory goes back to the pool when the batch's OWN last work on every
    // stream has completed (its events: ev_mark / ev_dp / ev_fetch; pcl_batch_reap on the next create / destroy, pcl_sync, pcl_destroy).
    std::vector<struct pcl_batch *> graves;
    // Descriptor uploads of a batch under construction (pcl_desc_group): the arrays are packed into ONE page-locked staging buffer and
    // queued as asynchronous copies on stream_desc, with one wait at the end of the group -- pcl_batch_create_labels made 16 separate
    // synchronous copies from pageable memory (5.3 ms of host time per 1024-utterance batch beside a busy GPU, tools/fresh_batch_probe.py).
    // The copy itself is a KERNEL on stream_desc that reads the staging buffer over PCIe, not hipMemcpyAsync: the copy engines serve
    // their queue in order, and in a pipeline that also moves frames up and results down the descriptors sat behind the NEXT step's
    // frame upload, which waits for its slot's last reader -- 15 ms per pcl_batch_create_labels (bench.py pcie_inclusive_loop).
    char *desc_pin = nullptr;
    size_t desc_pin_cap = 0, desc_pin_used = 0;
    int desc_group = 0;          // > 0: inside a group, pcl_h2d_fresh stages and does not wait
    void *desc_dst[24];          // the staged entries of the open group (PCL_DESC_MAX)
    unsigned long long desc_off[24], desc_bytes[24];
    int desc_n = 0;
};

struct pcl_batch {
    pcl_ctx *ctx = nullptr;
    int U = 0, Nmax = 0, Tmax = 0;
    bool has_one_frame = false;   // an utterance of ONE frame: the reference's Baum-Welch raises on it (golden G15) and it adds nothing to any accumulator
    long long sumNT = 0, sumN = 0, sumT = 0, sumNN = 0;
    long long max_frame_end = 0;   // max over utterances of frame_begin + T: re-validated against the CURRENT frame matrix
    int model_J = 0;               // J of the model the state lists were built for
    int max_state = -1;            // largest GMM state id any row refers to: re-validated against the CURRENT model
    std::vector<UttDesc> utt;  // host copy
    std::vector<int32_t> row_state;
    bool have_trans = false, have_states = false, have_B = false, have_fb = false, have_vit = false, have_post = false;
    bool launched = false;           // a kernel that reads this batch's descriptors may be in flight: their re-upload goes through the main stream
    bool virt_rows_filled = false;   // the constant entry / exit rows of Bt are in place for the current row map
    int max_outdeg = 0, max_indeg = 0;
    long long nnz = 0;
    // device
    UttDesc *d_utt = nullptr;
    double *Bt = nullptr, *alpha = nullptr, *beta = nullptr, *lgam = nullptr;
    double *logpi = nullptr, *pi_out = nullptr, *gamma_out = nullptr, *ksai = nullptr;
    double *logp = nullptr, *qtrace = nullptr, *point = nullptr;
    int32_t *npass = nullptr, *path = nullptr;
    int *row_ptr = nullptr, *col_idx = nullptr;   // CSR (successors)
    double *csr_val = nullptr;
    int *col_ptr = nullptr, *row_idx = nullptr;   // CSC (predecessors, ascending source index)
    double *csc_val = nullptr;
    double *xi_m = nullptr, *xi_s = nullptr;      // per CSR entry online-LSE state
    // scaled linear-domain forward-backward (hmm_fb_linear.hip): packed exp(B), the int32 exponents of alpha / beta (their
    // mantissas live in `alpha` / `beta`), per utterance the exponent maxima of the range test
    unsigned long long *Bp = nullptr;
    int *alpha_e = nullptr, *beta_e = nullptr, *fb_kmax = nullptr;
    double *fb_dump = nullptr;
    double *fb_part_m = nullptr;                  // per utterance, sum and wave: the posterior kernel's partial sums (mantissa, exponent)
    int *fb_part_e = nullptr;
    bool left_right = false;                      // every state is reached from itself / the state before it only (AcousticModel.embedded)
    bool fb_linear = false;                       // the last forward-backward left (mantissa, exponent) pairs in alpha / beta
    unsigned short *bp = nullptr;                 // Viterbi back-pointers, time-major (t, n)
    int32_t *d_row_state = nullptr;
    // scoring work lists
    ScoreSeg *d_segs = nullptr;
    std::vector<ScoreSeg> segs;              // host copy, sorted by state
    std::vector<int> state_seg_lo, state_seg_hi;  // per state with work: segment range
    std::vector<int> state_seg_hip;               // ... of which [lo, hip) are scored and [hip, hi) are copies of a scored row
    std::vector<DupRow> dups;
    DupRow *d_dups = nullptr;
    std::vector<int> work_states;
    ScoreTile *d_tiles = nullptr;            // tiles for the precision last scored with (MFMA kernel in MFMA mode)
    std::vector<int> acc_ws, acc_lo, acc_hi, acc_split; // accumulate's state order (well-conditioned first)
    hipEvent_t ev_main = nullptr, ev_dp = nullptr;   // main stream -> stream_dp hand-over, and back
    bool mark_is_score = false;              // ev_mark sits right behind this batch's last scoring: the second stream may wait for IT instead of a new record on the main stream
    hipEvent_t ev_mark = nullptr;            // the main stream behind the last work this batch queued there (pcl_batch_mark): what pcl_batch_destroy waits for -- not the work later batches queued behind it
    bool dp_pending = false;                 // forward-backward queued on stream_dp and not yet joined
    hipEvent_t ev_fetch = nullptr, ev_fetch_src = nullptr;   // pcl_batch_fetch_async: copies done / the main stream at the time of the call
    bool fetch_pending = false;              // result copies queued on stream_d2h: the next compute call on this batch waits for them
    int *d_tile_flags = nullptr;             // split-f16 scoring: per tile, 1 = a scaled feature left the f16 range (rescored)
    ScoreTile *d_tiles_v = nullptr;          // MFMA mode only: tiles of ill-conditioned states for the VALU kernel
    ScoreTile *d_tiles_s = nullptr;          // MFMA mode only: tiles of the split states for the subset launch (their off-pipe mixtures)
    ScoreTile *d_tiles_c = nullptr;          // ... and for the coarse pass (gmm_score_coarse.hip): the same states at the matrix-pipe tile size
    int *d_tile_flags_c = nullptr;           // coarse pass: per tile, 1 = a scaled feature left the f16 range (the direct-form subset kernel rescored it)
    int n_tiles_c = 0;
    int n_segs = 0, n_tiles = 0, n_tiles_v = 0, n_tiles_s = 0, tile_frames = 0, tile_gen = -1;
    double *tmp = nullptr;                   // sumNT staging buffer for layout conversion
    double *nz_tmp = nullptr;                // nnz staging buffer for the sparse xi download
    std::vector<int> seg_of_row;              // (utterance, row) -> its segment (-1: not a GMM row); d_seg_of_row: the device copy (sumN ints)
    int *d_seg_of_row = nullptr;
    int max_N = 0;                            // rows of the largest sentence HMM of the batch
    // (the accumulate pass's device work lists and tile images are the CONTEXT's scratch since round 5: pcl_ctx::acc)
    // decoder state (hmm_decode.hip): token buffers, node -> token map, scratch, results
    double *dec_f64 = nullptr;
    int *dec_slot = nullptr, *dec_work = nullptr, *dec_int = nullptr;
    double *dec_score = nullptr;
    int dec_cap = 0, dec_cand = 0, dec_nodes = 0;
    bool have_dec = false;
    // label-built batches (pcl_batch_create_labels): the labels, and per unit the list of its occurrences
    bool from_labels = false;
    std::vector<int32_t> label_len, labels;
    std::vector<double> logpi_u;             // ln pi of every state of utterance u (uniform 1/N, AcousticModel.py:1003-1006)
    int n_occ = 0;
    int *occ_ptr = nullptr, *occ_utt = nullptr, *occ_row0 = nullptr;   // unit -> [occ_ptr[u], occ_ptr[u+1]) -> (utterance, first emitting row)
};

// ---------------------------------------------------------------- scaled linear-domain forward-backward (hmm_fb_linear.hip)
// value = m 2^e, e an int32 per lane.  Zero lives in the exponent: below LE_PZ a value IS zero; a zero factor (ln 0) adds
// LE_ZADD; sums are floored at LE_FLOOR so that nothing wraps; real exponents stay inside (-LE_LIMIT, LE_LIMIT).
constexpr int LE_ZADD = -(1 << 29);
constexpr int LE_FLOOR = -(1 << 30);
constexpr int LE_PZ = -(1 << 28);
constexpr int LE_LIMIT = 1 << 26;
// does utterance u (T frames) fit the int32 exponents and the packed emission word?  kmax = per utterance a record of
// PCL_FB_KREC ints written by hmm_emis_pack_kernel: max |k| (k = power of two of exp(.)) of the emissions, one per packing
// workgroup [0..7], of ln A [8], of the caller's ln pi [9]; nullptr = the scaled kernels are not in use
constexpr int PCL_FB_KREC = 10;
__device__ __forceinline__ bool pcl_fb_linear_ok(const int *kmax, int u, int T) {
    if (!kmax) return false;
    const int *r = kmax + (size_t)PCL_FB_KREC * u;
    int kb = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) kb = max(kb, r[k]);
    const long long ka = r[8], kp = r[9];
    return kb < 32760 && ((long long)kb + ka + 4) * (long long)(T + 2) + kp < (long long)LE_LIMIT;
}

// ---------------------------------------------------------------- error helpers
#define PCL_FAIL(ctx, code, ...)                         \
    do {                                                 \
        char _b[512];                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);           \
        pcl_set_error((ctx), _b);                        \
        return (code);                                   \
    } while (0)
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) PCL_FAIL(ctx, PCL_ERR_HIP, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

void pcl_set_error(pcl_ctx *ctx, const char *msg);

// Device memory comes from a process-wide caching pool (pcl_api.hip): hipMalloc / hipFree cost 0.1-1 ms each and hipFree
// waits for the whole device, which a library that creates and drops a batch per utterance (the drop-in classes) or per
// chunk (streaming) cannot afford.  A freed block goes back to the pool; dev_free first waits for the device, as hipFree
// did, unless the caller has already made sure the GPU is done with the block (pcl_free_synced_scope).
void *pcl_pool_alloc(int device, size_t bytes);           // nullptr: out of memory even after the cache was released
void pcl_pool_free(void *p);
extern thread_local int pcl_tls_free_synced;              // > 0: dev_free skips its device-wide wait
struct pcl_free_synced_scope {
    pcl_free_synced_scope() { ++pcl_tls_free_synced; }
    ~pcl_free_synced_scope() { --pcl_tls_free_synced; }
};
template <typename T>
static inline int dev_alloc(pcl_ctx *ctx, T **p, size_t n) {
    if (n == 0) n = 1;
    *p = static_cast<T *>(pcl_pool_alloc(ctx->device, n * sizeof(T)));
    if (!*p) PCL_FAIL(ctx, PCL_ERR_NOMEM, "device memory: %zu bytes", n * sizeof(T));
    return PCL_OK;
}
template <typename T>
static inline void dev_free(T *&p) {
    if (p) pcl_pool_free((void *)p);
    p = nullptr;
}
// Host -> device copy on the context's main stream, complete on return: unlike hipMemcpy (legacy null stream) it does not
// wait for the work of the other streams (a decoder running on the second stream while the next chunk's batch is built).
static inline hipError_t pcl_h2d(pcl_ctx *ctx, void *dst, const void *src, size_t bytes);
#define TRY(x)                    \
    do {                          \
        int _r = (x);             \
        if (_r != PCL_OK) return _r; \
    } while (0)

static inline hipError_t pcl_h2d(pcl_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}
// Descriptors into a buffer NO queued kernel can be reading (freshly allocated, or any buffer of a batch that has not launched
// anything yet): a stream of their own, so that creating a batch in the middle of a stream of chunks does not wait for the
// scoring kernel that happens to run on the main stream (28 ms per new batch in the C5 pipeline).  Complete at return.
// One launch copies every staged array of a descriptor group from the page-locked staging buffer (the kernel reads host memory
// over PCIe) to its device array: pcl_api.hip.  entries: (dst, offset into the staging buffer, bytes).
constexpr int PCL_DESC_MAX = 24;
struct DescCopyArgs {
    void *dst[PCL_DESC_MAX];
    unsigned long long off[PCL_DESC_MAX], bytes[PCL_DESC_MAX];
    int n;
};
hipError_t pcl_desc_flush(pcl_ctx *ctx);      // launch the pending entries on stream_desc and wait for them

static inline hipError_t pcl_h2d_fresh(pcl_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (!ctx->stream_desc) return pcl_h2d(ctx, dst, src, bytes);
    if (ctx->desc_group > 0) {                                       // inside a pcl_desc_group: staged, copied by ONE kernel at its end
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (ctx->desc_pin_used + need > ctx->desc_pin_cap || ctx->desc_n == PCL_DESC_MAX) {
            hipError_t e = pcl_desc_flush(ctx);                      // what is staged so far has to land before the buffer is reused
            if (e != hipSuccess) return e;
            if (need > ctx->desc_pin_cap) {
                if (ctx->desc_pin) (void)hipHostFree(ctx->desc_pin);
                ctx->desc_pin = nullptr;
                ctx->desc_pin_cap = 0;
                const size_t cap = std::max<size_t>((size_t)16 << 20, need * 2);
                e = hipHostMalloc((void **)&ctx->desc_pin, cap, hipHostMallocDefault);
                if (e != hipSuccess) return e;
                ctx->desc_pin_cap = cap;
            }
        }
        memcpy(ctx->desc_pin + ctx->desc_pin_used, src, bytes);
        // ONE kernel copies all pending entries side by side: two entries with the same destination would race.  A later upload of an array
        // replaces the earlier one (pcl_batch_create_labels uploads the utterance descriptors twice: before and after the transition
        // offsets are known -- when the first version won, every utterance read utterance 0's transitions: wrong results as soon as the
        // unit matrices differ, tests/test_gpu_sweep.py::test_the_queueing_knobs_do_not_move_a_bit)
#ifndef PCL_DESC_RACE_REPRO                                            // (a build WITH this macro restores the bug: what the sweep tests must catch)
        for (int k = 0; k < ctx->desc_n; ++k)
            if (ctx->desc_dst[k] == dst) ctx->desc_bytes[k] = 0;
#endif
        ctx->desc_dst[ctx->desc_n] = dst;
        ctx->desc_off[ctx->desc_n] = ctx->desc_pin_used;
        ctx->desc_bytes[ctx->desc_n] = bytes;
        ++ctx->desc_n;
        ctx->desc_pin_used += need;
        return hipSuccess;
    }
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream_desc);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream_desc);
}
// Scope of a batch's descriptor uploads: every pcl_h2d_fresh inside is staged; `finish` copies and waits once for all of them
// (the destructor does too, for the error paths).  Groups nest: the outermost one copies.
struct pcl_desc_group {
    pcl_ctx *ctx;
    bool open;
    explicit pcl_desc_group(pcl_ctx *c) : ctx(c), open(true) { ++ctx->desc_group; }
    hipError_t finish() {
        if (!open) return hipSuccess;
        open = false;
        if (--ctx->desc_group > 0) return hipSuccess;
        return pcl_desc_flush(ctx);
    }
    ~pcl_desc_group() { (void)finish(); }
};

// the main stream behind an asynchronous pcl_stats_zero: called by every entry point that reads or writes the statistics block
static inline hipError_t pcl_stats_join(pcl_ctx *ctx) {
    if (!ctx->zero_pending) return hipSuccess;
    ctx->zero_pending = false;
    return hipStreamWaitEvent(ctx->stream, ctx->ev_zero, 0);
}

// Called at the end of every entry point that queues main-stream work reading or writing the batch's buffers (the auxiliary
// stream's producers are always joined by a main-stream consumer queued behind them).
static inline hipError_t pcl_batch_mark(pcl_batch *b) {
    if (!b->ev_mark) {
        hipError_t e = hipEventCreateWithFlags(&b->ev_mark, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    return hipEventRecord(b->ev_mark, b->ctx->stream);
}

// Fewer packets between two scoring launches on the main queue (tools/trace_gaps.py: every event record is a marker the command
// processor handles between the kernels): the recursion on the second stream waits for the event pcl_batch_score left (ev_mark), and
// pcl_batch_fetch_async for the recursion's event, instead of each recording one more on the main stream.  PCL_FEWER_MARKERS=0: as rounds 1-4.
static inline bool pcl_fewer_markers() {
    static const bool on = !(getenv("PCL_FEWER_MARKERS") && atoi(getenv("PCL_FEWER_MARKERS")) == 0);
    return on;
}
// stream_dp behind this batch's main-stream work.  `after_fetch`: result copies of an earlier recursion (pcl_batch_fetch_async) were
// still pending when the caller was entered -- the caller put the MAIN stream behind them, but under the shortcut below stream_dp does
// not follow the main stream's head, so it waits for the copies itself (score -> FB -> fetch_async -> FB again would
// otherwise overwrite lgam / ksai / logp / path under the D2H copies).  The shortcut is good for ONE recursion per scoring: whatever
// the main stream queues on this batch afterwards (the accumulate pass reads lgam) has to be ahead of the next recursion, so the flag
// is consumed here and the next caller records a fresh event on the main stream.
static inline hipError_t pcl_dp_follows_main(pcl_batch *b, bool after_fetch = false) {
    pcl_ctx *ctx = b->ctx;
    hipError_t e;
    if (after_fetch && b->ev_fetch && (e = hipStreamWaitEvent(ctx->stream_dp, b->ev_fetch, 0)) != hipSuccess) return e;
    const bool shortcut = pcl_fewer_markers() && b->mark_is_score && b->ev_mark;
    b->mark_is_score = false;
    if (shortcut) return hipStreamWaitEvent(ctx->stream_dp, b->ev_mark, 0);
    if (!b->ev_main && (e = hipEventCreateWithFlags(&b->ev_main, hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(b->ev_main, ctx->stream)) != hipSuccess) return e;
    return hipStreamWaitEvent(ctx->stream_dp, b->ev_main, 0);
}

// shared by pcl_api.hip and hmm_units.hip (C linkage, internal)
extern "C" {
int pcl_batch_upload_sparse(pcl_batch *b, const std::vector<int> &row_ptr, const std::vector<int> &col_idx,
                            const std::vector<double> &csr_val, const std::vector<int> &col_ptr,
                            const std::vector<int> &row_idx, const std::vector<double> &csc_val, const double *logpi);
int pcl_batch_set_states_impl(pcl_batch *b, const int32_t *row_state);
}
void pcl_timer_begin(pcl_ctx *ctx, const char *which);
void pcl_timer_end(pcl_ctx *ctx, const char *which);

// ---------------------------------------------------------------- kernel launchers (one per .hip file)
int pcl_launch_score(pcl_ctx *ctx, pcl_batch *b, int precision, const ScoreTile *tiles, int n_tiles);
int pcl_launch_fill_virtual_rows(pcl_ctx *ctx, pcl_batch *b);
int pcl_launch_forward_backward(pcl_ctx *ctx, pcl_batch *b, int fix_pi, double threshold);
int pcl_launch_viterbi(pcl_ctx *ctx, pcl_batch *b, int end_state_back);
bool pcl_fb_linear_enabled();                                              // env PCL_FB_LINEAR=0: the log-domain kernels only
int pcl_launch_fb_linear(pcl_ctx *ctx, pcl_batch *b, int fix_pi, double threshold);
int pcl_launch_fb_linear_post(pcl_ctx *ctx, pcl_batch *b);
int pcl_launch_fb_to_log(pcl_ctx *ctx, pcl_batch *b, const double *m, const int *e, double *out);
int pcl_launch_regroup(pcl_ctx *ctx, pcl_batch *b, const int32_t *d_row_unit, int gmm_num, int32_t *d_frame_unit, int32_t *d_frame_k);
int pcl_launch_ksai_gather(pcl_ctx *ctx, pcl_batch *b, double *dst);
int pcl_launch_clock_probe(pcl_ctx *ctx, int spin_us, unsigned long long *d_out);
int pcl_launch_accumulate(pcl_ctx *ctx, pcl_batch *b, int precision);
void pcl_accumulate_release(pcl_ctx *ctx);           // the context's accumulate scratch (pcl_destroy, pcl_model_upload)
int pcl_launch_transpose(pcl_ctx *ctx, pcl_batch *b, const double *src, double *dst, int to_time_major);
int pcl_score_tile_frames(int D, int precision);
int pcl_launch_score_mfma(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles);
int pcl_launch_acc16_produce(pcl_ctx *ctx, pcl_batch *b, int first, int ns, int max_tiles, int buf, hipStream_t stream);
int pcl_launch_acc16_consume(pcl_ctx *ctx, pcl_batch *b, int first, int ns, int buf, bool fresh, hipStream_t stream);
size_t pcl_acc16_image_bytes(int D);
int pcl_launch_score_split16(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles);
int pcl_score_split16_tile_frames();
int pcl_launch_score_fixup(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles, const int *flags);
int pcl_launch_score_subset(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles);
int pcl_score_subset_tile_frames(int D);
int pcl_score_mfma_tile_frames();
bool pcl_score_mfma_supported(int D);
int pcl_launch_dup_rows(pcl_ctx *ctx, pcl_batch *b);
int pcl_launch_derive(pcl_ctx *ctx);
int pcl_launch_derive_range(pcl_ctx *ctx, int j_lo, int j_hi);   // no wait, no generation bump: pcl_derive_finish closes
int pcl_derive_finish(pcl_ctx *ctx);
int pcl_pipe_begin(pcl_ctx *ctx, double c_covariance, int payload, int n_chunks);
int pcl_pipe_progress(pcl_ctx *ctx, int final_below);
int pcl_pipe_finish(pcl_ctx *ctx, int update_transitions);
void pcl_pipe_release(pcl_ctx *ctx);
int pcl_ensure_layouts(pcl_ctx *ctx, int need);
enum { PCL_LAYOUT_P32 = 1, PCL_LAYOUT_P64 = 2, PCL_LAYOUT_PM32 = 4, PCL_LAYOUT_COND = 32, PCL_LAYOUT_PM16F = 128 };
inline bool pcl_state_uses_valu(const pcl_ctx *ctx, int j) {
    if (ctx->cond.empty()) return false;
    if (ctx->split_max > 0 && !ctx->nbad.empty())              // split states: only too many off-pipe mixtures (or constants out of f16 range)
        return ctx->nbad[j] > ctx->split_max || ctx->cond[j] >= 1.0e30f;
    return ctx->cond[j] > ctx->cond_max;
}
inline bool pcl_state_is_split(const pcl_ctx *ctx, int j) { return !ctx->nbad.empty() && ctx->nbad[j] > 0 && !pcl_state_uses_valu(ctx, j); }
// ... and for the accumulate pass (its own limit: see acc_split_max)
inline bool pcl_state_acc_uses_valu(const pcl_ctx *ctx, int j) {
    if (ctx->cond.empty()) return false;
    if (ctx->split_max > 0 && !ctx->nbad.empty()) return ctx->nbad[j] > ctx->acc_split_max || ctx->cond[j] >= 1.0e30f;
    return ctx->cond[j] > ctx->cond_max;
}
inline bool pcl_state_acc_is_split(const pcl_ctx *ctx, int j) { return !ctx->nbad.empty() && ctx->nbad[j] > 0 && !pcl_state_acc_uses_valu(ctx, j); }
bool pcl_coarse_enabled(const pcl_ctx *ctx);
bool pcl_coarse_enabled_for(const pcl_ctx *ctx, int D);   // ... for a model of (padded) dimension D about to be uploaded
int pcl_coarse_tile_frames();
void pcl_coarse_release(pcl_ctx *ctx);
int pcl_ensure_coarse(pcl_ctx *ctx);
int pcl_launch_score_coarse(pcl_ctx *ctx, pcl_batch *b);
int pcl_launch_compact_main(pcl_ctx *ctx, int j_lo, int j_hi);
int pcl_launch_score_subset_flagged(pcl_ctx *ctx, pcl_batch *b, const ScoreTile *tiles, int n_tiles, const int *flags);
inline float pcl_split_threshold(const pcl_ctx *ctx) { return ctx->split_max > 0 ? ctx->cond_max : 3.0e38f; }   // cond_m above this: off the pipe
int pcl_launch_cast(pcl_ctx *ctx, const double *src64, float *f32, double *dst64, size_t n);
int pcl_launch_mstep(pcl_ctx *ctx, double floor_var);
int pcl_launch_mstep_range(pcl_ctx *ctx, double floor_var, int j_lo, int j_hi);
int pcl_launch_hmm_acc_merge_prepare(pcl_ctx *ctx, double *top);          // top[i] = acc[i] (copy for the max all-reduce)
int pcl_launch_hmm_acc_merge_scale(pcl_ctx *ctx, const double *top);      // acc[i] = exp(acc[i] - top[i]) (0 where top = -inf)
int pcl_launch_hmm_acc_merge_finish(pcl_ctx *ctx, const double *top);     // acc[i] = top[i] + ln(acc[i])
int pcl_launch_trans_mstep(pcl_ctx *ctx);
void pcl_units_release(pcl_ctx *ctx);
void pcl_batch_units_release(pcl_batch *b);
void pcl_comm_release(pcl_ctx *ctx);
void pcl_lexicon_release(pcl_ctx *ctx);
void pcl_batch_decode_release(pcl_batch *b);
int pcl_launch_pack(pcl_ctx *ctx, const double *src, int inner, double *dst);

/*
 * poccala_hip.h  --  C-ABI of libpoccala_hip.so, the MI355X (gfx950) engine for the
 * GMM-HMM hot path of Byshx/Poccala.
 *
 * The reference has no FFI layer: its boundary is the Python class surface
 * (StatisticalModel/LHMM.py, StatisticalModel/Clustering.py, AcousticModel/AcousticModel.py).
 * Every entry point below names the reference function(s) it replaces.  The Python
 * drop-in classes in poccala_amd/ call these through ctypes and nothing else.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Return 0 on success, a negative
 *     pcl_status on failure; pcl_last_error() gives the message.  Nothing throws
 *     across the ABI.
 *   - The caller owns every host buffer (C-contiguous NumPy arrays).  The library
 *     owns all device memory (inside pcl_ctx / pcl_batch).
 *   - One pcl_ctx = one GPU.  Calls on a ctx are serialised by the caller (one process or
 *     thread per GPU).  Internally a ctx owns two HIP streams: pcl_batch_forward_backward and
 *     pcl_batch_viterbi run on the second one, ordered after everything queued before them, so that they
 *     overlap the scoring of ANOTHER batch queued after them; any later call on the same
 *     batch, pcl_sync and every download wait for it (env PCL_DP_STREAM=0: one stream).  Uploads, downloads (pcl_batch_get,
 *     pcl_*_download) and pcl_stats_allreduce are synchronous at return.  The compute
 *     calls -- pcl_batch_score / _forward_backward / _viterbi / _accumulate,
 *     pcl_stats_zero, pcl_mstep -- are ASYNCHRONOUS: they enqueue kernels in call order
 *     and return; pcl_sync() or any download completes them.
 *   - Host-side matrices use the REFERENCE layout: (N,T) row-major emission /
 *     alpha / beta matrices, float64, log domain, -inf for impossible.
 *   - log A and log pi are passed ALREADY LOGGED by the caller (np.log), because
 *     bit-exact Viterbi is defined on those values (LHMM.py:571,577).
 */
#ifndef POCCALA_HIP_H
#define POCCALA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is what libpoccala_hip.so exports, nothing else
 * (tests/test_cabi_loads.py holds exported pcl_* == declared). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct pcl_ctx pcl_ctx;
typedef struct pcl_batch pcl_batch;

typedef enum {
    PCL_OK = 0,
    PCL_ERR_INVALID = -1,  /* bad argument / shape (reference: DataDimensionError, AssertionError) */
    PCL_ERR_HIP = -2,      /* HIP runtime error */
    PCL_ERR_STATE = -3,    /* call order (e.g. score before model upload) */
    PCL_ERR_NOMEM = -4,
    PCL_ERR_COMM = -5      /* RCCL error */
} pcl_status;

/* arithmetic of the scoring / accumulate kernels */
#define PCL_F32 0 /* f32 Gaussian arithmetic, f64 dynamic programming (headline mode) */
#define PCL_F64 1 /* everything float64 (alignment-parity mode)            */

/* pcl_model_upload flags */
#define PCL_MODEL_Q1_SUMVAR 0 /* reference constant: -D/2 ln2pi - 1/2 sum(var)   (util.py:29, quirk Q1) */
#define PCL_MODEL_LOGDET 1    /* textbook constant:  -D/2 ln2pi - 1/2 sum(ln var) (opt-in, not parity)  */

/* row kinds in pcl_batch_set_states: >=0 is a GMM state id */
#define PCL_ROW_ENTRY (-1) /* VirtualState(1.): ln 1 = 0   (AcousticModel.py:218,1039-1043) */
#define PCL_ROW_EXIT (-2)  /* VirtualState(0.): ln 0 = -inf (AcousticModel.py:219)          */

/* pcl_batch_get selectors */
typedef enum {
    PCL_GET_B = 0,       /* emission matrices, ragged, (N_u,T_u) row-major f64  == LHMM.B_p / embedded() B   */
    PCL_GET_ALPHA = 1,   /* forward  matrices of the final pass, same layout    == LHMM.__result_f           */
    PCL_GET_BETA = 2,    /* backward matrices of the final pass                 == LHMM.__result_b           */
    PCL_GET_LGAMMA = 3,  /* ln gamma_t(i) = alpha+beta - LSE_i(alpha+beta), (N_u,T_u)  (LHMM.py:486-500); ln 0 for a one-frame utterance (the reference raises on it) */
    PCL_GET_KSAI = 4,    /* un-normalised ln xi, ragged dense (N_u,N_u) f64     == LHMM.__ksai (quirk Q5)    */
    PCL_GET_GAMMA = 5,   /* un-normalised ln gamma, ragged (N_u,) f64           == LHMM.__gamma              */
    PCL_GET_PI = 6,      /* pi after the final pass, ragged (N_u,) f64, LINEAR  == LHMM.pi                   */
    PCL_GET_LOGP = 7,    /* LSE_i alpha_{T-1}(i) of the final pass, (U,) f64    == __expectation, LHMM.py:412 */
    PCL_GET_NPASS = 8,   /* Baum-Welch passes run, (U,) int32 (quirk Q6)                                     */
    PCL_GET_QTRACE = 9,  /* Q after each pass, (U, PCL_MAX_PASS) f64, unused = NaN                            */
    PCL_GET_PATH = 10,   /* Viterbi state indices, ragged (T_u,) int32          == LHMM.viterbi mark_state   */
    PCL_GET_POINT = 11,  /* Viterbi score, (U,) f64                             == LHMM.viterbi point        */
    PCL_GET_KSAI_NZ = 12 /* ln xi of the stored transitions only (ln A > -inf), row-major order per utterance, f64;
                            a sentence HMM has ~2N of them instead of N*N                                     */
} pcl_get_what;
#define PCL_MAX_PASS 16

/* ---------------------------------------------------------------- context */
int pcl_init(int device, pcl_ctx **out);
/* pcl_destroy may be called with work in flight: it drains every stream of the context first (so asynchronous result copies have
 * landed when it returns), then releases the communicator, the batches already handed to pcl_batch_destroy, the model and the
 * streams.  Batches the caller never destroyed are NOT walked: their device blocks stay in the process-wide pool's books (a
 * leak, not a fault) -- destroy batches first.  Page-locked host memory from pcl_host_alloc belongs to the caller: free it
 * with pcl_host_free BEFORE pcl_destroy (pcl_host_free itself waits for every stream of the context, because hipHostFree does
 * not wait for copies still using the block).  tools/lifecycle_stress.py exercises all of this. */
int pcl_destroy(pcl_ctx *ctx);
const char *pcl_last_error(pcl_ctx *ctx); /* ctx may be NULL: error of a failed pcl_init */
int pcl_sync(pcl_ctx *ctx);
/* name (cap bytes), compute units, HBM bytes */
int pcl_device_info(pcl_ctx *ctx, char *name, int cap, int *cus, size_t *hbm_bytes);
/* GPU time of a kernel group since the last query, measured with HIP events recorded on the ctx
 * stream around every launch: which = "score" | "fb" | "viterbi" | "accumulate" | "allreduce".
 * Returns the summed milliseconds and the number of launches, then resets the group. */
int pcl_kernel_time(pcl_ctx *ctx, const char *which, float *total_ms, int *launches);
/* The events behind pcl_kernel_time are recorded only while timing is on (default off, or env PCL_TIMERS=1): a
 * training run that never queries them pays nothing and keeps nothing.  Turning it off drops pending events. */
int pcl_timing_enable(pcl_ctx *ctx, int on);
/* HIP devices visible to this process (0 without a GPU); does not create a context. */
int pcl_device_count(int *n);

/* ------------------------------------------------------------------ model
 * Replaces Clustering.GMM parameter state (T2: mean (M,D), covariance (M,D,D) of which only the
 * diagonal is used -- util.py:23 --, alpha (M,)) for J GMM states at once.  `var` is the DIAGONAL.
 * Derived device layouts (f32 and f64) are built here, in float64, once. */
int pcl_model_upload(pcl_ctx *ctx, int J, int M, int D, const double *mean /* J*M*D */,
                     const double *var /* J*M*D */, const double *weight /* J*M */, int flags);

/* ----------------------------------------------------------------- frames
 * The (F,D) MFCC matrix of the whole batch/corpus shard, rows = frames (LHMM.add_data, the `data`
 * argument of cal_observation_pro).  dtype: PCL_F32 or PCL_F64 host element type. */
int pcl_frames_upload(pcl_ctx *ctx, int64_t F, int D, const void *frames, int dtype);

/* Streaming form for a corpus that is fed chunk by chunk (BASELINE config 5; the reference reads one utterance at a time
 * from disk inside its worker, AcousticModel.py:723-768).  Two device slots: pcl_frames_stage queues the H2D copy of the
 * NEXT chunk (float32, row-major (F,D)) on the library's copy stream, into the slot that is not current, behind the
 * last kernels that read that slot, and returns at once; pcl_frames_swap waits for that copy (after it returns the
 * caller's buffer is free again) and makes the staged chunk the current frame matrix for batches created afterwards.
 * Batches of the previous chunk stay valid for everything that does not read frames (decode, forward-backward, Viterbi,
 * downloads); scoring / accumulating them again fails the row check unless the new chunk is at least as long.
 * pcl_host_alloc returns page-locked host memory (the copy is then truly asynchronous); any host pointer works. */
int pcl_frames_stage(pcl_ctx *ctx, int64_t F, int D, const float *frames);
int pcl_frames_swap(pcl_ctx *ctx);
int pcl_host_alloc(pcl_ctx *ctx, size_t bytes, void **out);
int pcl_host_free(pcl_ctx *ctx, void *ptr);

/* ------------------------------------------------------------------ batch
 * A batch = U sentence-level HMMs (AcousticModel.embedded, AcousticModel.py:957-1014), utterance u
 * having N[u] states and T[u] frames starting at row frame_begin[u] of the uploaded frame matrix
 * (frame_begin may be NULL when emissions are supplied with pcl_batch_set_emissions).  At most 65535 utterances, 2^31 - 1 rows
 * (sum N) and frames (sum T) per batch: PCL_ERR_INVALID beyond that -- a corpus goes through in several batches. */
int pcl_batch_create(pcl_ctx *ctx, int U, const int32_t *N, const int32_t *T, const int64_t *frame_begin,
                     pcl_batch **out);
int pcl_batch_destroy(pcl_batch *b);

/* ln A (ragged dense (N_u,N_u) row-major, -inf where A == 0) and ln pi (ragged (N_u,)).
 * == the transmat / pi arguments of LHMM.__init__ (LHMM.py:19) and LHMM.viterbi (LHMM.py:547). */
int pcl_batch_set_transitions(pcl_batch *b, const double *logA, const double *logpi);

/* Row -> GMM state map, ragged (N_u,): state id in [0,J), or PCL_ROW_ENTRY / PCL_ROW_EXIT.
 * == the profunction list of each unit HMM laid out by embedded() (AcousticModel.py:990-1001). */
int pcl_batch_set_states(pcl_batch *b, const int32_t *row_state);

/* Emissions given by the caller (ragged (N_u,T_u) row-major f64) instead of scored here
 * == LHMM(probmat=[B]) (LHMM.py:75) and the `prob` argument of LHMM.viterbi (LHMM.py:547). */
int pcl_batch_set_emissions(pcl_batch *b, const double *B);

/* Posteriors given by the caller (ragged (N_u,T_u) row-major f64, ln gamma_t(i)) instead of computed by
 * pcl_batch_forward_backward == the l_value argument of Clustering.GMM.update_acc (Clustering.py:653). */
int pcl_batch_set_posteriors(pcl_batch *b, const double *lgamma);

/* A1+A4+A5+A6: fill every row of every emission matrix:  ln b_j(o_t) = LSE_m[ln w_m + N(o_t; mu_m, var_m)]
 * == LHMM.cal_observation_pro (LHMM.py:163-187) -> Clustering.GMM.point (Clustering.py:740-767)
 *    -> util.gaussian_function (util.py:20-31), batched state-major over the whole batch. */
int pcl_batch_score(pcl_batch *b, int precision);

/* A8..A11 (+ the per-frame posteriors of A12): the Baum-Welch pass loop of LHMM.baulm_welch
 * (LHMM.py:526-544) for an LHMM built with probmat: forward (:335-351), backward (:353-366),
 * xi/gamma/pi (:394-471), Q (:412-422); passes repeat while Q - Q_prev > threshold (0.64, :539).
 * fix_pi = bit0 of the reference's fix_code (LHMM.py:140-145).  Results stay on the device;
 * read them with pcl_batch_get. */
int pcl_batch_forward_backward(pcl_batch *b, int fix_pi, double threshold);

/* A14: LHMM.viterbi (LHMM.py:546-609), end_state_back = 0|1 (quirk Q9). Bit-exact in f64. */
int pcl_batch_viterbi(pcl_batch *b, int end_state_back);

/* Next row f2: what follows forced alignment in training scheme 1, per frame, on the device.  row_unit: ragged (N_u,)
 * int32, the unit each HMM row belongs to (the `states` dictionary of AcousticModel.embedded, AcousticModel.py:968-976,
 * as integers).  Outputs, ragged (T_u,) int32: frame_unit[t] = unit of the Viterbi path at t (the name sequence of
 * AcousticModel.viterbi, :1016-1027) and frame_k[t] = which of the unit's gmm_num GMM states the frame is given to when
 * every run of equal unit (AcousticModel.discriminate, :937-955) is cut into gmm_num slices the way __eq_segment mode 'g'
 * (:614-625) and __get_gmmdata (:629-644) do.  Needs pcl_batch_viterbi first.  Synchronous. */
int pcl_batch_regroup(pcl_batch *b, const int32_t *row_unit, int gmm_num, int32_t *frame_unit, int32_t *frame_k);

/* ----------------------------------------------------------------- unit inventory and label-built batches
 * The reference builds, PER UTTERANCE, one LHMM per label unit (AcousticModel.init_unit / init_parameter,
 * AcousticModel.py:164-240: transmat (S,S), S-2 GMM states between an entry and an exit VirtualState) and glues them
 * into a sentence HMM (AcousticModel.embedded, :957-1014).  Here the inventory is uploaded once:
 *   trans      [n_units][S][S]  unit transition matrices (LHMM.transmat), linear
 *   log_trans  the same through np.log on the caller's side (bit-exact Viterbi is defined on those values; NULL:
 *              this library's libm log is used)
 * Unit i owns the GMM states i*(S-2) .. i*(S-2)+S-3 of the uploaded model (pcl_model_upload with J = n_units*(S-2)). */
int pcl_units_upload(pcl_ctx *ctx, int n_units, int S, const double *trans, const double *log_trans);
int pcl_units_download(pcl_ctx *ctx, double *trans /* [n_units][S][S], after pcl_mstep_transitions */);

/* A7 for U utterances at once: labels = concatenated unit ids, label_len[u] of them per utterance.  Builds what
 * AcousticModel.embedded builds -- N_u = (S-2) L_u + 2 states, the banded transition structure (embedded_transmat
 * :979-989), the row -> GMM state map (embedded_prob :990-1001), uniform pi (embedded_pi :1003-1006; logpi[u] =
 * np.log(1/N_u) from the caller, NULL: libm) -- and keeps the label structure with the batch for pcl_batch_accumulate_hmm.
 * Equivalent to pcl_batch_create + pcl_batch_set_transitions + pcl_batch_set_states on host-built matrices. */
int pcl_batch_create_labels(pcl_ctx *ctx, int U, const int32_t *label_len, const int32_t *labels, const int32_t *T,
                            const int64_t *frame_begin, const double *logpi, pcl_batch **out);
/* Rebuild the batch's transitions from the CURRENT unit inventory (after pcl_mstep_transitions / pcl_em_exchange). */
int pcl_batch_refresh_transitions(pcl_batch *b);

/* A12, HMM half: LHMM.update_acc (LHMM.py:473-500) + LHMM.add_acc (:149-161) for every utterance x label position of
 * the batch: the (S-2,S) block of un-normalised ln xi and the (S-2,) slice of ln gamma (quirk Q5) of each position are
 * log-sum-exp'ed into context-resident per-unit accumulators ksai_acc [n_units][S-2][S], gamma_acc [n_units][S-2]
 * (log domain, initial -inf, LHMM.py:84-85).  pcl_stats_zero resets them too.  Needs pcl_batch_forward_backward. */
int pcl_batch_accumulate_hmm(pcl_batch *b);
int pcl_hmm_acc_zero(pcl_ctx *ctx);
int pcl_hmm_acc_download(pcl_ctx *ctx, double *ksai_acc, double *gamma_acc);
/* A15, transition half: LHMM.update_param (LHMM.py:519-520): transmat[1:-1,:] = exp(ksai_acc - gamma_acc[:,None]) for
 * every unit that occurred; a unit that never occurred keeps its matrix (the reference would write NaN). */
int pcl_mstep_transitions(pcl_ctx *ctx);

/* ----------------------------------------------------------------- decoder (next row f3: the decode half of config 5)
 * The pronunciation tree Lexicon.PronunciationLexicon builds (Lexicon/PronunciationLexicon.py:45-94), flattened: node i
 * spells node_nunits[i] (1 or 2) units node_units[2 i ..] of the uploaded inventory (a reading split at the comma:
 * initial + final, or a lone final: Token.__init__, Decoder.py:225), its children are child_idx[child_ptr[i] ..
 * child_ptr[i+1]) in the tree's insertion order, node_word[i] != 0 where words end, roots = the first-character nodes.
 * Needs pcl_units_upload first; a new pcl_units_upload drops the tree. */
int pcl_lexicon_upload(pcl_ctx *ctx, int n_nodes, const int32_t *node_units, const int32_t *node_nunits, const int32_t *child_ptr,
                       const int32_t *child_idx, const int32_t *node_word, int n_roots, const int32_t *roots);
/* A16: frame-synchronous token passing (Decoder.py:91-167, Token.viterbi :250-288) for every utterance of an ALL-STATE
 * batch (rows entry, GMM state 0 .. J-1, exit; emissions from pcl_batch_score).  beam 0.85 and min_distinct 8 are the
 * reference's pruning rule (:34,:159-167), candidate its `transfer` width (:175); max_tokens bounds the live tokens of one
 * utterance (hand-overs beyond it are dropped and flagged); logpi_* = np.log(1/N) for N = S+0 / 2(S-2)+2 states from the
 * caller.  The reference code is dead: the gaps D1-D5 filled here are listed in
 * oracle/decoder_oracle.py, the restatement this entry point is tested against bit for bit (recursion, pruning, frame loop and in-word
 * hand-over pinned by golden G14 from the reference's own Decoder.py; the completion rules D1-D5 unpinned).
 * Two kernels, the same bits: left-to-right 5-state units (every model the reference builds, AcousticModel.py:176-181) run one
 * lane per token (hmm_decode_lr.hip), any other unit matrices 8 lanes per token (hmm_decode.hip; env PCL_DEC_GENERAL=1 forces it). */
int pcl_batch_decode(pcl_batch *b, double beam, int min_distinct, int candidate, int max_tokens, double logpi_one_unit,
                     double logpi_two_units);
/* Results (NULL pointers are skipped): n_final (U,) tokens returned per utterance; node / score / hist (U, candidate), best
 * first; the word history: hist_n (U,) entries, hist_prev / hist_node (U, Tmax): entry h = (previous entry or -1, node whose
 * word ended); n_tokens (U, Tmax): live tokens after every frame; overflow (U,): 1 if max_tokens was hit. */
int pcl_batch_decode_get(pcl_batch *b, int32_t *n_final, int32_t *node, double *score, int32_t *hist, int32_t *hist_n,
                         int32_t *hist_prev, int32_t *hist_node, int32_t *n_tokens, int32_t *overflow);

/* Copy a result to a caller buffer (layouts in pcl_get_what). */
int pcl_batch_get(pcl_batch *b, int what, void *host);

/* Results on their way to the host WHILE the GPU goes on (the end-to-end protocol moves B / gamma / paths off the
 * device).  pcl_batch_fetch_async queues, behind everything this batch has queued so far (scoring, forward-backward on the second
 * stream, Viterbi), device-to-host copies of the selected results on the library's download stream and returns at once;
 * pcl_batch_fetch_wait blocks until they have landed.  A later compute call on the SAME batch waits for the copies on the device
 * (its buffers are being read), other batches run beside them.  Destinations should be page-locked (pcl_host_alloc): a copy
 * into pageable memory is staged by the runtime and is not asynchronous.  NULL pointers are skipped.  Layouts:
 *   logp   (U,) f64                                  == PCL_GET_LOGP
 *   lgamma ragged, per utterance TIME-MAJOR (T_u, N_u) f64: the transpose of PCL_GET_LGAMMA's (N_u, T_u) -- the device layout,
 *          so that nothing but the copy stands between the kernel and the host (the caller views it transposed);
 *          == l - sum_value of LHMM.update_acc (LHMM.py:486-500), what Clustering.GMM.update_acc is fed
 *   ksai_nz f64, ln xi of the stored transitions     == PCL_GET_KSAI_NZ (LHMM.__ksai, quirk Q5)
 *   path   ragged (T_u,) int32, point (U,) f64       == PCL_GET_PATH / PCL_GET_POINT (needs pcl_batch_viterbi) */
/* sizes of the ragged result arrays of a batch: sum N_u T_u, sum N_u, sum T_u, stored transitions (ln A > -inf).  NULLs are skipped. */
int pcl_batch_sizes(pcl_batch *b, int64_t *sum_nt, int64_t *sum_n, int64_t *sum_t, int64_t *nnz);
int pcl_batch_fetch_async(pcl_batch *b, double *logp, double *lgamma_tm, double *ksai_nz, int32_t *path, double *point);
int pcl_batch_fetch_wait(pcl_batch *b);

/* The clock the shader engines actually hold, measured on the device: one wavefront on the library's auxiliary stream reads the
 * shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) `spin_us` microseconds apart, beside
 * whatever the other streams are running.  rocm-smi's sclk is the REQUESTED level; under the matrix-pipe kernels the chip holds
 * 1.6-1.8 GHz of its 2.4.  Synchronous (waits for the probe only). */
int pcl_clock_probe(pcl_ctx *ctx, int spin_us, double *shader_mhz);

/* ----------------------------------------------------------------- E-step statistics
 * A13: Clustering.GMM.update_acc (Clustering.py:653-680) for every (utterance, emitting row) of the
 * batch, summed into ctx-resident per-state statistics.  The reference keeps them in the log domain
 * per label position and merges files later (Clustering.py:314-367); here they are LINEAR sums
 *   acc[j,m]      = sum_t gamma_t(j,m)                      (exp of GMM.acc)
 *   alpha_acc[j]  = sum_t gamma_t(j)                        (exp of GMM.alpha_acc)
 *   mean_acc[j,m,d] = sum_t gamma_t(j,m) (o_td + bias)      (exp of GMM.mean_acc, bias = 100)
 *   cov_acc[j,m,d]  = sum_t gamma_t(j,m) (o_td - mu_jmd)^2  (exp of GMM.__covariance_acc)
 * which is what RCCL can sum.  Needs pcl_batch_forward_backward first. */
int pcl_stats_zero(pcl_ctx *ctx);
int pcl_batch_accumulate(pcl_batch *b, int precision);
/* J*M, J, J*M*D, J*M*D doubles */
/* Approximate mode of pcl_batch_accumulate (off by default): (frame, state) pairs with gamma_t(j) < 2^log2_threshold are
 * left out of the statistics.  The default leaves out only pairs whose every term is EXACTLY zero in the kernel's arithmetic
 * (2^-150 in f32, 2^-1076 in f64), which changes no bit; a threshold such as -40 drops contributions below 1e-12 of a frame
 * -- far inside the f32 path's own rounding -- and shortens the pass when the posteriors are flat.  The reference has no
 * such cut (it sums everything in the log domain, Clustering.py:653-680). */
int pcl_accumulate_prune(pcl_ctx *ctx, double log2_threshold);

int pcl_stats_download(pcl_ctx *ctx, double *acc, double *alpha_acc, double *mean_acc, double *cov_acc);

/* A15: Clustering.GMM.update_param (Clustering.py:682-693) for every state, on the device, from the resident
 * (all-reduced) statistics: w = acc/alpha_acc, mu = mean_acc/acc - bias, var = max(cov_acc/acc, c_covariance);
 * then every scoring layout is rebuilt, so the next E-step can start without leaving the GPU.
 * pcl_model_download returns the float64 master copy (J*M*D, J*M*D, J*M; NULL pointers are skipped). */
int pcl_mstep(pcl_ctx *ctx, double c_covariance);
int pcl_model_download(pcl_ctx *ctx, double *mean, double *var, double *weight);

/* Numerical guard of the f32 matrix-core path.  The MFMA kernels evaluate the Gaussian exponent in a form expanded
 * around a per-state centre c_j; its f32 rounding error grows with cond[j] = max_m log2(e) * sum_d (mu_jmd - c_jd)^2 /
 * (2 var_jmd) (about 5e-7 * cond nats).  States with cond[j] > *cond_max (default 96, env PCL_MFMA_COND_MAX) are scored
 * and accumulated by the direct-form (x-mu)^2 kernels instead, so util.py:78-88's result keeps its 1e-4 tolerance on
 * any model.  cond: J floats (may be NULL); cond_max: 1 float (may be NULL).  Recomputed by upload and by pcl_mstep. */
int pcl_model_conditioning(pcl_ctx *ctx, float *cond, float *cond_max);

/* Split states (round 4).  The limit above is a property of single mixtures: one tight mixture far from the state's centre
 * used to send its whole state (2048 mixtures) to the direct-form kernels, and after an M-step nearly every state has a few.
 * A mixture whose own term cond_jm exceeds cond_max is now taken out of the matrix-core layouts instead (it looks like a
 * zero-weight mixture there and does not enter the state's feature scales or cond[j]); the direct-form kernels evaluate the
 * state's list of such mixtures and the two parts are merged -- ln(e^a + e^b) of the two partial log-sum-exps in scoring,
 * += into the same statistics in the accumulate pass -- so the result is the reference's sum over all mixtures
 * (Clustering.py:740-767, :653-680) whichever kernel evaluated a term.  A state leaves the matrix cores as a whole only when
 * more than *limit of its mixtures are out.  Round 6: in SCORING the list is no longer evaluated in direct form but by the coarse pass
 * (csrc/gmm_score_coarse.hip: a bound of each off-pipe mixture computed on the matrix pipe rules out almost every (frame, mixture) pair,
 * the pairs it cannot rule out are evaluated in direct form), and *limit is 0.99 M there (env PCL_COARSE_SPLIT_MAX; PCL_COARSE=0:
 * direct form, limit 0.5 M).  The accumulate pass keeps direct form and 0.5 M.  env PCL_SPLIT_MAX = share of M sets both limits;
 * 0 = whole states, as before round 4.  *limit reports the scoring limit.
 * n_off: J ints, off-pipe mixtures per state (may be NULL); limit: 1 int (may be NULL). */
int pcl_model_split_info(pcl_ctx *ctx, int *n_off, int *limit);
/* Diagnostics of the coarse pass over the off-pipe mixtures (csrc/gmm_score_coarse.hip; counted only under env PCL_COARSE_STATS=1):
 * *pairs = (frame, mixture) pairs evaluated in direct form since the last reset -- the pairs the bound on the matrix pipe could not rule
 * out; every other pair of an off-pipe mixture was proven to lie 2^-36 below its frame's likelihood.  reset != 0 clears the count. */
int pcl_coarse_counter(pcl_ctx *ctx, unsigned long long *pairs, int reset);
/* ... and *tiles_given_up (may be NULL) = tiles of 256 frames x one state on which the pass gave up -- a wave had evaluated more than
 * max(4096, 2 x the state's off-pipe mixtures) pairs: the state's on-pipe part is no reference for those frames -- and which the direct-form
 * subset kernel rescored in the same call (as it does tiles with a feature out of the f16 range).  Same results either way. */
int pcl_coarse_counters(pcl_ctx *ctx, unsigned long long *pairs, unsigned long long *tiles_given_up, int reset);

/* How much of a CU the matrix-core scoring kernel takes.  0 (default): three workgroups per CU, the fastest for the kernel alone.
 * 2: two -- a third of the registers stays free for kernels of OTHER streams, which is what lets the token passing of chunk k-1
 * (pcl_batch_decode, second stream) run beside the scoring of chunk k in a streamed decode (Decoder.py has no such loop: the
 * reference decodes utterance by utterance, Decoder.py:146-187).  Same kernel, same results; applies to the launches that follow. */
int pcl_score_occupancy(pcl_ctx *ctx, int workgroups_per_cu);

/* ----------------------------------------------------------------- MFCC front-end (next row f4: the step before the path)
 * AudioProcessing.MFCC.mfcc (StatisticalModel/AudioProcessing.py:416-448) for U signals at once, float64:
 * pre-emphasis 0.98 (:184), framing sampletime/overlap (:201), per-FRAME window factor (:228, as the reference
 * computes it), |rFFT_nfft| (:250), mel filter bank + frame energy (:279), ln + DCT (:347), c0 <- ln(energy)
 * (flags bit0), deltas / delta-deltas over +-2 frames (flags bit1 / bit2, :401).  signal = concatenated samples,
 * sig_off[U+1]; twiddle_cos/sin[nfft], mel_response[filterbanks][nfft/2+1] and dct_matrix[rank][filterbanks] are
 * built by the caller (poccala_amd/StatisticalModel/AudioProcessing.py) so the reference's filter and DCT
 * conventions are defined in one place.  out: out_rows x (rank * {1,2,3}) row-major, out_rows = total frames. */
int pcl_mfcc(pcl_ctx *ctx, int U, const double *signal, const int64_t *sig_off, int framerate, double sampletime,
             double overlap, int nfft, int filterbanks, int rank, int flags, const double *twiddle_cos,
             const double *twiddle_sin, const double *mel_response, const double *dct_matrix, double *out,
             int64_t out_rows);

/* ----------------------------------------------------------------- multi-GPU (RCCL over xGMI)
 * Replaces the reference's file-based accumulator merge (LHMM.py:256-290, Clustering.py:314-367).
 * id_bytes is a 128-byte ncclUniqueId made by rank 0 and distributed by the caller. */
int pcl_comm_unique_id(void *id_bytes128);
int pcl_comm_init(pcl_ctx *ctx, int rank, int nranks, const void *id_bytes128);
/* Rehearsal transport for several ranks on ONE device (RCCL refuses that: "Duplicate GPU detected"): every collective
 * becomes an all-gather of host bytes through `fn` (returns 0 on success; recv_all = nranks * bytes, in rank order).
 * Same orchestration code as the RCCL path; refuses exchanges beyond 256 MiB.  Not a production path. */
typedef int (*pcl_allgather_fn)(void *user, const void *send, size_t bytes, void *recv_all);
int pcl_comm_init_host(pcl_ctx *ctx, int rank, int nranks, pcl_allgather_fn fn, void *user);
/* transport: 0 none, 1 RCCL, 2 host rehearsal; rccl_nranks = ncclCommCount (0 unless transport 1).  NULLs are skipped. */
int pcl_comm_info(pcl_ctx *ctx, int *rank, int *nranks, int *transport, int *rccl_nranks);
/* Sum all-reduce of all GMM statistics (f64) + log-sum-exp merge of the per-unit HMM accumulators: afterwards every
 * rank holds the global statistics (then pcl_mstep on every rank).  Kept for parity checks; the E-step uses: */
int pcl_stats_allreduce(pcl_ctx *ctx);
/* The E-step exchange + M-step: reduce-scatter of the GMM statistics by state range -> GMM.update_param
 * (Clustering.py:682-693) on the owned J/nranks states -> all-gather of (mean, var, weight) -> every layout re-derived;
 * the per-unit HMM accumulators are merged by max + sum all-reduces and, if update_transitions, LHMM.update_param's
 * transition update (LHMM.py:519-520) follows.  payload: PCL_F64, or PCL_F32 = half the bytes on the wire (sums and
 * parameters rounded to f32 in flight; every rank continues from the same rounded model).  With one rank and no
 * communicator this is pcl_mstep (+ pcl_mstep_transitions).  Synchronous at return. */
int pcl_em_exchange(pcl_ctx *ctx, double c_covariance, int payload, int update_transitions);
/* The LAST accumulate pass of an E-step and the exchange as one call, pipelined: the states are cut into n_chunks equal
 * chunks, and as soon as the accumulate pass (which walks the states in ascending order, a group at a time) is done with a
 * chunk, the chunk goes through reduce-scatter (inside a chunk rank r owns the r-th slice) -> GMM.update_param -> all-gather ->
 * its layouts re-derived, on a stream of its own, beside the accumulation of the later states.  Replaces, like pcl_em_exchange,
 * the reference's file merge + per-unit M-step (LHMM.py:256-290, Clustering.py:314-367,682-693; AcousticModel.py:918-935);
 * same sums and M-step arithmetic, so the gathered model equals pcl_batch_accumulate + pcl_em_exchange (bit for bit on
 * the rehearsal transport; RCCL's ring order may differ in the last bit).  Default (env PCL_PIPE_MODE=1): only a chunk's
 * reduce-scatter leaves early, M-steps / all-gathers / derive run at the end; PCL_PIPE_MODE=0: the whole chain per chunk (on one
 * rank: M-step + derive of finished chunks beside the rest of the pass).  Synchronous at return. */
int pcl_batch_accumulate_exchange(pcl_batch *b, int precision, double c_covariance, int payload, int update_transitions, int n_chunks);
/* The same exchange for a rank that has NO batch for this last pass (fewer batches on this rank than on the others: every rank must run
 * the same sequence of collectives): the rank's statistics -- zero, or whatever its earlier batches accumulated -- go through the
 * n_chunks chunk exchanges of pcl_batch_accumulate_exchange without an accumulate pass in front.  Same arguments, same result. */
int pcl_accumulate_exchange_idle(pcl_ctx *ctx, double c_covariance, int payload, int update_transitions, int n_chunks);
/* Of the last pcl_batch_accumulate_exchange: its number of chunks and how many of them left for the exchange WHILE the accumulate pass was
 * still running (0: the pass released none -- states out of ascending order or on the direct-form kernel -- and the call was the plain
 * accumulate + exchange).  NULLs are skipped. */
int pcl_pipe_info(pcl_ctx *ctx, int *chunks, int *released_early);
int pcl_comm_destroy(pcl_ctx *ctx);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* POCCALA_HIP_H */

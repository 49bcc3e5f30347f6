// hmm_units.hip -- the unit inventory on the device, sentence HMMs built from labels, and the per-unit transition
// accumulators (SURVEY.md section 8a rows A7, A12 and the transition half of A15).
//
// Replaces, for a whole batch of utterances at once:
//   AcousticModel.embedded            AcousticModel/AcousticModel.py:957-1014   sentence HMM = concatenated unit HMMs
//   LHMM.update_acc (HMM half)        StatisticalModel/LHMM.py:473-500          slices of xi / gamma per label position
//   LHMM.add_acc                      StatisticalModel/LHMM.py:149-161          log-domain merge into ksai_acc / gamma_acc
//   LHMM.update_param (transitions)   StatisticalModel/LHMM.py:509-520          A[1:-1,:] = exp(ksai_acc - gamma_acc)
// In the reference these are Python loops over utterances x label positions (164 k iterations per E-step at BASELINE
// config 4) and a file merge; here the label structure is kept with the batch, one wave group per unit folds the
// un-normalised ln xi / ln gamma (quirk Q5) of all its occurrences with a fixed-order log-sum-exp, and the result is
// log-added to context-resident accumulators that pcl_stats_allreduce / pcl_em_exchange merge across GPUs.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "pcl_internal.h"

namespace {

constexpr int ACC_WAVES = 4;

// One workgroup per unit, lane v <-> one accumulator entry: v < e*S is ksai_acc[k = v / S][c = v % S], then e entries
// of gamma_acc.  The ACC_WAVES waves stride over the unit's occurrences (independent loads), each lane keeps an online
// log-sum-exp, the partial (max, sum) pairs are combined in wave order: the result does not depend on timing.
__global__ void hmm_acc_kernel(const UttDesc *__restrict__ utts, const double *__restrict__ ksai,
                               const double *__restrict__ gamma_out, const int *__restrict__ occ_ptr,
                               const int *__restrict__ occ_utt, const int *__restrict__ occ_row0, int e, int S,
                               double *__restrict__ acc_ksai, double *__restrict__ acc_gamma) {
    __shared__ double pm[ACC_WAVES][64], ps[ACC_WAVES][64];
    const int unit = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nk = e * S, nv = nk + e;
    const int o0 = occ_ptr[unit], o1 = occ_ptr[unit + 1];
    double m = -INFINITY, s = 0.0;
    if (lane < nv) {
        const bool is_k = lane < nk;
        const int k = is_k ? lane / S : lane - nk, c = is_k ? lane % S : 0;
        for (int o = o0 + wave; o < o1; o += ACC_WAVES) {
            const UttDesc d = utts[occ_utt[o]];
            const int i0 = occ_row0[o];                    // first emitting row of this label position: 1 + pos * e
            // LHMM.py:494-496: ksai[1:-1][y:y+e, x:x+S] with y = x = pos*e  ->  rows i0 + k, columns i0 - 1 + c
            const double x = is_k ? ksai[d.mat_off + (long long)(i0 + k) * d.N + (i0 - 1 + c)] : gamma_out[d.vec_off + i0 + k];
            if (x > m) {
                s = s * exp(m - x) + 1.0;                  // exp(-inf) = 0 on the first finite value
                m = x;
            } else if (x > -INFINITY) {
                s += exp(x - m);
            }
        }
    }
    pm[wave][lane] = m;
    ps[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && lane < nv) {
        double M = pm[0][lane];
        for (int w = 1; w < ACC_WAVES; ++w) M = fmax(M, pm[w][lane]);
        double v = -INFINITY;
        if (M > -INFINITY) {
            double t = 0.0;
            for (int w = 0; w < ACC_WAVES; ++w)
                if (pm[w][lane] > -INFINITY) t += ps[w][lane] * exp(pm[w][lane] - M);
            v = M + log(t);
        }
        // add_acc (LHMM.py:149-161): elementwise log-add with the running accumulator (util.log_sum_exp, quirk Q4)
        double *dst = (lane < nk) ? acc_ksai + (size_t)unit * nk + lane : acc_gamma + (size_t)unit * e + (lane - nk);
        const double a = *dst;
        const double top = fmax(a, v);
        *dst = (top == -INFINITY) ? -INFINITY : top + log(exp(a - top) + exp(v - top));
    }
}

__global__ void fill_kernel(double *__restrict__ p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// cross-rank log-sum-exp of the accumulators in three steps around two tiny all-reduces (max, then sum)
__global__ void merge_scale_kernel(double *__restrict__ acc, const double *__restrict__ top, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) acc[i] = (top[i] == -INFINITY) ? 0.0 : exp(acc[i] - top[i]);
}
__global__ void merge_finish_kernel(double *__restrict__ acc, const double *__restrict__ top, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) acc[i] = (top[i] == -INFINITY) ? -INFINITY : top[i] + log(acc[i]);
}

// LHMM.update_param, transition half (LHMM.py:519-520): A[1:-1, :] = exp(ksai_acc - gamma_acc[:, None]); rows 0 and S-1 stay.
// A row whose gamma_acc is -inf (the unit never occurred) keeps its transitions: exp(-inf - -inf) would be NaN.
__global__ void trans_mstep_kernel(const double *__restrict__ acc_ksai, const double *__restrict__ acc_gamma, int n_units,
                                   int e, int S, double *__restrict__ trans) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_units * e * S) return;
    const int c = i % S, k = (i / S) % e, u = i / (S * e);
    const double g = acc_gamma[(size_t)u * e + k];
    if (!(g > -INFINITY)) return;
    trans[((size_t)u * S + 1 + k) * S + c] = exp(acc_ksai[i] - g);
}

}  // namespace

// The decoder's device copy of ln A (hmm_decode.hip) follows every change of the unit transitions: a transition M-step or a new
// inventory of the same shape.  Waits for a decoder that may still be reading the old values on the second stream.
static int lexicon_refresh_logtrans(pcl_ctx *ctx) {
    if (!ctx->lex_nodes || !ctx->d_unit_logtrans) return PCL_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_dp));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_unit_logtrans, ctx->unit_logtrans.data(), ctx->unit_logtrans.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

static void units_free(pcl_ctx *ctx) {
    dev_free(ctx->d_unit_trans);
    dev_free(ctx->hmm_ksai);
    ctx->hmm_gamma = nullptr;                        // inside the hmm_ksai allocation
    ctx->unit_trans.clear();
    ctx->unit_logtrans.clear();
}

void pcl_units_release(pcl_ctx *ctx) {
    pcl_lexicon_release(ctx);                        // the tree names units of this inventory: upload it again after a DIFFERENT inventory
    dev_free(ctx->d_unit_trans);
    dev_free(ctx->hmm_ksai);
    ctx->hmm_gamma = nullptr;                        // inside the hmm_ksai allocation
    ctx->unit_trans.clear();
    ctx->unit_logtrans.clear();
    ctx->n_units = ctx->S = 0;
}

void pcl_batch_units_release(pcl_batch *b) {
    dev_free(b->occ_ptr);
    dev_free(b->occ_utt);
    dev_free(b->occ_row0);
}

static size_t hmm_acc_len(const pcl_ctx *ctx) { return (size_t)ctx->n_units * (ctx->S - 2) * (ctx->S + 1); }

int pcl_launch_hmm_acc_merge_prepare(pcl_ctx *ctx, double *top) {
    HIPCHK(ctx, hipMemcpyAsync(top, ctx->hmm_ksai, hmm_acc_len(ctx) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return PCL_OK;
}
int pcl_launch_hmm_acc_merge_scale(pcl_ctx *ctx, const double *top) {
    const size_t n = hmm_acc_len(ctx);
    hipLaunchKernelGGL(merge_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hmm_ksai, top, n);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}
int pcl_launch_hmm_acc_merge_finish(pcl_ctx *ctx, const double *top) {
    const size_t n = hmm_acc_len(ctx);
    hipLaunchKernelGGL(merge_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hmm_ksai, top, n);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

static int hmm_acc_reset(pcl_ctx *ctx) {
    const size_t n = hmm_acc_len(ctx);
    if (n == 0) return PCL_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, ctx->stream, ctx->hmm_ksai, n, -INFINITY);   // LHMM.py:84-85
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

// refresh the host copies after the device wrote new transitions; ln A is taken by the host libm
static int units_pull(pcl_ctx *ctx) {
    const size_t n = ctx->unit_trans.size();
    HIPCHK(ctx, hipMemcpyAsync(ctx->unit_trans.data(), ctx->d_unit_trans, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) ctx->unit_logtrans[i] = log(ctx->unit_trans[i]);
    return lexicon_refresh_logtrans(ctx);            // a resident pronunciation tree decodes with the NEW transitions
}

int pcl_launch_trans_mstep(pcl_ctx *ctx) {
    const int e = ctx->S - 2, n = ctx->n_units * e * ctx->S;
    hipLaunchKernelGGL(trans_mstep_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->hmm_ksai, ctx->hmm_gamma, ctx->n_units, e,
                       ctx->S, ctx->d_unit_trans);
    HIPCHK(ctx, hipGetLastError());
    return units_pull(ctx);
}

// Sparse structure and ln pi of every sentence HMM of a label-built batch, from the CURRENT unit transitions
// (AcousticModel.embedded: embedded_transmat :979-989, embedded_pi :1003-1006).
static int build_sentences(pcl_batch *b) {
    pcl_ctx *ctx = b->ctx;
    const int S = ctx->S, e = S - 2;
    std::vector<int> row_ptr((size_t)b->sumN + b->U), col_ptr((size_t)b->sumN + b->U);
    std::vector<int> col_idx, row_idx;
    std::vector<double> csr_val, csc_val, logpi((size_t)b->sumN);
    b->max_outdeg = b->max_indeg = 0;
    size_t lo = 0;
    std::vector<int> indeg, fillp;
    for (int u = 0; u < b->U; ++u) {
        UttDesc &d = b->utt[u];
        const int N = d.N, L = b->label_len[u];
        const int32_t *lab = b->labels.data() + lo;
        lo += L;
        d.nnz_off = (int)col_idx.size();
        const size_t base = col_idx.size();
        int cnt = 0;
        for (int i = 0; i < N; ++i) {
            row_ptr[d.ptr_off + i] = cnt;
            if (i < N - 1) {                                     // the exit row has no successors
                // row 0: unit 0's entry row at columns 0..S-1; row 1 + p e + k: row 1 + k of unit lab[p] at columns p e + c
                const int p = (i == 0) ? 0 : (i - 1) / e, r = (i == 0) ? 0 : 1 + (i - 1) % e;
                const double *row = ctx->unit_logtrans.data() + ((size_t)lab[p] * S + r) * S;
                for (int c = 0; c < S; ++c)
                    if (!(row[c] == -INFINITY)) {
                        col_idx.push_back(p * e + c);
                        csr_val.push_back(row[c]);
                        ++cnt;
                    }
            }
            b->max_outdeg = std::max(b->max_outdeg, cnt - row_ptr[d.ptr_off + i]);
        }
        row_ptr[d.ptr_off + N] = cnt;
        // CSC by a counting sort of the CSR entries: sources come out in ascending order
        indeg.assign(N + 1, 0);
        for (int k = 0; k < cnt; ++k) ++indeg[col_idx[base + k] + 1];
        for (int j = 0; j < N; ++j) {
            b->max_indeg = std::max(b->max_indeg, indeg[j + 1]);
            indeg[j + 1] += indeg[j];
        }
        for (int j = 0; j <= N; ++j) col_ptr[d.ptr_off + j] = indeg[j];
        fillp.assign(indeg.begin(), indeg.end() - 1);
        row_idx.resize(base + cnt);
        csc_val.resize(base + cnt);
        for (int i = 0; i < N; ++i)
            for (int k = row_ptr[d.ptr_off + i]; k < row_ptr[d.ptr_off + i + 1]; ++k) {
                const int j = col_idx[base + k], q = fillp[j]++;
                row_idx[base + q] = i;
                csc_val[base + q] = csr_val[base + k];
            }
        for (int i = 0; i < N; ++i) logpi[d.vec_off + i] = b->logpi_u[u];
        if (col_idx.size() > 0x7fffffffULL) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create_labels: too many transitions");
    }
    return pcl_batch_upload_sparse(b, row_ptr, col_idx, csr_val, col_ptr, row_idx, csc_val, logpi.data());
}

extern "C" {

int pcl_units_upload(pcl_ctx *ctx, int n_units, int S, const double *trans, const double *log_trans) {
    if (!ctx) return PCL_ERR_INVALID;
    if (n_units <= 0 || S < 3 || S > 8 || !trans) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_units_upload: bad arguments (n_units=%d, S=%d; 3 <= S <= 8)", n_units, S);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // an inventory of the same shape keeps a resident pronunciation tree (its unit ids stay valid): only the transitions change
    const bool keep_tree = ctx->lex_nodes && n_units == ctx->n_units && S == ctx->S;
    if (keep_tree) units_free(ctx);
    else pcl_units_release(ctx);
    const size_t n = (size_t)n_units * S * S;
    ctx->unit_trans.assign(trans, trans + n);
    ctx->unit_logtrans.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (!(trans[i] >= 0.0)) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_units_upload: transition %zu = %g is negative or NaN", i, trans[i]);
        ctx->unit_logtrans[i] = log_trans ? log_trans[i] : log(trans[i]);
    }
    ctx->n_units = n_units;
    ctx->S = S;
    TRY(dev_alloc(ctx, &ctx->d_unit_trans, n));
    HIPCHK(ctx, hipMemcpy(ctx->d_unit_trans, trans, n * sizeof(double), hipMemcpyHostToDevice));
    // ksai_acc then gamma_acc in ONE allocation, so that the cross-rank merge is one pair of all-reduces
    TRY(dev_alloc(ctx, &ctx->hmm_ksai, hmm_acc_len(ctx)));
    ctx->hmm_gamma = ctx->hmm_ksai + (size_t)n_units * (S - 2) * S;
    TRY(hmm_acc_reset(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return lexicon_refresh_logtrans(ctx);
}

int pcl_units_download(pcl_ctx *ctx, double *trans) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->n_units) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_units_download: no units uploaded");
    if (!trans) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_units_download: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(trans, ctx->unit_trans.data(), ctx->unit_trans.size() * sizeof(double));
    return PCL_OK;
}

int pcl_batch_create_labels(pcl_ctx *ctx, int U, const int32_t *label_len, const int32_t *labels, const int32_t *T,
                            const int64_t *frame_begin, const double *logpi, pcl_batch **out) {
    if (!ctx || !out) return PCL_ERR_INVALID;
    *out = nullptr;
    if (!ctx->n_units) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_create_labels: pcl_units_upload first");
    if (ctx->J == 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_create_labels: pcl_model_upload first");
    const int S = ctx->S, e = S - 2;
    if (ctx->J != ctx->n_units * e)
        PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_create_labels: the model has %d GMM states, %d units x %d emitting states need %d", ctx->J, ctx->n_units, e, ctx->n_units * e);
    if (U <= 0 || !label_len || !labels || !T || !frame_begin) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create_labels: bad arguments (U=%d)", U);
    std::vector<int32_t> N(U);
    size_t tot = 0;
    for (int u = 0; u < U; ++u) {
        if (label_len[u] < 1) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create_labels: utterance %d has an empty label", u);
        N[u] = e * label_len[u] + 2;                               // AcousticModel.py:966
        tot += label_len[u];
    }
    for (size_t i = 0; i < tot; ++i)
        if (labels[i] < 0 || labels[i] >= ctx->n_units) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create_labels: unit id %d outside [0,%d)", labels[i], ctx->n_units);
    pcl_batch *b = nullptr;
    TRY(pcl_batch_create(ctx, U, N.data(), T, frame_begin, &b));
    pcl_desc_group uploads(ctx);                                   // every descriptor array below: staged, one wait at the end
    b->from_labels = true;
    b->label_len.assign(label_len, label_len + U);
    b->labels.assign(labels, labels + tot);
    b->logpi_u.resize(U);
    for (int u = 0; u < U; ++u) b->logpi_u[u] = logpi ? logpi[u] : log(1.0 / N[u]);
    // row -> GMM state (embedded_prob, AcousticModel.py:990-1001) and unit -> occurrences
    std::vector<int32_t> row_state((size_t)b->sumN);
    std::vector<int> occ_ptr(ctx->n_units + 1, 0);
    for (size_t i = 0; i < tot; ++i) ++occ_ptr[labels[i] + 1];
    for (int k = 0; k < ctx->n_units; ++k) occ_ptr[k + 1] += occ_ptr[k];
    std::vector<int> occ_utt(tot), occ_row0(tot), fillp(occ_ptr.begin(), occ_ptr.end() - 1);
    size_t lo = 0;
    for (int u = 0; u < U; ++u) {
        const UttDesc &d = b->utt[u];
        row_state[d.vec_off] = PCL_ROW_ENTRY;
        row_state[d.vec_off + d.N - 1] = PCL_ROW_EXIT;
        for (int p = 0; p < label_len[u]; ++p) {
            const int unit = labels[lo + p];
            for (int k = 0; k < e; ++k) row_state[d.vec_off + 1 + p * e + k] = unit * e + k;
            const int q = fillp[unit]++;
            occ_utt[q] = u;
            occ_row0[q] = 1 + p * e;
        }
        lo += label_len[u];
    }
    int rc = pcl_batch_set_states_impl(b, row_state.data());
    if (rc == PCL_OK) rc = build_sentences(b);
    b->n_occ = (int)tot;
    if (rc == PCL_OK) rc = dev_alloc(ctx, &b->occ_ptr, occ_ptr.size());
    if (rc == PCL_OK) rc = dev_alloc(ctx, &b->occ_utt, tot);
    if (rc == PCL_OK) rc = dev_alloc(ctx, &b->occ_row0, tot);
    if (rc == PCL_OK && (pcl_h2d_fresh(ctx, b->occ_ptr, occ_ptr.data(), occ_ptr.size() * sizeof(int)) != hipSuccess ||
                         pcl_h2d_fresh(ctx, b->occ_utt, occ_utt.data(), tot * sizeof(int)) != hipSuccess ||
                         pcl_h2d_fresh(ctx, b->occ_row0, occ_row0.data(), tot * sizeof(int)) != hipSuccess)) {
        pcl_set_error(ctx, "pcl_batch_create_labels: copy failed");
        rc = PCL_ERR_HIP;
    }
    if (rc == PCL_OK && uploads.finish() != hipSuccess) {
        pcl_set_error(ctx, "pcl_batch_create_labels: copy failed");
        rc = PCL_ERR_HIP;
    }
    if (rc == PCL_OK && ctx->stream_desc && pcl_fewer_markers()) {
        // the constant entry / exit rows of the emission matrix (AcousticModel.py:218-219), here instead of in front of the batch's first
        // scoring launch: on the descriptor stream, beside whatever the main stream is doing (the buffer is fresh, nobody reads it yet)
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream_desc;
        rc = pcl_launch_fill_virtual_rows(ctx, b);
        ctx->stream = main_stream;
        if (rc == PCL_OK && hipStreamSynchronize(ctx->stream_desc) != hipSuccess) {
            pcl_set_error(ctx, "pcl_batch_create_labels: fill failed");
            rc = PCL_ERR_HIP;
        }
        if (rc == PCL_OK) b->virt_rows_filled = true;
    }
    if (rc != PCL_OK) {
        const std::string keep = ctx->err;
        (void)uploads.finish();
        pcl_batch_destroy(b);
        ctx->err = keep;
        return rc;
    }
    *out = b;
    return PCL_OK;
}

int pcl_batch_refresh_transitions(pcl_batch *b) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!b->from_labels) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_refresh_transitions: the batch was not created from labels");
    if (ctx->J != ctx->n_units * (ctx->S - 2) || ctx->J != b->model_J) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_refresh_transitions: the unit inventory changed since the batch was created");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_dp));
    b->dp_pending = false;
    return build_sentences(b);
}

int pcl_batch_accumulate_hmm(pcl_batch *b) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!b->from_labels) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_hmm: the batch was not created from labels (pcl_batch_create_labels)");
    if (!b->have_fb) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_hmm: run pcl_batch_forward_backward first");
    if (!ctx->hmm_ksai || (int)b->label_len.size() != b->U) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_hmm: no units uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (b->dp_pending) {                                           // the recursion ran on the second stream
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b->ev_dp, 0));
        b->dp_pending = false;
    }
    const int e = ctx->S - 2;
    pcl_timer_begin(ctx, "hmm_acc");
    hipLaunchKernelGGL(hmm_acc_kernel, dim3(ctx->n_units), dim3(64 * ACC_WAVES), 0, ctx->stream, b->d_utt, b->ksai, b->gamma_out, b->occ_ptr,
                       b->occ_utt, b->occ_row0, e, ctx->S, ctx->hmm_ksai, ctx->hmm_gamma);
    pcl_timer_end(ctx, "hmm_acc");
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, pcl_batch_mark(b));
    return PCL_OK;
}

int pcl_hmm_acc_zero(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_hmm_acc_zero: no units uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return hmm_acc_reset(ctx);
}

int pcl_hmm_acc_download(pcl_ctx *ctx, double *ksai_acc, double *gamma_acc) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_hmm_acc_download: no units uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nk = (size_t)ctx->n_units * (ctx->S - 2) * ctx->S, ng = (size_t)ctx->n_units * (ctx->S - 2);
    if (ksai_acc) HIPCHK(ctx, hipMemcpyAsync(ksai_acc, ctx->hmm_ksai, nk * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (gamma_acc) HIPCHK(ctx, hipMemcpyAsync(gamma_acc, ctx->hmm_ksai + nk, ng * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_mstep_transitions(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_mstep_transitions: no units uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return pcl_launch_trans_mstep(ctx);
}

}  // extern "C"

// pcl_comm.hip -- the one exchange of the path: the E-step statistics of the GPUs of a node, over RCCL / xGMI.
//
// Replaces the reference's file-based accumulator merge and the per-unit M-step that follows it
// (StatisticalModel/LHMM.py:256-290, StatisticalModel/Clustering.py:314-367; AcousticModel.multi_embedded_training_2,
// AcousticModel/AcousticModel.py:918-935): there every worker np.save()s log-domain accumulators, a reducer
// log-sum-exps the files unit by unit and re-estimates that unit.  Here:
//
//   pcl_em_exchange   reduce-scatter of the GMM statistics by STATE RANGE (rank r owns states [r J/n, (r+1) J/n))
//                     -> GMM.update_param on the owned states only -> all-gather of the new (mean, var, weight)
//                     -> every rank re-derives its scoring layouts.
//                     Same bytes on the wire as an all-reduce, but the M-step is 1/n of the work per GPU and what
//                     comes back is the next iteration's model.  xGMI is point to point: each rank receives n-1 shards
//                     of J/n states concurrently over its n-1 links.  payload PCL_F64 (parity) or PCL_F32 (half the
//                     bytes: sums and parameters rounded to f32 on the wire, f64 on both sides of it).
//   pcl_stats_allreduce   the plain sum all-reduce of everything (every rank then holds the global statistics).
// The per-unit transition accumulators are un-normalised LOG values (quirk Q5): their merge is a log-sum-exp =
// max all-reduce, exp(x - max), sum all-reduce, log (tiny: units x 18 doubles), done by both entry points.
//
// Transport: RCCL (pcl_comm_init).  pcl_comm_init_host is a rehearsal transport for several ranks on ONE device --
// RCCL refuses that ("Duplicate GPU detected") -- where every collective goes through a caller-supplied all-gather
// of host bytes, in chunks of 64 MiB per rank (the C4-shape exchange, 3.9 GB of statistics, goes through in
// 64-MiB pieces); it runs the same orchestration code at the speed of the callback (a TCP hub in bench.py: seconds, not ms).
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pcl_internal.h"

namespace {

// bytes per rank and call of the host callback (PCL_HOST_CHUNK_KB: the tests use 1 KiB so that chunk edges fall inside states)
static size_t host_chunk_bytes() {
    const char *e = getenv("PCL_HOST_CHUNK_KB");
    return e && atol(e) > 0 ? (size_t)atol(e) << 10 : (size_t)64 << 20;
}

__global__ void to_f32_kernel(const double *__restrict__ src, float *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void to_f64_kernel(const float *__restrict__ src, double *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

// mean_acc = sum g (o + bias) sits near (100 + mu) acc: rounded to f32 as it is, the bias would turn a 6e-8 relative
// rounding into 6e-6 absolute on the re-estimated mean.  On the wire travels  sum g (o - c_j) = mean_acc - (bias + c_j) acc
// (c_j = the state's expansion centre), which is of the size of the features' spread; the owner adds the term back in f64.
__global__ void mean_to_f32_kernel(const double *__restrict__ st_mean, const double *__restrict__ st_acc, const float *__restrict__ centers,
                                   int Mpad, int D, double bias, float *__restrict__ dst, size_t first, size_t n) {
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const size_t i = first + k, jm = i / D;
        const int d = (int)(i - jm * D);
        const size_t j = jm / Mpad;
        dst[i] = (float)(st_mean[i] - (bias + (double)centers[j * D + d]) * st_acc[jm]);
    }
}
__global__ void mean_from_f32_kernel(const float *__restrict__ src, const double *__restrict__ st_acc, const float *__restrict__ centers,
                                     int Mpad, int D, double bias, double *__restrict__ st_mean, size_t lo, size_t cnt) {
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < cnt; k += (size_t)gridDim.x * blockDim.x) {
        const size_t i = lo + k, jm = i / D;
        const int d = (int)(i - jm * D);
        const size_t j = jm / Mpad;
        st_mean[i] = (double)src[i] + (bias + (double)centers[j * D + d]) * st_acc[jm];
    }
}

struct Part {            // one array that is partitioned by state: `per_state` elements per state
    double *base;
    size_t per_state;
};

int nccl_fail(pcl_ctx *ctx, const char *what, ncclResult_t r) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, ncclGetErrorString(r));
    pcl_set_error(ctx, buf);
    return PCL_ERR_COMM;
}

// ---- host-callback transport: all-gather of every rank's bytes, reduced / sliced here, chunk by chunk
// visit(first element, count, all): `all` holds the ranks' copies of elements [first, first + count) back to back
template <typename T, typename F>
int host_gather_chunks(pcl_ctx *ctx, const T *dev, size_t n, F visit) {
    const size_t per = std::max<size_t>(host_chunk_bytes() / sizeof(T), 1);
    std::vector<T> mine, all;
    for (size_t first = 0; first < n; first += per) {
        const size_t cnt = std::min(per, n - first);
        mine.resize(cnt);
        all.resize(cnt * (size_t)ctx->nranks);
        HIPCHK(ctx, hipMemcpyAsync(mine.data(), dev + first, cnt * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->host_allgather(ctx->host_user, mine.data(), cnt * sizeof(T), all.data()) != 0)
            PCL_FAIL(ctx, PCL_ERR_COMM, "host rehearsal transport: the all-gather callback failed");
        TRY(visit(first, cnt, all.data()));
    }
    return PCL_OK;
}

template <typename T>
int host_allreduce(pcl_ctx *ctx, T *dev, size_t n, bool is_max) {
    std::vector<T> out;
    return host_gather_chunks<T>(ctx, dev, n, [&](size_t first, size_t cnt, const T *a) -> int {
        out.resize(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            T v = a[i];
            for (int r = 1; r < ctx->nranks; ++r) {
                const T w = a[(size_t)r * cnt + i];
                v = is_max ? (w > v ? w : v) : v + w;          // ranks in order: the same sum on every rank
            }
            out[i] = v;
        }
        HIPCHK(ctx, hipMemcpyAsync(dev + first, out.data(), cnt * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return PCL_OK;
    });
}

// sum all-reduce / max all-reduce of a device array, on whichever transport is up
template <typename T>
int allreduce(pcl_ctx *ctx, T *dev, size_t n, bool is_max) {
    if (ctx->transport == 2) return host_allreduce(ctx, dev, n, is_max);
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
    ncclResult_t r = ncclAllReduce(dev, dev, n, dt, is_max ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclAllReduce", r);
    return PCL_OK;
}

// state range of a rank: balanced, contiguous; equal sizes when nranks divides J (then native reduce-scatter / all-gather)
inline int range_lo(int J, int nranks, int r) { return (int)((long long)J * r / nranks); }

// In-place reduce-scatter by state range of every array of `parts`: afterwards rank r holds the global sums of ITS states
// (the rest of the array is stale partial data).
template <typename T>
int reduce_scatter_parts(pcl_ctx *ctx, T *const *bases, const size_t *per_state, int nparts, int J) {
    const int n = ctx->nranks, me = ctx->rank;
    if (ctx->transport == 2) {
        std::vector<T> out;
        for (int p = 0; p < nparts; ++p) {
            const size_t tot = per_state[p] * (size_t)J;
            const size_t lo = per_state[p] * (size_t)range_lo(J, n, me), hi = per_state[p] * (size_t)range_lo(J, n, me + 1);
            T *base = bases[p];
            TRY(host_gather_chunks<T>(ctx, base, tot, [&](size_t first, size_t cnt, const T *a) -> int {
                const size_t a0 = std::max(first, lo), a1 = std::min(first + cnt, hi);      // what this rank owns of the chunk
                if (a1 <= a0) return PCL_OK;
                out.resize(a1 - a0);
                for (size_t i = a0; i < a1; ++i) {
                    T v = a[i - first];
                    for (int r = 1; r < n; ++r) v += a[(size_t)r * cnt + (i - first)];
                    out[i - a0] = v;
                }
                HIPCHK(ctx, hipMemcpyAsync(base + a0, out.data(), (a1 - a0) * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                return PCL_OK;
            }));
        }
        return PCL_OK;
    }
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclGroupStart", r);
    for (int p = 0; p < nparts && r == ncclSuccess; ++p) {
        if (J % n == 0) {
            const size_t cnt = per_state[p] * (size_t)(J / n);
            r = ncclReduceScatter(bases[p], bases[p] + cnt * me, cnt, dt, ncclSum, comm, ctx->stream);
        } else {
            for (int root = 0; root < n && r == ncclSuccess; ++root) {      // uneven ranges: one rooted reduce per owner
                const size_t lo = per_state[p] * (size_t)range_lo(J, n, root), hi = per_state[p] * (size_t)range_lo(J, n, root + 1);
                if (hi > lo) r = ncclReduce(bases[p] + lo, bases[p] + lo, hi - lo, dt, ncclSum, root, comm, ctx->stream);
            }
        }
    }
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclReduceScatter", r);
    if (re != ncclSuccess) return nccl_fail(ctx, "ncclGroupEnd", re);
    return PCL_OK;
}

// In-place all-gather by state range: every rank contributes its own states and ends with all of them.
template <typename T>
int all_gather_parts(pcl_ctx *ctx, T *const *bases, const size_t *per_state, int nparts, int J) {
    const int n = ctx->nranks, me = ctx->rank;
    if (ctx->transport == 2) {
        std::vector<T> out;
        for (int p = 0; p < nparts; ++p) {
            const size_t tot = per_state[p] * (size_t)J;
            T *base = bases[p];
            TRY(host_gather_chunks<T>(ctx, base, tot, [&](size_t first, size_t cnt, const T *a) -> int {
                out.resize(cnt);
                for (int r = 0; r < n; ++r) {                      // every element from the rank that owns its state
                    const size_t lo = per_state[p] * (size_t)range_lo(J, n, r), hi = per_state[p] * (size_t)range_lo(J, n, r + 1);
                    const size_t a0 = std::max(first, lo), a1 = std::min(first + cnt, hi);
                    if (a1 > a0) memcpy(out.data() + (a0 - first), a + (size_t)r * cnt + (a0 - first), (a1 - a0) * sizeof(T));
                }
                HIPCHK(ctx, hipMemcpyAsync(base + first, out.data(), cnt * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                return PCL_OK;
            }));
        }
        (void)me;
        return PCL_OK;
    }
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclGroupStart", r);
    for (int p = 0; p < nparts && r == ncclSuccess; ++p) {
        if (J % n == 0) {
            const size_t cnt = per_state[p] * (size_t)(J / n);
            r = ncclAllGather(bases[p] + cnt * me, bases[p], cnt, dt, comm, ctx->stream);
        } else {
            for (int root = 0; root < n && r == ncclSuccess; ++root) {
                const size_t lo = per_state[p] * (size_t)range_lo(J, n, root), hi = per_state[p] * (size_t)range_lo(J, n, root + 1);
                if (hi > lo) r = ncclBroadcast(bases[p] + lo, bases[p] + lo, hi - lo, dt, root, comm, ctx->stream);
            }
        }
    }
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclAllGather", r);
    if (re != ncclSuccess) return nccl_fail(ctx, "ncclGroupEnd", re);
    return PCL_OK;
}

// log-sum-exp merge of the per-unit transition accumulators over the ranks (LHMM.py:272-290 does it over files)
int merge_hmm_acc(pcl_ctx *ctx) {
    if (!ctx->hmm_ksai || ctx->nranks == 1) return PCL_OK;
    const size_t n = (size_t)ctx->n_units * (ctx->S - 2) * (ctx->S + 1);
    double *top = nullptr;
    TRY(dev_alloc(ctx, &top, n));
    int rc = pcl_launch_hmm_acc_merge_prepare(ctx, top);
    if (rc == PCL_OK) rc = allreduce(ctx, top, n, true);
    if (rc == PCL_OK) rc = pcl_launch_hmm_acc_merge_scale(ctx, top);
    if (rc == PCL_OK) rc = allreduce(ctx, ctx->hmm_ksai, n, false);
    if (rc == PCL_OK) rc = pcl_launch_hmm_acc_merge_finish(ctx, top);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PCL_OK) rc = PCL_ERR_HIP;
    dev_free(top);
    return rc;
}

int ensure_payload32(pcl_ctx *ctx, size_t n) {
    if (ctx->payload32_len >= n) return PCL_OK;
    dev_free(ctx->payload32);                        // back to the pool it came from (a raw hipFree left a stale pool entry)
    ctx->payload32_len = 0;
    TRY(dev_alloc(ctx, &ctx->payload32, n));
    ctx->payload32_len = n;
    return PCL_OK;
}


// The statistics of the states [j0, j0 + len): in-place reduce-scatter (inside the range rank r owns the r-th slice, returned in
// *lo / *hi); payload PCL_F32 stages them as f32 (mean_acc relative to the old centre, see mean_to_f32_kernel) and puts the owned
// slice back as f64.
int exchange_reduce_scatter(pcl_ctx *ctx, int payload, int j0, int len, int *lo_out, int *hi_out) {
    const int n = ctx->nranks, me = ctx->rank, J = ctx->J;
    const size_t mp = (size_t)ctx->Mpad, mpd = mp * ctx->D;
    const int j_lo = j0 + range_lo(len, n, me), j_hi = j0 + range_lo(len, n, me + 1);
    *lo_out = j_lo;
    *hi_out = j_hi;
    const size_t per[4] = {mp, 1, mpd, mpd};
    int rc = PCL_OK;
    if (payload == PCL_F64) {
        double *bases[4] = {ctx->st_acc + per[0] * j0, ctx->st_alpha + per[1] * j0, ctx->st_mean + per[2] * j0, ctx->st_cov + per[3] * j0};
        return reduce_scatter_parts<double>(ctx, bases, per, 4, len);
    }
    TRY(ensure_payload32(ctx, ctx->stats_len));
    float *f = ctx->payload32;
    float *full[4] = {f, f + (size_t)J * mp, f + (size_t)J * mp + J, f + (size_t)J * mp + J + (size_t)J * mpd};
    double *src[4] = {ctx->st_acc, ctx->st_alpha, ctx->st_mean, ctx->st_cov};
    float *bases[4];
    for (int p = 0; p < 4; ++p) {
        bases[p] = full[p] + per[p] * j0;
        const size_t cnt = per[p] * (size_t)len;
        if (p == 2) hipLaunchKernelGGL(mean_to_f32_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->st_mean, ctx->st_acc, ctx->centers32, ctx->Mpad,
                                       ctx->D, 100.0, full[2], per[2] * (size_t)j0, cnt);
        else hipLaunchKernelGGL(to_f32_kernel, dim3(2048), dim3(256), 0, ctx->stream, src[p] + per[p] * j0, bases[p], cnt);
    }
    rc = reduce_scatter_parts<float>(ctx, bases, per, 4, len);
    for (int p = 0; p < 4 && rc == PCL_OK; ++p) {                  // the owned slices back to f64, where the M-step reads them
        const size_t lo = per[p] * (size_t)j_lo, cnt = per[p] * (size_t)(j_hi - j_lo);
        if (!cnt) continue;
        if (p == 2) hipLaunchKernelGGL(mean_from_f32_kernel, dim3(1024), dim3(256), 0, ctx->stream, full[2], ctx->st_acc, ctx->centers32,
                                       ctx->Mpad, ctx->D, 100.0, ctx->st_mean, lo, cnt);      // after p == 0 put the summed acc back
        else hipLaunchKernelGGL(to_f64_kernel, dim3(1024), dim3(256), 0, ctx->stream, full[p] + lo, src[p] + lo, cnt);
    }
    return rc;
}

// The new (mean, var, weight) of the states [j0, j0 + len): in-place all-gather from the ranks that own the slices.
int exchange_all_gather(pcl_ctx *ctx, int payload, int j0, int len) {
    const int n = ctx->nranks, me = ctx->rank, J = ctx->J;
    const size_t mp = (size_t)ctx->Mpad, mpd = mp * ctx->D;
    const int j_lo = j0 + range_lo(len, n, me), j_hi = j0 + range_lo(len, n, me + 1);
    const size_t per[3] = {mpd, mpd, mp};
    double *src[3] = {ctx->mean64, ctx->var64, ctx->w64};
    if (payload == PCL_F64) {
        double *bases[3] = {src[0] + per[0] * j0, src[1] + per[1] * j0, src[2] + per[2] * j0};
        return all_gather_parts<double>(ctx, bases, per, 3, len);
    }
    const size_t tot = (size_t)J * (2 * mpd + mp);
    TRY(ensure_payload32(ctx, std::max(tot, ctx->stats_len)));
    float *f = ctx->payload32;
    float *full[3] = {f, f + (size_t)J * mpd, f + 2 * (size_t)J * mpd};
    float *bases[3];
    for (int p = 0; p < 3; ++p) {
        bases[p] = full[p] + per[p] * j0;
        const size_t lo = per[p] * (size_t)j_lo, cnt = per[p] * (size_t)(j_hi - j_lo);
        if (cnt) hipLaunchKernelGGL(to_f32_kernel, dim3(1024), dim3(256), 0, ctx->stream, src[p] + lo, full[p] + lo, cnt);
    }
    int rc = all_gather_parts<float>(ctx, bases, per, 3, len);
    // every rank, the owner included, continues from the f32-rounded parameters: one model on all GPUs
    for (int p = 0; p < 3 && rc == PCL_OK; ++p)
        hipLaunchKernelGGL(to_f64_kernel, dim3(2048), dim3(256), 0, ctx->stream, bases[p], src[p] + per[p] * j0, per[p] * (size_t)len);
    return rc;
}

}  // namespace

void pcl_comm_release(pcl_ctx *ctx) {
    pcl_pipe_release(ctx);
    dev_free(ctx->payload32);
    ctx->payload32_len = 0;
}

extern "C" {

int pcl_device_count(int *n) {
    if (!n) return PCL_ERR_INVALID;
    int c = 0;
    const hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
    return PCL_OK;
}

int pcl_comm_unique_id(void *id_bytes128) {
    if (!id_bytes128) return PCL_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) PCL_FAIL(nullptr, PCL_ERR_COMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id_bytes128, &id, sizeof(id));
    return PCL_OK;
}

int pcl_comm_init(pcl_ctx *ctx, int rank, int nranks, const void *id_bytes128) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!id_bytes128 || nranks < 1 || rank < 0 || rank >= nranks) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_comm_init: rank %d of %d", rank, nranks);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_comm_destroy(ctx);
    ncclUniqueId id;
    memcpy(&id, id_bytes128, sizeof(id));
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) PCL_FAIL(ctx, PCL_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
    ctx->comm = (void *)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    ctx->transport = 1;
    return PCL_OK;
}

int pcl_comm_init_host(pcl_ctx *ctx, int rank, int nranks, pcl_allgather_fn fn, void *user) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!fn || nranks < 1 || rank < 0 || rank >= nranks) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_comm_init_host: rank %d of %d", rank, nranks);
    pcl_comm_destroy(ctx);
    ctx->host_allgather = fn;
    ctx->host_user = user;
    ctx->rank = rank;
    ctx->nranks = nranks;
    ctx->transport = 2;
    return PCL_OK;
}

int pcl_pipe_info(pcl_ctx *ctx, int *chunks, int *released_early) {
    if (!ctx) return PCL_ERR_INVALID;
    if (chunks) *chunks = ctx->pipe_K;
    if (released_early) *released_early = ctx->pipe_early;
    return PCL_OK;
}

int pcl_comm_info(pcl_ctx *ctx, int *rank, int *nranks, int *transport, int *rccl_nranks) {
    if (!ctx) return PCL_ERR_INVALID;
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
    if (transport) *transport = ctx->transport;
    if (rccl_nranks) {
        *rccl_nranks = 0;
        if (ctx->transport == 1) {
            int c = 0;
            ncclResult_t r = ncclCommCount((ncclComm_t)ctx->comm, &c);
            if (r != ncclSuccess) PCL_FAIL(ctx, PCL_ERR_COMM, "ncclCommCount: %s", ncclGetErrorString(r));
            *rccl_nranks = c;
        }
    }
    return PCL_OK;
}

int pcl_stats_allreduce(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_allreduce: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    if (ctx->transport == 0) {
        if (ctx->nranks == 1) return PCL_OK;   // single GPU: nothing to merge
        PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_allreduce: pcl_comm_init was not called");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_timer_begin(ctx, "allreduce");
    int rc = allreduce(ctx, ctx->stats, ctx->stats_len, false);
    pcl_timer_end(ctx, "allreduce");
    if (rc != PCL_OK) return rc;
    TRY(merge_hmm_acc(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_em_exchange(pcl_ctx *ctx, double c_covariance, int payload, int update_transitions) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_em_exchange: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    if (payload != PCL_F64 && payload != PCL_F32) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_em_exchange: payload %d", payload);
    if (ctx->transport == 0 && ctx->nranks != 1) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_em_exchange: pcl_comm_init was not called");
    if (update_transitions && !ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_em_exchange: update_transitions without pcl_units_upload");
    if (ctx->pipe_active) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_em_exchange: a pipelined exchange is in progress");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int J = ctx->J;
    const bool solo = ctx->transport == 0;                       // one GPU: the M-step alone
    int j_lo = 0, j_hi = J, rc = PCL_OK;
    if (!solo) {
        pcl_timer_begin(ctx, "reduce_scatter");
        rc = exchange_reduce_scatter(ctx, payload, 0, J, &j_lo, &j_hi);
        pcl_timer_end(ctx, "reduce_scatter");
        if (rc != PCL_OK) return rc;
    }
    // GMM.update_param (Clustering.py:682-693) for the owned states
    pcl_timer_begin(ctx, "mstep_owned");
    rc = pcl_launch_mstep_range(ctx, c_covariance, j_lo, j_hi);
    pcl_timer_end(ctx, "mstep_owned");
    if (rc != PCL_OK) return rc;
    if (!solo) {
        pcl_timer_begin(ctx, "all_gather");
        rc = exchange_all_gather(ctx, payload, 0, J);
        pcl_timer_end(ctx, "all_gather");
        if (rc != PCL_OK) return rc;
        TRY(merge_hmm_acc(ctx));
    }
    if (update_transitions) TRY(pcl_launch_trans_mstep(ctx));     // tiny, identical on every rank after the merge
    pcl_timer_begin(ctx, "derive");
    rc = pcl_launch_derive(ctx);                                   // every scoring layout, from the gathered master copy
    pcl_timer_end(ctx, "derive");
    if (rc != PCL_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- the exchange behind the accumulate pass, chunk by chunk
// The accumulate pass walks the states in ascending order, a group at a time (gmm_accumulate.hip).  With a pipelined exchange
// open, the states are cut into K equal chunks, and as soon as the pass has queued the last kernel that touches a chunk, the
// chunk goes -- on its own stream, behind an event -- through reduce-scatter (inside the chunk rank r owns the r-th slice)
// -> GMM.update_param on the owned slice -> all-gather -> the chunk's layouts re-derived, while the matrix pipes work on the
// later groups.  One rank: M-step + derive of finished chunks beside the rest of the pass.  Same sums, same M-step arithmetic
// as pcl_em_exchange; only WHO re-estimates a state differs (the r-th slice of every chunk instead of one range).
namespace {
struct StreamSwap {                                                // the launchers and their timers use ctx->stream
    pcl_ctx *c;
    hipStream_t keep;
    StreamSwap(pcl_ctx *ctx, hipStream_t s) : c(ctx), keep(ctx->stream) { ctx->stream = s; }
    ~StreamSwap() { c->stream = keep; }
};

// mode 1 (default): only the chunk's REDUCE-SCATTER leaves early -- it costs the GPU little beside the accumulate pass, and it
// is half of the bytes on the wire; M-step, all-gather and derive follow in pcl_pipe_finish.  mode 0 (PCL_PIPE_MODE=0): the
// chunk's whole chain leaves early (measured on one GPU: the M-step and derive kernels beside the power-limited accumulate pass
// stretch it by more than they take alone).
int pipe_chunk_tail(pcl_ctx *ctx, int c, int lo, int hi) {          // M-step of the owned slice -> all-gather -> (mode 0: derive)
    const int J = ctx->J, K = ctx->pipe_K, a = range_lo(J, K, c), b = range_lo(J, K, c + 1);
    const bool solo = ctx->transport == 0;
    pcl_timer_begin(ctx, "mstep_owned");
    int rc = pcl_launch_mstep_range(ctx, ctx->pipe_c_cov, lo, hi);
    pcl_timer_end(ctx, "mstep_owned");
    if (rc != PCL_OK) return rc;
    if (!solo) {
        pcl_timer_begin(ctx, "all_gather");
        rc = exchange_all_gather(ctx, ctx->pipe_payload, a, b - a);
        pcl_timer_end(ctx, "all_gather");
        if (rc != PCL_OK) return rc;
    }
    if (ctx->pipe_mode == 0) {
        pcl_timer_begin(ctx, "derive");
        rc = pcl_launch_derive_range(ctx, a, b);
        pcl_timer_end(ctx, "derive");
    }
    return rc;
}

// INVARIANT the early release rests on: every kernel of the accumulate pass (count / scan / fill, producer, consumer, the masked and
// the direct-form fix-up) and the M-step / derive kernels index the statistics, the master copy and every layout strictly PER STATE --
// no kernel reads a state's mean64 / centers32 / fscale / kzero / pm16f while working on another state.  A chunk [a, b) may therefore be
// re-estimated (mode 0: and re-derived) on stream_comm while the main and auxiliary streams still accumulate states >= b.  Whoever
// adds a kernel that looks across states (a global normaliser, a shared codebook) must close the pipe first.  pcl_pipe_info reports
// how many chunks did leave early: 0 means the pass could not release any (states out of ascending order, or states on the direct-form
// kernel, which are accumulated last) and the call was the plain accumulate + exchange.
int pipe_issue_chunk(pcl_ctx *ctx, int c) {
    const int J = ctx->J, K = ctx->pipe_K, a = range_lo(J, K, c), b = range_lo(J, K, c + 1);
    if (b <= a) return PCL_OK;
    const bool solo = ctx->transport == 0;
    if (solo && ctx->pipe_mode != 0) return PCL_OK;                // one GPU, nothing to send: everything waits for pcl_pipe_finish
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_comm, ctx->pipe_ev[c], 0));
    StreamSwap sw(ctx, ctx->stream_comm);
    int lo = a, hi = b, rc = PCL_OK;
    if (!solo) {
        pcl_timer_begin(ctx, "reduce_scatter");
        rc = exchange_reduce_scatter(ctx, ctx->pipe_payload, a, b - a, &lo, &hi);
        pcl_timer_end(ctx, "reduce_scatter");
        if (rc != PCL_OK) return rc;
    }
    if (ctx->pipe_mode != 0) return PCL_OK;
    return pipe_chunk_tail(ctx, c, lo, hi);
}
}  // namespace

int pcl_pipe_begin(pcl_ctx *ctx, double c_covariance, int payload, int n_chunks) {
    if (ctx->pipe_active) PCL_FAIL(ctx, PCL_ERR_STATE, "pipelined exchange: already open");
    if (!ctx->stream_comm) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream_comm, hipStreamNonBlocking));
    const int K = std::max(1, std::min(n_chunks, std::min(ctx->J, 64)));
    while ((int)ctx->pipe_ev.size() < K) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_ev.push_back(e);
    }
    if (payload == PCL_F32 && ctx->transport != 0)
        TRY(ensure_payload32(ctx, std::max(ctx->stats_len, (size_t)ctx->J * (2 * (size_t)ctx->Mpad * ctx->D + ctx->Mpad))));
    ctx->pipe_K = K;
    ctx->pipe_mode = getenv("PCL_PIPE_MODE") ? atoi(getenv("PCL_PIPE_MODE")) : 1;
    ctx->pipe_next = 0;
    ctx->pipe_early = 0;
    ctx->pipe_payload = payload;
    ctx->pipe_c_cov = c_covariance;
    ctx->pipe_active = true;
    return PCL_OK;
}

// every state below `final_below` has its final statistics once what is queued on ctx->stream by now has run
int pcl_pipe_progress(pcl_ctx *ctx, int final_below) {
    if (!ctx->pipe_active) return PCL_OK;
    while (ctx->pipe_next < ctx->pipe_K && range_lo(ctx->J, ctx->pipe_K, ctx->pipe_next + 1) <= final_below) {
        const int c = ctx->pipe_next++;
        if (final_below < ctx->J) ++ctx->pipe_early;              // released by the accumulate pass itself, not by pcl_pipe_finish
        HIPCHK(ctx, hipEventRecord(ctx->pipe_ev[c], ctx->stream));
        TRY(pipe_issue_chunk(ctx, c));
    }
    return PCL_OK;
}

int pcl_pipe_finish(pcl_ctx *ctx, int update_transitions) {
    if (!ctx->pipe_active) PCL_FAIL(ctx, PCL_ERR_STATE, "pipelined exchange: not open");
    int rc = pcl_pipe_progress(ctx, ctx->J);                       // whatever the pass did not release itself
    ctx->pipe_active = false;
    if (rc != PCL_OK) return rc;
    if (!ctx->pipe_done) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->pipe_done, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->pipe_done, ctx->stream_comm));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_done, 0));
    if (ctx->pipe_mode != 0) {                                     // the chunks' M-steps and all-gathers, then every layout at once
        const int J = ctx->J, K = ctx->pipe_K, n = ctx->nranks, me = ctx->rank;
        for (int c = 0; c < K; ++c) {
            const int a = range_lo(J, K, c), b = range_lo(J, K, c + 1);
            if (b <= a) continue;
            const bool solo = ctx->transport == 0;
            TRY(pipe_chunk_tail(ctx, c, solo ? a : a + range_lo(b - a, n, me), solo ? b : a + range_lo(b - a, n, me + 1)));
        }
    }
    if (ctx->transport != 0) TRY(merge_hmm_acc(ctx));
    if (update_transitions) TRY(pcl_launch_trans_mstep(ctx));
    if (ctx->pipe_mode != 0) {
        pcl_timer_begin(ctx, "derive");
        const int rd = pcl_launch_derive(ctx);                     // (waits, bumps the model generation)
        pcl_timer_end(ctx, "derive");
        if (rd != PCL_OK) return rd;
    } else {
        TRY(pcl_derive_finish(ctx));
    }
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

void pcl_pipe_release(pcl_ctx *ctx) {
    for (hipEvent_t e : ctx->pipe_ev) hipEventDestroy(e);
    ctx->pipe_ev.clear();
    if (ctx->pipe_done) hipEventDestroy(ctx->pipe_done);
    ctx->pipe_done = nullptr;
    if (ctx->stream_comm) {
        hipStreamSynchronize(ctx->stream_comm);
        hipStreamDestroy(ctx->stream_comm);
    }
    ctx->stream_comm = nullptr;
    ctx->pipe_active = false;
}

extern "C" {

int pcl_comm_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (ctx->comm) {
        ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->host_allgather = nullptr;
    ctx->host_user = nullptr;
    ctx->transport = 0;
    ctx->rank = 0;
    ctx->nranks = 1;
    pcl_comm_release(ctx);
    return PCL_OK;
}

}  // extern "C"

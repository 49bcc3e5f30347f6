// pcl_comm.hip -- the one collective of the path: sum of the E-step statistics over the GPUs of a
// node with RCCL over xGMI.  Replaces the reference's file-based accumulator merge
// (StatisticalModel/LHMM.py:256-290, StatisticalModel/Clustering.py:314-367): there every worker
// np.save()s log-domain accumulators and a reducer log-sum-exps the files; here the statistics are
// linear-domain float64 sums resident in HBM, so the merge is a single ncclSum all-reduce.
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include "pcl_internal.h"

extern "C" {

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
    return PCL_OK;
}

int pcl_stats_allreduce(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_allreduce: no model uploaded");
    if (ctx->nranks == 1 && !ctx->comm) return PCL_OK;   // single GPU: nothing to merge
    if (!ctx->comm) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_allreduce: pcl_comm_init was not called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_timer_begin(ctx, "allreduce");
    ncclResult_t r = ncclAllReduce(ctx->stats, ctx->stats, ctx->stats_len, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    pcl_timer_end(ctx, "allreduce");
    if (r != ncclSuccess) PCL_FAIL(ctx, PCL_ERR_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_comm_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (ctx->comm) {
        ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->rank = 0;
    ctx->nranks = 1;
    return PCL_OK;
}

}  // extern "C"

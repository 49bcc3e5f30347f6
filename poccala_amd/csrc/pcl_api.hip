// pcl_api.hip -- host side of the C-ABI declared in include/poccala_hip.h.
// Owns device memory, builds the state-major scoring work lists and the sparse transition
// structure, and launches the kernels in gmm_score.hip / hmm_dp.hip / gmm_accumulate.hip.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <map>
#include <mutex>
#include <unordered_map>

#include "pcl_internal.h"

static std::string g_init_error;

void pcl_set_error(pcl_ctx *ctx, const char *msg) {
    if (ctx) ctx->err = msg;
    else g_init_error = msg;
}

// ---------------------------------------------------------------- timers (HIP events on ctx->stream)
// Opt-in (pcl_timing_enable / env PCL_TIMERS=1): a training run that never asks for kernel times must not pay two
// hipEventCreate per launch nor keep the events alive until the context dies.
void pcl_timer_begin(pcl_ctx *ctx, const char *which) {
    if (!ctx->timing) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    hipEventRecord(a, ctx->stream);
    ctx->timers[which].ev.push_back({a, b});
}
void pcl_timer_end(pcl_ctx *ctx, const char *which) {
    if (!ctx->timing) return;
    auto &t = ctx->timers[which];
    if (!t.ev.empty()) hipEventRecord(t.ev.back().second, ctx->stream);
}

static void drop_timers(pcl_ctx *ctx) {
    for (auto &kv : ctx->timers) {
        for (auto &p : kv.second.ev) {
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        kv.second.ev.clear();
    }
}

// ================================================================ device memory pool
// Size classes: powers of two up to 1 MiB, eighths of a power of two above (a block wastes < 12.5 %), so that batches of
// similar shape -- the chunks of a ragged stream, the one-utterance batches of the drop-in classes -- reuse each other's
// blocks.  The cache is trimmed (largest blocks first) when it exceeds PCL_POOL_MAX_MB (default 65536), and released
// altogether before an allocation is reported as failed.
thread_local int pcl_tls_free_synced = 0;
namespace {
struct PoolBlock { size_t bytes; int device; };
std::mutex g_pool_mu;
std::unordered_map<void *, PoolBlock> g_pool_live;                 // every block handed out or cached
std::multimap<std::pair<int, size_t>, void *> g_pool_free;           // (device, class bytes) -> cached block
size_t g_pool_cached = 0;
size_t pool_class(size_t n) {
    if (n <= 256) return 256;
    size_t p2 = 256;
    while (p2 < n) p2 <<= 1;
    if (p2 <= (1u << 20)) return p2;
    const size_t step = p2 >> 4;                                    // eighths of the lower power of two
    return (n + step - 1) / step * step;
}
size_t pool_limit() {
    static const size_t lim = (size_t)(getenv("PCL_POOL_MAX_MB") ? atol(getenv("PCL_POOL_MAX_MB")) : 65536) << 20;
    return lim;
}
void pool_release_locked(size_t keep) {                              // hipFree cached blocks, largest first, down to `keep` bytes
    while (g_pool_cached > keep && !g_pool_free.empty()) {
        auto best = g_pool_free.begin();
        for (auto it = g_pool_free.begin(); it != g_pool_free.end(); ++it)
            if (it->first.second > best->first.second) best = it;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != best->first.first) (void)hipSetDevice(best->first.first);
        (void)hipFree(best->second);
        if (cur != best->first.first) (void)hipSetDevice(cur);
        g_pool_cached -= best->first.second;
        g_pool_live.erase(best->second);
        g_pool_free.erase(best);
    }
}
}  // namespace

void *pcl_pool_alloc(int device, size_t bytes) {
    const size_t cls = pool_class(bytes);
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto it = g_pool_free.find(std::make_pair(device, cls));
    if (it != g_pool_free.end()) {
        void *p = it->second;
        g_pool_free.erase(it);
        g_pool_cached -= cls;
        return p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, cls) != hipSuccess) {
        (void)hipGetLastError();
        pool_release_locked(0);                                      // give the cache back and try once more
        if (hipMalloc(&p, cls) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    g_pool_live[p] = PoolBlock{cls, device};                        // (overwrites a stale entry, should a block ever have left the pool by a raw hipFree)
    return p;
}

void pcl_pool_free(void *p) {
    if (!p) return;
    if (pcl_tls_free_synced <= 0) (void)hipDeviceSynchronize();     // what hipFree did: nobody on the device still uses the block
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) {                                   // not ours (never happens): the runtime's problem
        (void)hipFree(p);
        return;
    }
    g_pool_free.emplace(std::make_pair(it->second.device, it->second.bytes), p);
    g_pool_cached += it->second.bytes;
    if (g_pool_cached > pool_limit()) pool_release_locked(pool_limit() / 2);
}

// ---------------------------------------------------------------- descriptor uploads (pcl_desc_group, pcl_internal.h)
namespace {
__global__ void desc_copy_kernel(DescCopyArgs a, const char *__restrict__ stage) {
    const int e = blockIdx.y;
    if (e >= a.n) return;
    const unsigned long long n = a.bytes[e];                          // (0: the entry was superseded by a later upload of the same array)
    if (n == 0) return;
    const char *src = stage + a.off[e];
    char *dst = (char *)a.dst[e];
    const unsigned long long n16 = n / 16;                            // (staging offsets and pool blocks are 256-byte aligned)
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x)
        ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
    if (blockIdx.x == 0)
        for (unsigned long long i = n16 * 16 + threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
}  // namespace

hipError_t pcl_desc_flush(pcl_ctx *ctx) {
    ctx->desc_pin_used = 0;
    if (ctx->desc_n == 0) return hipSuccess;
    static_assert(PCL_DESC_MAX == 24, "pcl_ctx::desc_dst");
    DescCopyArgs a;
    a.n = ctx->desc_n;
    for (int k = 0; k < ctx->desc_n; ++k) {
        a.dst[k] = ctx->desc_dst[k];
        a.off[k] = ctx->desc_off[k];
        a.bytes[k] = ctx->desc_bytes[k];
    }
    ctx->desc_n = 0;
    void *dev_stage = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev_stage, ctx->desc_pin, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(desc_copy_kernel, dim3(32, a.n), dim3(256), 0, ctx->stream_desc, a, (const char *)dev_stage);
    e = hipGetLastError();
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream_desc);
}

static int pcl_batch_reap(pcl_ctx *ctx, bool wait);   // frees the destroyed batches the GPU is done with (wait: all of them); returns how many are left

extern "C" {

// ================================================================ context
int pcl_init(int device, pcl_ctx **out) {
    if (!out) PCL_FAIL(nullptr, PCL_ERR_INVALID, "pcl_init: out is NULL");
    *out = nullptr;
    // five streams per context on the runtime's default of four hardware queues would make two of them share one (see poccala_amd/_lib.py);
    // honoured only if the HIP runtime has not started yet -- a C caller that initialises HIP first exports GPU_MAX_HW_QUEUES=8 itself
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        PCL_FAIL(nullptr, PCL_ERR_HIP, "pcl_init: no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) PCL_FAIL(nullptr, PCL_ERR_INVALID, "pcl_init: device %d out of range [0,%d)", device, n);
    pcl_ctx *ctx = new pcl_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_init_error = std::string("pcl_init: ") + hipGetErrorString(e);
        delete ctx;
        return PCL_ERR_HIP;
    }
    {
        hipDeviceProp_t prop0;
        if (hipGetDeviceProperties(&prop0, device) == hipSuccess) ctx->cus = prop0.multiProcessorCount;
    }
    // (compute units set aside for the second stream with hipExtStreamCreateWithCUMask were tried in round 4: 8 of 256 cost the scoring
    //  kernel 20 %, 16 cost 70 %, and the posterior kernel's in-loop span did not move -- it waits for registers, not for CUs)
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->stream_dp, hipStreamDefault, -1)) != hipSuccess) {      // (priority 0 measured: no difference)
        g_init_error = std::string("pcl_init: ") + hipGetErrorString(e);
        delete ctx;
        return PCL_ERR_HIP;
    }
    if ((e = hipStreamCreate(&ctx->stream_aux)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream_desc, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream_d2h, hipStreamNonBlocking)) != hipSuccess) {
        g_init_error = std::string("pcl_init: ") + hipGetErrorString(e);
        delete ctx;
        return PCL_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cus = prop.multiProcessorCount;
    // PCL_SCORE_VARIANT: 7 (default) = f32-class contraction on the f16 matrix pipe, 3 = strict f32 on the f32-input MFMA,
    // 1 = direct form on the VALU (the kernels states leave the matrix pipe for: fix-up, ill-conditioned models)
    const char *var = getenv("PCL_SCORE_VARIANT");
    ctx->score_variant = var ? atoi(var) : 7;
    if (ctx->score_variant != 1 && ctx->score_variant != 3 && ctx->score_variant != 7) {
        g_init_error = "pcl_init: PCL_SCORE_VARIANT must be 1, 3 or 7";
        hipStreamDestroy(ctx->stream); hipStreamDestroy(ctx->stream_dp); hipStreamDestroy(ctx->stream_aux); hipStreamDestroy(ctx->stream_desc);
        delete ctx;
        return PCL_ERR_INVALID;
    }
    if (const char *cm = getenv("PCL_MFMA_COND_MAX")) ctx->cond_max = (float)atof(cm);
    if (const char *sm = getenv("PCL_SPLIT_MAX")) {
        ctx->split_frac = std::min(1.0f, std::max(0.0f, (float)atof(sm)));
        ctx->split_frac_set = true;
    }
    if (const char *ds = getenv("PCL_DP_STREAM")) ctx->dp_async = atoi(ds) != 0;
    if (const char *co = getenv("PCL_COARSE")) ctx->coarse_on = atoi(co) != 0;
    if (const char *cm2 = getenv("PCL_COMPACT_MAIN")) ctx->compact_main = atoi(cm2) != 0;
    if (const char *cs = getenv("PCL_COARSE_STATS")) ctx->coarse_stats = atoi(cs) != 0;
    if (const char *cp = getenv("PCL_COARSE_PASSES")) ctx->coarse_np = atoi(cp) == 3 ? 3 : 1;
    if (const char *cm_ = getenv("PCL_COARSE_SPLIT_MAX")) ctx->coarse_split_frac = std::min(1.0f, std::max(0.0f, (float)atof(cm_)));   // (scoring only: A/B, tests)
    if (const char *tm = getenv("PCL_TIMERS")) ctx->timing = atoi(tm) != 0;
    *out = ctx;
    return PCL_OK;
}

static void free_model(pcl_ctx *ctx) {
    pcl_accumulate_release(ctx);                                 // (sized for the model's states and mixtures)
    pcl_coarse_release(ctx);
    dev_free(ctx->params32);
    dev_free(ctx->params64);
    dev_free(ctx->mean32);
    dev_free(ctx->mean64);
    dev_free(ctx->var64);
    dev_free(ctx->w64);
    dev_free(ctx->pm32);
    dev_free(ctx->pm16f);
    dev_free(ctx->kzero);
    dev_free(ctx->fscale);
    dev_free(ctx->centers32);
    dev_free(ctx->d_cond);
    dev_free(ctx->d_bad);
    dev_free(ctx->d_bad_idx);
    dev_free(ctx->d_nbad);
    dev_free(ctx->d_non);
    dev_free(ctx->d_good_idx);
    dev_free(ctx->d_npt);
    ctx->nbad.clear();
    if (ctx->zero_pending && ctx->ev_zero) (void)hipEventSynchronize(ctx->ev_zero);
    ctx->zero_pending = false;
    dev_free(ctx->stats);
    ctx->st_acc = ctx->st_alpha = ctx->st_mean = ctx->st_cov = nullptr;
    ctx->J = ctx->M = ctx->Mpad = 0;
}

// frames32 is either a plain upload (owned) or a view of a streaming slot
static void release_frames32(pcl_ctx *ctx) {
    if (ctx->frames_front >= 0) {
        ctx->frames32 = nullptr;
        ctx->frames_front = -1;
    } else {
        dev_free(ctx->frames32);
    }
}

int pcl_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_OK;
    hipSetDevice(ctx->device);
    // Drain first, release second: every stream of the context (twice -- a stream drained early may have been handed work by an event
    // of one drained later only in the sense that its wait completes; nothing new is queued, so the second round returns at once and
    // is there for the reader), THEN the communicator (collectives and the pipelined exchange run on these streams and stream_comm;
    // rounds 1-5 destroyed it first), the buried batches, events, memory, streams.
    for (int round = 0; round < 2; ++round) {
        hipStream_t all[] = {ctx->stream, ctx->stream_dp, ctx->stream_aux, ctx->stream_d2h, ctx->stream_desc, ctx->stream_comm};
        for (hipStream_t s : all)
            if (s) hipStreamSynchronize(s);
    }
    pcl_comm_destroy(ctx);
    pcl_batch_reap(ctx, true);
    // (round 6, tools/lifecycle_stress.py: a context destroyed with an asynchronous pcl_stats_zero still un-joined -- zero_pending --
    //  had ev_zero destroyed HERE and then synchronised on by free_model below: a use of a dead event, SIGSEGV inside the runtime.
    //  The streams are drained: nothing is pending any more, and the handles are cleared as they go.)
    ctx->zero_pending = false;
    if (ctx->ev_zero) hipEventDestroy(ctx->ev_zero);
    if (ctx->ev_zero_src) hipEventDestroy(ctx->ev_zero_src);
    ctx->ev_zero = ctx->ev_zero_src = nullptr;
    if (ctx->desc_pin) hipHostFree(ctx->desc_pin);
    ctx->desc_pin = nullptr;
    drop_timers(ctx);
    free_model(ctx);
    pcl_units_release(ctx);
    release_frames32(ctx);
    dev_free(ctx->frames64);
    dev_free(ctx->d_softplus);
    dev_free(ctx->frames_slot[0]);
    dev_free(ctx->frames_slot[1]);
    if (ctx->ev_stage) hipEventDestroy(ctx->ev_stage);
    if (ctx->ev_slot_free) hipEventDestroy(ctx->ev_slot_free);
    hipStreamDestroy(ctx->stream);
    hipStreamDestroy(ctx->stream_dp);
    hipStreamDestroy(ctx->stream_aux);
    if (ctx->stream_desc) hipStreamDestroy(ctx->stream_desc);
    if (ctx->stream_d2h) hipStreamDestroy(ctx->stream_d2h);
    delete ctx;
    return PCL_OK;
}

const char *pcl_last_error(pcl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int pcl_sync(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_dp));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_aux));
    if (ctx->stream_d2h) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_d2h));
    pcl_batch_reap(ctx, false);
    return PCL_OK;
}

// The dynamic-programming kernels of a batch run on a second, higher-priority stream (one wave per utterance, no
// matrix work: they fit beside another batch's scoring).  Anything else that touches the batch on the main stream
// first orders the main stream behind them.
static int batch_join(pcl_batch *b) {
    if (b->dp_pending) {
        HIPCHK(b->ctx, hipStreamWaitEvent(b->ctx->stream, b->ev_dp, 0));
        b->dp_pending = false;
    }
    if (b->fetch_pending) {                                  // result copies still reading this batch's buffers (pcl_batch_fetch_async)
        HIPCHK(b->ctx, hipStreamWaitEvent(b->ctx->stream, b->ev_fetch, 0));
        b->fetch_pending = false;
    }
    return PCL_OK;
}

int pcl_device_info(pcl_ctx *ctx, char *name, int cap, int *cus, size_t *hbm_bytes) {
    if (!ctx) return PCL_ERR_INVALID;
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && cap > 0) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return PCL_OK;
}

int pcl_timing_enable(pcl_ctx *ctx, int on) {
    if (!ctx) return PCL_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_dp));
    if (!on) drop_timers(ctx);
    ctx->timing = on != 0;
    return PCL_OK;
}

int pcl_kernel_time(pcl_ctx *ctx, const char *which, float *total_ms, int *launches) {
    if (!ctx || !which) return PCL_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_dp));
    float tot = 0.f;
    int n = 0;
    auto it = ctx->timers.find(which);
    if (it != ctx->timers.end()) {
        for (auto &p : it->second.ev) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                tot += ms;
                ++n;
            }
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        it->second.ev.clear();
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return PCL_OK;
}

// ================================================================ model
// Device feature dimension.  Up to 47 it is one of 13 / 26 / 39 / 47 -- the sizes with a spare K slot in their last k-step of
// 8, which the matrix-pipe kernels need for the folded constants -- so that EVERY model of up to 47 features is scored and
// accumulated on the matrix pipe (the padding features are zero in the frames and carry zero coefficients; round 2 padded
// e.g. D = 20 to 24 and silently dropped it to the 4-5x slower VALU kernels).  Above: 48 / 64 on the VALU kernels.
static int device_dim(int D) {
    const int pipe[] = {13, 26, 39, 47};
    for (int o : pipe)
        if (D <= o) return o;
    const int opts[] = {48, 64};
    for (int o : opts)
        if (D <= o) return o;
    return -1;
}

int pcl_model_upload(pcl_ctx *ctx, int J, int M, int D, const double *mean, const double *var, const double *weight,
                     int flags) {
    if (!ctx) return PCL_ERR_INVALID;
    if (J <= 0 || M <= 0 || D <= 0 || !mean || !var || !weight) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_model_upload: bad shape J=%d M=%d D=%d", J, M, D);
    const int Dd = device_dim(D);
    if (Dd < 0) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_model_upload: feature dimension %d > 64 is not supported", D);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    free_model(ctx);
    const int Mpad = (M + 3) / 4 * 4;
    const int row = (2 * Dd + 1 + 3) / 4 * 4;
    const int Mp32 = (M + 31) / 32 * 32, KS4 = (Dd + 1 + 3) / 4;
    const size_t np = (size_t)J * Mpad * row, nm = (size_t)J * Mpad * Dd, nw = (size_t)J * Mpad;
    const size_t npm = (size_t)J * (Mp32 / 32) * KS4 * 64 * 4;
    // float64 master copy in the padded device layout; every derived layout is built on the device
    std::vector<double> m64(nm, 0.0), v64(nm, 1.0), w64(nw, 0.0);
    for (int j = 0; j < J; ++j)
        for (int m = 0; m < M; ++m) {
            w64[(size_t)j * Mpad + m] = weight[(size_t)j * M + m];
            for (int d = 0; d < D; ++d) {
                const double vr = var[((size_t)j * M + m) * D + d];
                if (!(vr > 0.0)) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_model_upload: variance[%d,%d,%d] = %g is not positive", j, m, d, vr);
                m64[((size_t)j * Mpad + m) * Dd + d] = mean[((size_t)j * M + m) * D + d];
                v64[((size_t)j * Mpad + m) * Dd + d] = vr;
            }
        }
    (void)np;                                                     // (params32 / params64 / mean32: allocated when first derived, model_derive.hip)
    TRY(dev_alloc(ctx, &ctx->mean64, nm));
    TRY(dev_alloc(ctx, &ctx->var64, nm));
    TRY(dev_alloc(ctx, &ctx->w64, nw));
    TRY(dev_alloc(ctx, &ctx->pm32, npm));
    TRY(dev_alloc(ctx, &ctx->pm16f, (size_t)J * (Mp32 / 32) * 2 * ((Dd + 7) / 8) * 64 * 8));
    TRY(dev_alloc(ctx, &ctx->kzero, (size_t)J));
    TRY(dev_alloc(ctx, &ctx->fscale, (size_t)J * 2 * ((Dd + 7) / 8) * 8));
    TRY(dev_alloc(ctx, &ctx->centers32, (size_t)J * Dd));
    TRY(dev_alloc(ctx, &ctx->d_cond, (size_t)J));
    TRY(dev_alloc(ctx, &ctx->d_bad, (size_t)J * Mpad));
    TRY(dev_alloc(ctx, &ctx->d_bad_idx, (size_t)J * Mpad));
    TRY(dev_alloc(ctx, &ctx->d_nbad, (size_t)J));
    TRY(dev_alloc(ctx, &ctx->d_non, (size_t)J));
    TRY(dev_alloc(ctx, &ctx->d_good_idx, (size_t)J * Mpad));
    TRY(dev_alloc(ctx, &ctx->d_npt, (size_t)J));
    // with the coarse pass (gmm_score_coarse.hip) a state's off-pipe mixtures cost the scoring about what they would cost on the pipe, so
    // states stay split for scoring up to coarse_split_frac (0.99) of their mixtures.  Beyond it the route stops paying: tens of pairs per
    // frame and state pass the bound (many of a state's broader mixtures have by then crossed cond_max themselves and sit, off-pipe,
    // within reach of most frames) for a pass that evaluates a pair per lane, and the whole-state direct form with its
    // partial-distance test is the cheaper route.  Config 4's EM iterations 5 / 6 / 7 (98.8 / 99.7 / 99.8 % off-pipe) at
    // the limits 0.95, 0.99, 0.998: 274 / 256 / 275, 206 / 260 / 306 and 160 / 276 / 362 ms (profiles/r06_coarse_rework.txt).  PCL_SPLIT_MAX overrides both limits; the accumulate pass keeps the round 4-5 half
    ctx->acc_split_max = (int)((ctx->split_frac_set ? ctx->split_frac : 0.5f) * (float)M);
    ctx->split_max = (int)(((ctx->split_frac_set || !pcl_coarse_enabled_for(ctx, Dd)) ? ctx->split_frac : ctx->coarse_split_frac) * (float)M);
    HIPCHK(ctx, hipMemcpy(ctx->mean64, m64.data(), nm * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->var64, v64.data(), nm * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->w64, w64.data(), nw * sizeof(double), hipMemcpyHostToDevice));
    ctx->J = J;
    ctx->M = M;
    ctx->Mpad = Mpad;
    ctx->Mpad32 = Mp32;
    ctx->D = Dd;
    ctx->Dhost = D;
    ctx->row = row;
    ctx->model_flags = flags;
    TRY(pcl_launch_derive(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // statistics: [acc J*Mpad | alpha J | mean J*Mpad*Dd | cov J*Mpad*Dd]
    ctx->stats_len = (size_t)J * Mpad + J + 2 * nm;
    TRY(dev_alloc(ctx, &ctx->stats, ctx->stats_len));
    ctx->st_acc = ctx->stats;
    ctx->st_alpha = ctx->st_acc + (size_t)J * Mpad;
    ctx->st_mean = ctx->st_alpha + J;
    ctx->st_cov = ctx->st_mean + nm;
    HIPCHK(ctx, hipMemset(ctx->stats, 0, ctx->stats_len * sizeof(double)));
    return PCL_OK;
}

// ================================================================ frames
int pcl_frames_upload(pcl_ctx *ctx, int64_t F, int D, const void *frames, int dtype) {
    if (!ctx) return PCL_ERR_INVALID;
    if (F <= 0 || D <= 0 || !frames) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_frames_upload: bad shape F=%lld D=%d", (long long)F, D);
    const int Dd = device_dim(D);
    if (Dd < 0) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_frames_upload: feature dimension %d > 64 is not supported", D);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    release_frames32(ctx);
    dev_free(ctx->frames64);
    const size_t n = (size_t)F * Dd;
    // Direct PCIe copy in the host element type (padded on the host only when D has no exact kernel);
    // the other precision is derived on the device: f32 now (every mode reads it), f64 lazily (parity mode).
    const size_t esz = (dtype == PCL_F64) ? sizeof(double) : sizeof(float);
    const void *src = frames;
    std::vector<char> padded;
    if (Dd != D) {
        padded.assign(n * esz, 0);
        for (int64_t f = 0; f < F; ++f) memcpy(&padded[(size_t)f * Dd * esz], (const char *)frames + (size_t)f * D * esz, (size_t)D * esz);
        src = padded.data();
    }
    TRY(dev_alloc(ctx, &ctx->frames32, n));
    if (dtype == PCL_F64) {
        TRY(dev_alloc(ctx, &ctx->frames64, n));
        HIPCHK(ctx, hipMemcpy(ctx->frames64, src, n * esz, hipMemcpyHostToDevice));
        TRY(pcl_launch_cast(ctx, ctx->frames64, ctx->frames32, nullptr, n));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        HIPCHK(ctx, hipMemcpy(ctx->frames32, src, n * esz, hipMemcpyHostToDevice));
    }
    ctx->F = F;
    ctx->FD = Dd;
    ctx->FDhost = D;
    return PCL_OK;
}

// Streaming (BASELINE config 5: the corpus does not fit a batch): the next chunk's frames travel on the copy stream
// into the slot that is not being scored, so the H2D leg runs beside the scoring / decoding of the current chunk.
int pcl_frames_stage(pcl_ctx *ctx, int64_t F, int D, const float *frames) {
    if (!ctx) return PCL_ERR_INVALID;
    if (F <= 0 || D <= 0 || !frames) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_frames_stage: bad shape F=%lld D=%d", (long long)F, D);
    const int Dd = device_dim(D);
    if (Dd < 0) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_frames_stage: feature dimension %d > 64 is not supported", D);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int back = ctx->frames_front == 0 ? 1 : 0;
    const size_t n = (size_t)F * Dd;
    if (n > ctx->frames_slot_cap[back]) {                          // (grows only: a steady stream of equal chunks allocates twice)
        dev_free(ctx->frames_slot[back]);
        ctx->frames_slot_cap[back] = 0;
        TRY(dev_alloc(ctx, &ctx->frames_slot[back], n));
        ctx->frames_slot_cap[back] = n;
    }
    if (!ctx->ev_stage) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming));
    if (ctx->have_slot_free) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_slot_free, 0));   // its last readers
    if (Dd == D) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->frames_slot[back], frames, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream_aux));
    } else {
        // a dimension without a kernel instance of its own (D = 8, 20, 40, ...): rows are padded to the device dimension on the way
        // in -- the pad columns zeroed, the rows copied with a pitch -- as pcl_frames_upload pads on the host
        HIPCHK(ctx, hipMemsetAsync(ctx->frames_slot[back], 0, n * sizeof(float), ctx->stream_aux));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->frames_slot[back], (size_t)Dd * sizeof(float), frames, (size_t)D * sizeof(float), (size_t)D * sizeof(float),
                                     (size_t)F, hipMemcpyHostToDevice, ctx->stream_aux));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_stage, ctx->stream_aux));
    ctx->staged_slot = back;
    ctx->staged_F = F;
    ctx->staged_D = D;
    return PCL_OK;
}

int pcl_frames_swap(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (ctx->staged_slot < 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_frames_swap: nothing staged (pcl_frames_stage first)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_stage));               // the copy is done: the caller's buffer is free again
    release_frames32(ctx);
    dev_free(ctx->frames64);
    if (!ctx->ev_slot_free) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_slot_free, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_slot_free, ctx->stream));   // everything queued so far read the old slot
    ctx->have_slot_free = true;
    ctx->frames_front = ctx->staged_slot;
    ctx->frames32 = ctx->frames_slot[ctx->frames_front];
    ctx->F = ctx->staged_F;
    ctx->FDhost = ctx->staged_D;
    ctx->FD = device_dim(ctx->staged_D);
    ctx->staged_slot = -1;
    return PCL_OK;
}

int pcl_host_alloc(pcl_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out || !bytes) return PCL_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return PCL_OK;
}

// hipHostFree does not wait for copies that still read or write the block: a result set queued by pcl_batch_fetch_async on stream_d2h
// (or a chunk staged by pcl_frames_stage) and never waited for would be DMA into unmapped host memory -- a GPU-side fault at the next
// teardown (round 5 saw one at a fixture's engine.close(), which freed its page-locked result buffers before pcl_destroy drained the
// streams).  Freeing page-locked memory is rare (engine close, a staging buffer outgrown): drain every stream of the context first.
int pcl_host_free(pcl_ctx *ctx, void *ptr) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ptr) return PCL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t all[] = {ctx->stream_d2h, ctx->stream_aux, ctx->stream_dp, ctx->stream, ctx->stream_desc, ctx->stream_comm};
    for (hipStream_t s : all)
        if (s) HIPCHK(ctx, hipStreamSynchronize(s));
    HIPCHK(ctx, hipHostFree(ptr));
    return PCL_OK;
}

// ================================================================ batch
int pcl_batch_create(pcl_ctx *ctx, int U, const int32_t *N, const int32_t *T, const int64_t *frame_begin, pcl_batch **out) {
    if (!ctx || !out) return PCL_ERR_INVALID;
    *out = nullptr;
    if (U <= 0 || !N || !T) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create: bad arguments (U=%d)", U);
    // several kernels index the utterance with the grid's second dimension (HIP: at most 65535): say so here instead of failing a launch later
    if (U > 65535) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create: %d utterances in one batch, at most 65535 (split the batch)", U);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pcl_batch_reap(ctx, false);                              // destroyed batches the GPU has finished with: their blocks first
    pcl_batch *b = new pcl_batch();
    b->ctx = ctx;
    b->U = U;
    b->utt.resize(U);
    long long bo = 0, mo = 0, vo = 0, po = 0, to = 0;
    for (int u = 0; u < U; ++u) {
        if (N[u] < 1 || T[u] < 1) {
            delete b;
            PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create: utterance %d has N=%d T=%d", u, N[u], T[u]);
        }
        if (frame_begin && (frame_begin[u] < 0 || frame_begin[u] + T[u] > ctx->F)) {
            delete b;
            PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create: utterance %d frames [%lld,%lld) outside the uploaded %lld frames", u,
                     (long long)frame_begin[u], (long long)frame_begin[u] + T[u], (long long)ctx->F);
        }
        UttDesc &d = b->utt[u];
        d.b_off = bo;
        d.mat_off = mo;
        d.frame0 = frame_begin ? frame_begin[u] : -1;
        if (frame_begin) b->max_frame_end = std::max(b->max_frame_end, (long long)frame_begin[u] + T[u]);
        d.T = T[u];
        d.N = N[u];
        d.vec_off = (int)vo;
        d.ptr_off = (int)po;
        d.nnz_off = 0;
        d.path_off = (int)to;
        bo += (long long)N[u] * T[u];
        mo += (long long)N[u] * N[u];
        vo += N[u];
        po += N[u] + 1;
        to += T[u];
        b->Nmax = std::max(b->Nmax, (int)N[u]);
        b->Tmax = std::max(b->Tmax, (int)T[u]);
        if (T[u] == 1) b->has_one_frame = true;
    }
    b->sumNT = bo;
    b->sumNN = mo;
    b->sumN = vo;
    b->sumT = to;
    if (vo > 0x7fffffffLL || to > 0x7fffffffLL) {
        delete b;
        PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_create: batch too large");
    }
    int r = PCL_OK;
    auto A = [&](int rr) { if (r == PCL_OK) r = rr; };
    A(dev_alloc(ctx, &b->d_utt, (size_t)U));
    A(dev_alloc(ctx, &b->Bt, (size_t)bo));
    A(dev_alloc(ctx, &b->logpi, (size_t)vo));
    A(dev_alloc(ctx, &b->d_row_state, (size_t)vo));
    if (r != PCL_OK) {
        pcl_batch_destroy(b);
        return r;
    }
    *out = b;
    return PCL_OK;
}

// Everything a batch owns goes back to the pool.  The caller has made sure the GPU is done with the batch.
static void batch_free_now(pcl_batch *b) {
    pcl_free_synced_scope done;                              // the frees below skip their device-wide wait
    if (b->ev_fetch) hipEventDestroy(b->ev_fetch);
    if (b->ev_fetch_src) hipEventDestroy(b->ev_fetch_src);
    if (b->ev_dp) hipEventDestroy(b->ev_dp);
    if (b->ev_main) hipEventDestroy(b->ev_main);
    if (b->ev_mark) hipEventDestroy(b->ev_mark);
    pcl_batch_units_release(b);
    pcl_batch_decode_release(b);
    dev_free(b->d_utt); dev_free(b->Bt); dev_free(b->alpha); dev_free(b->beta); dev_free(b->lgam);
    dev_free(b->logpi); dev_free(b->pi_out); dev_free(b->gamma_out); dev_free(b->ksai);
    dev_free(b->logp); dev_free(b->qtrace); dev_free(b->point); dev_free(b->npass); dev_free(b->path);
    dev_free(b->row_ptr); dev_free(b->col_idx); dev_free(b->csr_val);
    dev_free(b->col_ptr); dev_free(b->row_idx); dev_free(b->csc_val);
    dev_free(b->xi_m); dev_free(b->xi_s); dev_free(b->bp); dev_free(b->d_row_state);
    dev_free(b->Bp); dev_free(b->alpha_e); dev_free(b->beta_e); dev_free(b->fb_kmax); dev_free(b->fb_dump); dev_free(b->fb_part_m); dev_free(b->fb_part_e);
    dev_free(b->d_dups);
    dev_free(b->d_segs); dev_free(b->d_seg_of_row); dev_free(b->d_tiles); dev_free(b->d_tiles_v); dev_free(b->d_tiles_s); dev_free(b->d_tiles_c); dev_free(b->d_tile_flags_c); dev_free(b->d_tile_flags); dev_free(b->tmp); dev_free(b->nz_tmp);
    delete b;
}

// has the GPU finished the batch's own work on every stream it used?  (wait: block until it has)
static bool batch_work_done(pcl_batch *b, bool wait) {
    hipEvent_t evs[3] = {b->ev_mark, b->ev_dp, b->ev_fetch};
    bool done = true;
    for (hipEvent_t ev : evs) {
        if (!ev) continue;
        const hipError_t e = wait ? hipEventSynchronize(ev) : hipEventQuery(ev);      // (an event never recorded reads as complete)
        if (e != hipSuccess) done = false;
        if (!done && !wait) break;
    }
    (void)hipGetLastError();                                 // (hipErrorNotReady is not an error)
    return done;
}

static int pcl_batch_reap(pcl_ctx *ctx, bool wait) {
    size_t keep = 0;
    for (size_t g = 0; g < ctx->graves.size(); ++g) {
        pcl_batch *b = ctx->graves[g];
        if (batch_work_done(b, wait)) batch_free_now(b);
        else ctx->graves[keep++] = b;
    }
    ctx->graves.resize(keep);
    return (int)keep;
}

// The reference drops an utterance's objects when its worker returns (AcousticModel.py:884-916); a corpus sweep here drops the batch
// of step k - 3 while the GPU works on step k.  Waiting for the streams at that point (what this function did through round 4:
// 38 ms per call inside a sweep, tools/fresh_batch_probe.py) stalls the host that should be queueing step k + 1.  So the batch
// is freed at once when ITS OWN last work has completed -- the events it left behind its main-stream work (pcl_batch_mark), its
// recursion on the second stream and its result copies, not what later batches queued behind them -- and otherwise waits in the
// context's list until it has (pcl_batch_reap).  The handle is dead for the caller either way.
int pcl_batch_destroy(pcl_batch *b) {
    if (!b) return PCL_OK;
    pcl_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    pcl_batch_reap(ctx, false);
    static const bool sync_destroy = getenv("PCL_DESTROY_SYNC") && atoi(getenv("PCL_DESTROY_SYNC")) != 0;   // A/B: rounds 1-4 (wait here)
    if (batch_work_done(b, sync_destroy)) batch_free_now(b);
    else ctx->graves.push_back(b);
    return PCL_OK;
}

// Upload the sparse transition structure of every utterance (CSR successors / CSC predecessors, both with ascending
// indices; entries with ln A = -inf are not stored) and ln pi.  row_ptr / col_ptr: sumN + U entries (N_u + 1 per
// utterance, local offsets); the UttDesc nnz_off fields must already be set.
int pcl_batch_upload_sparse(pcl_batch *b, const std::vector<int> &row_ptr, const std::vector<int> &col_idx,
                            const std::vector<double> &csr_val, const std::vector<int> &col_ptr,
                            const std::vector<int> &row_idx, const std::vector<double> &csc_val, const double *logpi) {
    pcl_ctx *ctx = b->ctx;
    b->nnz = (long long)col_idx.size();
    // left to right: state i is entered from i - 1 and from itself only (what AcousticModel.embedded builds, AcousticModel.py:979-989)
    b->left_right = true;
    for (int u = 0; u < b->U && b->left_right; ++u) {
        const UttDesc &d = b->utt[u];
        for (int i = 0; i < d.N && b->left_right; ++i)
            for (int k = row_ptr[d.ptr_off + i]; k < row_ptr[d.ptr_off + i + 1]; ++k) {
                const int j = col_idx[(size_t)d.nnz_off + k];
                if (j != i && j != i + 1) {
                    b->left_right = false;
                    break;
                }
            }
    }
    dev_free(b->row_ptr); dev_free(b->col_idx); dev_free(b->csr_val);
    dev_free(b->col_ptr); dev_free(b->row_idx); dev_free(b->csc_val);
    dev_free(b->xi_m); dev_free(b->xi_s); dev_free(b->nz_tmp);
    const size_t nz = (size_t)b->nnz, np = row_ptr.size();
    TRY(dev_alloc(ctx, &b->row_ptr, np));
    TRY(dev_alloc(ctx, &b->col_ptr, np));
    TRY(dev_alloc(ctx, &b->col_idx, nz));
    TRY(dev_alloc(ctx, &b->row_idx, nz));
    TRY(dev_alloc(ctx, &b->csr_val, nz));
    TRY(dev_alloc(ctx, &b->csc_val, nz));
    TRY(dev_alloc(ctx, &b->xi_m, nz));
    TRY(dev_alloc(ctx, &b->xi_s, nz));
    // (a batch that has launched nothing: its buffers are fresh, the copies need not queue behind the main stream's kernels)
    auto up = b->launched ? pcl_h2d : pcl_h2d_fresh;
    HIPCHK(ctx, up(ctx, b->row_ptr, row_ptr.data(), np * sizeof(int)));
    HIPCHK(ctx, up(ctx, b->col_ptr, col_ptr.data(), np * sizeof(int)));
    if (nz) {
        HIPCHK(ctx, up(ctx, b->col_idx, col_idx.data(), nz * sizeof(int)));
        HIPCHK(ctx, up(ctx, b->row_idx, row_idx.data(), nz * sizeof(int)));
        HIPCHK(ctx, up(ctx, b->csr_val, csr_val.data(), nz * sizeof(double)));
        HIPCHK(ctx, up(ctx, b->csc_val, csc_val.data(), nz * sizeof(double)));
    }
    HIPCHK(ctx, up(ctx, b->logpi, logpi, (size_t)b->sumN * sizeof(double)));
    HIPCHK(ctx, up(ctx, b->d_utt, b->utt.data(), (size_t)b->U * sizeof(UttDesc)));
    b->have_trans = true;
    b->have_fb = b->have_vit = false;
    return PCL_OK;
}

int pcl_batch_set_transitions(pcl_batch *b, const double *logA, const double *logpi) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (!logA || !logpi) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_transitions: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // sparse structure: an entry is stored unless ln A == -inf
    std::vector<int> row_ptr((size_t)b->sumN + b->U), col_ptr((size_t)b->sumN + b->U);
    std::vector<int> col_idx, row_idx;
    std::vector<double> csr_val, csc_val;
    b->max_outdeg = b->max_indeg = 0;
    for (int u = 0; u < b->U; ++u) {
        UttDesc &d = b->utt[u];
        const int N = d.N;
        const double *A = logA + d.mat_off;
        d.nnz_off = (int)col_idx.size();
        int cnt = 0;
        for (int i = 0; i < N; ++i) {
            row_ptr[d.ptr_off + i] = cnt;
            for (int j = 0; j < N; ++j) {
                const double v = A[(size_t)i * N + j];
                if (!(v == -INFINITY)) {
                    col_idx.push_back(j);
                    csr_val.push_back(v);
                    ++cnt;
                }
            }
            b->max_outdeg = std::max(b->max_outdeg, cnt - row_ptr[d.ptr_off + i]);
        }
        row_ptr[d.ptr_off + N] = cnt;
        int cc = 0;
        for (int j = 0; j < N; ++j) {
            col_ptr[d.ptr_off + j] = cc;
            for (int i = 0; i < N; ++i) {
                const double v = A[(size_t)i * N + j];
                if (!(v == -INFINITY)) {
                    row_idx.push_back(i);
                    csc_val.push_back(v);
                    ++cc;
                }
            }
            b->max_indeg = std::max(b->max_indeg, cc - col_ptr[d.ptr_off + j]);
        }
        col_ptr[d.ptr_off + N] = cc;
        if (col_idx.size() > 0x7fffffffULL) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_transitions: too many transitions");
    }
    return pcl_batch_upload_sparse(b, row_ptr, col_idx, csr_val, col_ptr, row_idx, csc_val, logpi);
}

int pcl_batch_set_states(pcl_batch *b, const int32_t *row_state) {
    if (!b) return PCL_ERR_INVALID;
    TRY(batch_join(b));
    if (!row_state) PCL_FAIL(b->ctx, PCL_ERR_INVALID, "pcl_batch_set_states: NULL argument");
    return pcl_batch_set_states_impl(b, row_state);
}

int pcl_batch_set_states_impl(pcl_batch *b, const int32_t *row_state) {
    pcl_ctx *ctx = b->ctx;
    if (ctx->J == 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_set_states: upload a model first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    b->row_state.assign(row_state, row_state + b->sumN);
    // count segments per state
    std::vector<int> count(ctx->J, 0);
    int max_state = -1;
    for (int u = 0; u < b->U; ++u) {
        const UttDesc &d = b->utt[u];
        for (int n = 0; n < d.N; ++n) {
            const int st = row_state[d.vec_off + n];
            if (st >= ctx->J || st < PCL_ROW_EXIT) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_states: utterance %d row %d has state %d outside [0,%d)", u, n, st, ctx->J);
            if (st >= 0) {
                if (d.frame0 < 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_set_states: batch was created without frame_begin");
                ++count[st];
                max_state = std::max(max_state, st);
            }
        }
    }
    std::vector<int> start(ctx->J + 1, 0);
    for (int j = 0; j < ctx->J; ++j) start[j + 1] = start[j] + count[j];
    b->n_segs = start[ctx->J];
    b->segs.assign(b->n_segs, ScoreSeg());
    // A label that names a unit twice has the same (frames, state) pair on two rows: the reference scores it once per label
    // position (AcousticModel.py:897-902).  Here the FIRST row of a state in an utterance is scored (a state's segments list
    // those first: scoring tiles cover [lo, hip)), the others are copies of its emission row (dup_rows_kernel); the accumulate
    // pass walks all of a state's segments, each row has its own posteriors.
    std::vector<int> fill(start.begin(), start.end() - 1), nprim(ctx->J, 0);
    std::vector<long long> vtot(ctx->J, 0);
    std::vector<int> first_row(ctx->J, -1), touched;
    std::vector<char> is_dup((size_t)b->sumN, 0);
    b->dups.clear();
    for (int u = 0; u < b->U; ++u) {
        const UttDesc &d = b->utt[u];
        touched.clear();
        for (int n = 0; n < d.N; ++n) {
            const int st = row_state[d.vec_off + n];
            if (st < 0) continue;
            if (first_row[st] < 0) {
                first_row[st] = n;
                touched.push_back(st);
                ++nprim[st];
            } else {
                is_dup[d.vec_off + n] = 1;
                b->dups.push_back(DupRow{d.b_off + first_row[st], d.b_off + n, d.T, d.N});
            }
        }
        for (int st : touched) first_row[st] = -1;
    }
    b->seg_of_row.assign((size_t)b->sumN, -1);
    b->max_N = 0;
    for (int u = 0; u < b->U; ++u) b->max_N = std::max(b->max_N, b->utt[u].N);
    for (int pass = 0; pass < 2; ++pass)                         // the scored rows of every state first, then the copies
        for (int u = 0; u < b->U; ++u) {
            const UttDesc &d = b->utt[u];
            for (int n = 0; n < d.N; ++n) {
                const int st = row_state[d.vec_off + n];
                if (st < 0 || (int)is_dup[d.vec_off + n] != pass) continue;
                b->seg_of_row[d.vec_off + n] = fill[st];
                ScoreSeg &s = b->segs[fill[st]++];
                s.frame0 = d.frame0;
                s.out0 = d.b_off + n;
                s.len = d.T;
                s.out_stride = d.N;
                if (vtot[st] + d.T > 0x7fffffffLL) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_states: state %d has too many frames", st);
                s.vstart = (int)vtot[st];
                s.pad = st;
                vtot[st] += d.T;
            }
        }
    b->work_states.clear();
    b->state_seg_lo.clear();
    b->state_seg_hi.clear();
    b->state_seg_hip.clear();
    for (int j = 0; j < ctx->J; ++j)
        if (count[j]) {
            b->work_states.push_back(j);
            b->state_seg_lo.push_back(start[j]);
            b->state_seg_hi.push_back(start[j + 1]);
            b->state_seg_hip.push_back(start[j] + nprim[j]);
        }
    dev_free(b->d_dups);
    if (!b->dups.empty()) {
        TRY(dev_alloc(ctx, &b->d_dups, b->dups.size()));
        HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_dups, b->dups.data(), b->dups.size() * sizeof(DupRow)));
    }
    dev_free(b->d_segs);
    dev_free(b->d_tiles);
    dev_free(b->d_tiles_v);
    dev_free(b->d_tiles_s);
    dev_free(b->d_tiles_c);
    dev_free(b->d_tile_flags_c);
    b->n_tiles_s = b->n_tiles_c = 0;
    b->tile_frames = 0;
    TRY(dev_alloc(ctx, &b->d_segs, (size_t)b->n_segs));
    if (b->n_segs) HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_segs, b->segs.data(), (size_t)b->n_segs * sizeof(ScoreSeg)));
    dev_free(b->d_seg_of_row);
    TRY(dev_alloc(ctx, &b->d_seg_of_row, (size_t)b->sumN));
    if (b->sumN) HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_seg_of_row, b->seg_of_row.data(), (size_t)b->sumN * sizeof(int)));
    auto up = b->launched ? pcl_h2d : pcl_h2d_fresh;             // (batch-lifetime buffers: fresh only while nothing was launched)
    HIPCHK(ctx, up(ctx, b->d_row_state, row_state, (size_t)b->sumN * sizeof(int32_t)));
    HIPCHK(ctx, up(ctx, b->d_utt, b->utt.data(), (size_t)b->U * sizeof(UttDesc)));
    b->have_states = true;
    b->virt_rows_filled = false;
    b->max_state = max_state;
    b->model_J = ctx->J;
    return PCL_OK;
}

// A batch outlives pcl_frames_upload / pcl_model_upload calls (the drop-in classes re-upload on the shared engine all
// the time): its frame rows and state ids were checked against the buffers of THEN.  Re-check against the buffers of
// NOW before any kernel indexes them.
static int batch_revalidate(pcl_batch *b, const char *who) {
    pcl_ctx *ctx = b->ctx;
    if (b->max_frame_end > ctx->F)
        PCL_FAIL(ctx, PCL_ERR_STATE, "%s: the batch refers to frame rows up to %lld but the frame matrix now has %lld rows (re-uploaded after the batch was created)",
                 who, b->max_frame_end, (long long)ctx->F);
    if (b->have_states && (b->max_state >= ctx->J || b->model_J != ctx->J))
        PCL_FAIL(ctx, PCL_ERR_STATE, "%s: the batch was laid out for a model of %d states (largest id used %d) but the model now has %d (re-uploaded after pcl_batch_set_states)",
                 who, b->model_J, b->max_state, ctx->J);
    return PCL_OK;
}

static int ensure_tmp(pcl_batch *b) {
    if (!b->tmp) TRY(dev_alloc(b->ctx, &b->tmp, (size_t)b->sumNT));
    return PCL_OK;
}

int pcl_batch_set_emissions(pcl_batch *b, const double *B) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (!B) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_emissions: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TRY(ensure_tmp(b));
    b->launched = true;
    HIPCHK(ctx, pcl_h2d(ctx, b->d_utt, b->utt.data(), (size_t)b->U * sizeof(UttDesc)));
    HIPCHK(ctx, hipMemcpyAsync(b->tmp, B, (size_t)b->sumNT * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(pcl_launch_transpose(ctx, b, b->tmp, b->Bt, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    b->have_B = true;
    b->virt_rows_filled = false;                                   // (the caller's matrix is in the buffer now)
    b->have_fb = b->have_vit = false;
    return PCL_OK;
}

int pcl_batch_set_posteriors(pcl_batch *b, const double *lgamma) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (!lgamma) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_set_posteriors: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TRY(ensure_tmp(b));
    if (!b->lgam) TRY(dev_alloc(ctx, &b->lgam, (size_t)b->sumNT));
    HIPCHK(ctx, pcl_h2d(ctx, b->d_utt, b->utt.data(), (size_t)b->U * sizeof(UttDesc)));
    HIPCHK(ctx, hipMemcpyAsync(b->tmp, lgamma, (size_t)b->sumNT * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(pcl_launch_transpose(ctx, b, b->tmp, b->lgam, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    b->have_post = true;
    return PCL_OK;
}

static int ensure_frames64(pcl_ctx *ctx) {
    if (ctx->frames64 || !ctx->frames32) return PCL_OK;
    const size_t n = (size_t)ctx->F * ctx->FD;
    TRY(dev_alloc(ctx, &ctx->frames64, n));
    return pcl_launch_cast(ctx, nullptr, ctx->frames32, ctx->frames64, n);   // float -> double is exact
}

// XCD-aware tile order for a subset of the batch's states: workgroups are dealt round-robin over the 8 XCDs
// (block b -> XCD b % 8, observed, speed only), so all tiles of one state are placed at block indices with the
// same residue mod 8: the state's parameter block is then fetched into ONE XCD's L2 instead of eight.
// Shorter queues are padded with empty tiles (seg_lo == seg_hi), which exit immediately.
static std::vector<ScoreTile> make_tiles(const pcl_batch *b, const std::vector<size_t> &which, int tf) {
    constexpr int NXCD = 8;
    std::vector<ScoreTile> queue[NXCD];
    std::vector<long long> load(NXCD, 0);
    for (size_t k : which) {
        const int lo = b->state_seg_lo[k], hi = b->state_seg_hip[k];      // (the scored rows; the copies sit behind them)
        const long long tot = (long long)b->segs[hi - 1].vstart + b->segs[hi - 1].len;
        int g = 0;
        for (int x = 1; x < NXCD; ++x)
            if (load[x] < load[g]) g = x;          // least-loaded XCD queue
        int s0 = lo;
        for (long long v = 0; v < tot; v += tf) {
            while (s0 + 1 < hi && b->segs[s0 + 1].vstart <= v) ++s0;
            queue[g].push_back(ScoreTile{b->work_states[k], lo, hi, (int)v, s0});
        }
        load[g] += (tot + tf - 1) / tf;
    }
    size_t depth = 0;
    for (int x = 0; x < NXCD; ++x) depth = std::max(depth, queue[x].size());
    std::vector<ScoreTile> tiles;
    tiles.reserve(depth * NXCD);
    for (size_t q = 0; q < depth; ++q)
        for (int x = 0; x < NXCD; ++x) tiles.push_back(q < queue[x].size() ? queue[x][q] : ScoreTile{0, 0, 0, 0, 0});
    return tiles;
}

static int build_tiles(pcl_batch *b, int precision) {
    pcl_ctx *ctx = b->ctx;
    const bool mfma = precision == PCL_F32 && ctx->score_variant >= 3 && pcl_score_mfma_supported(ctx->D);
    const int tf = mfma ? (ctx->score_variant == 7 ? pcl_score_split16_tile_frames() : pcl_score_mfma_tile_frames())
                        : pcl_score_tile_frames(ctx->D, precision);
    if (b->d_tiles && b->tile_frames == tf && b->tile_gen == ctx->model_gen) return PCL_OK;
    // MFMA mode: states whose centred expansion is ill conditioned go to the direct-form VALU kernel
    std::vector<size_t> good, bad;
    for (size_t k = 0; k < b->work_states.size(); ++k) (mfma && pcl_state_uses_valu(ctx, b->work_states[k]) ? bad : good).push_back(k);
    const std::vector<ScoreTile> tiles = make_tiles(b, good, tf);
    const std::vector<ScoreTile> tiles_v = bad.empty() ? std::vector<ScoreTile>() : make_tiles(b, bad, pcl_score_tile_frames(ctx->D, PCL_F32));
    // split states: on the matrix pipe (in `good`) AND, for their off-pipe mixtures, in a list of their own at the direct-form tile size
    std::vector<size_t> split;
    if (mfma)
        for (size_t k : good)
            if (pcl_state_is_split(ctx, b->work_states[k])) split.push_back(k);
    // ... at the direct-form tile size (the subset launch, rounds 4-5) or, with the coarse pass, at the matrix pipe's
    const bool coarse = mfma && ctx->score_variant == 7 && pcl_coarse_enabled(ctx);
    const std::vector<ScoreTile> tiles_s = (split.empty() || coarse) ? std::vector<ScoreTile>() : make_tiles(b, split, pcl_score_subset_tile_frames(ctx->D));
    const std::vector<ScoreTile> tiles_c = (split.empty() || !coarse) ? std::vector<ScoreTile>() : make_tiles(b, split, pcl_coarse_tile_frames());
    dev_free(b->d_tiles);
    dev_free(b->d_tiles_v);
    dev_free(b->d_tiles_s);
    dev_free(b->d_tiles_c);
    dev_free(b->d_tile_flags_c);
    pcl_desc_group uploads(ctx);                                   // the tile lists: staged, one wait at the end
    b->n_tiles_s = (int)tiles_s.size();
    if (!tiles_s.empty()) {
        TRY(dev_alloc(ctx, &b->d_tiles_s, tiles_s.size()));
        HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_tiles_s, tiles_s.data(), tiles_s.size() * sizeof(ScoreTile)));
    }
    b->n_tiles_c = (int)tiles_c.size();
    if (!tiles_c.empty()) {
        TRY(dev_alloc(ctx, &b->d_tiles_c, tiles_c.size()));
        TRY(dev_alloc(ctx, &b->d_tile_flags_c, tiles_c.size()));
        HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_tiles_c, tiles_c.data(), tiles_c.size() * sizeof(ScoreTile)));
    }
    dev_free(b->d_tile_flags);
    TRY(dev_alloc(ctx, &b->d_tile_flags, tiles.size()));
    b->n_tiles = (int)tiles.size();
    b->n_tiles_v = (int)tiles_v.size();
    b->tile_frames = tf;
    b->tile_gen = ctx->model_gen;
    TRY(dev_alloc(ctx, &b->d_tiles, tiles.size()));
    if (!tiles.empty()) HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_tiles, tiles.data(), tiles.size() * sizeof(ScoreTile)));
    if (!tiles_v.empty()) {
        TRY(dev_alloc(ctx, &b->d_tiles_v, tiles_v.size()));
        HIPCHK(ctx, pcl_h2d_fresh(ctx, b->d_tiles_v, tiles_v.data(), tiles_v.size() * sizeof(ScoreTile)));
    }
    HIPCHK(ctx, uploads.finish());
    return PCL_OK;
}

int pcl_batch_score(pcl_batch *b, int precision) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (precision != PCL_F32 && precision != PCL_F64) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_score: precision %d", precision);
    if (ctx->J == 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_score: no model uploaded");
    if (ctx->F == 0) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_score: no frames uploaded");
    if (!b->have_states) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_score: pcl_batch_set_states was not called");
    if (ctx->FDhost != ctx->Dhost)  // DataDimensionError, Clustering.py:749-751
        PCL_FAIL(ctx, PCL_ERR_INVALID, "data dimension %d does not match model dimension %d", ctx->FDhost, ctx->Dhost);
    TRY(batch_revalidate(b, "pcl_batch_score"));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (precision == PCL_F64) {
        TRY(ensure_frames64(ctx));
        TRY(pcl_ensure_layouts(ctx, PCL_LAYOUT_P64));
    }
    TRY(build_tiles(b, precision));
    b->launched = true;
    if (!b->virt_rows_filled) {                                    // entry row ln 1, exit row ln 0 (AcousticModel.py:218-219): constants,
        TRY(pcl_launch_fill_virtual_rows(ctx, b));                // written once per row map, not once per scoring pass
        b->virt_rows_filled = true;
    }
    if (precision == PCL_F32 && ctx->score_variant >= 3 && pcl_score_mfma_supported(ctx->D)) {
        if (ctx->score_variant == 7) {
            TRY(pcl_launch_score_split16(ctx, b, b->d_tiles, b->n_tiles));
#ifndef PCL_DIAG_NOFIXUP
            TRY(pcl_launch_score_fixup(ctx, b, b->d_tiles, b->n_tiles, b->d_tile_flags));   // tiles with out-of-range features
#endif
        } else TRY(pcl_launch_score_mfma(ctx, b, b->d_tiles, b->n_tiles));
        if (b->n_tiles_c) {                                                       // split states: their off-pipe mixtures, log-added --
            TRY(pcl_launch_score_coarse(ctx, b));                                 // proven negligible by a bound on the matrix pipe, or evaluated exactly
            TRY(pcl_launch_score_subset_flagged(ctx, b, b->d_tiles_c, b->n_tiles_c, b->d_tile_flags_c));   // (tiles with features out of the f16 range)
        }
        TRY(pcl_launch_score_subset(ctx, b, b->d_tiles_s, b->n_tiles_s));         // ... or all of them in direct form (PCL_COARSE=0)
        TRY(pcl_launch_score(ctx, b, PCL_F32, b->d_tiles_v, b->n_tiles_v));      // ill-conditioned states, direct form
    } else {
        TRY(pcl_launch_score(ctx, b, precision, b->d_tiles, b->n_tiles));
    }
    TRY(pcl_launch_dup_rows(ctx, b));                              // rows of a state an utterance's label names again
    HIPCHK(ctx, pcl_batch_mark(b));
    b->mark_is_score = true;
    b->have_B = true;
    b->have_fb = b->have_vit = false;
    return PCL_OK;
}

int pcl_batch_forward_backward(pcl_batch *b, int fix_pi, double threshold) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!b->have_trans) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_forward_backward: no transitions set");
    if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_forward_backward: no emissions (score or set_emissions first)");  // LHMM.py:69
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool after_fetch = b->fetch_pending;
    if (b->fetch_pending) {                                  // the copies of the previous results read lgam / ksai: main stream first (stream_dp waits for them below)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b->ev_fetch, 0));
        b->fetch_pending = false;
    }
    if (!b->alpha) {
        TRY(dev_alloc(ctx, &b->alpha, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->beta, (size_t)b->sumNT));
        if (!b->lgam) TRY(dev_alloc(ctx, &b->lgam, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->pi_out, (size_t)b->sumN));
        TRY(dev_alloc(ctx, &b->gamma_out, (size_t)b->sumN));
        TRY(dev_alloc(ctx, &b->ksai, (size_t)b->sumNN));
        TRY(dev_alloc(ctx, &b->logp, (size_t)b->U));
        TRY(dev_alloc(ctx, &b->qtrace, (size_t)b->U * PCL_MAX_PASS));
        TRY(dev_alloc(ctx, &b->npass, (size_t)b->U));
    }
    if (ctx->dp_async) {
        // stream_dp waits for everything queued on the main stream so far (the scoring of this batch), runs the
        // recursion, and leaves an event for whoever touches the batch next
        if (!b->ev_dp) HIPCHK(ctx, hipEventCreateWithFlags(&b->ev_dp, hipEventDisableTiming));
        HIPCHK(ctx, pcl_dp_follows_main(b, after_fetch));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream_dp;                      // the launcher and its timer use ctx->stream
        const int rc = pcl_launch_forward_backward(ctx, b, fix_pi ? 1 : 0, threshold);
        ctx->stream = main_stream;
        if (rc != PCL_OK) return rc;
        HIPCHK(ctx, hipEventRecord(b->ev_dp, ctx->stream_dp));
        b->dp_pending = true;
    } else {
        TRY(pcl_launch_forward_backward(ctx, b, fix_pi ? 1 : 0, threshold));
        HIPCHK(ctx, pcl_batch_mark(b));
        b->mark_is_score = false;
    }
    b->have_fb = true;
    b->have_post = true;
    return PCL_OK;
}

int pcl_batch_viterbi(pcl_batch *b, int end_state_back) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!b->have_trans) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_viterbi: no transitions set");
    if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_viterbi: no emissions (score or set_emissions first)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool after_fetch = b->fetch_pending;
    if (b->fetch_pending) {                                  // result copies still reading this batch's path / point buffers
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b->ev_fetch, 0));
        b->fetch_pending = false;
    }
    if (!b->bp) {
        TRY(dev_alloc(ctx, &b->bp, (size_t)b->sumNT));
        TRY(dev_alloc(ctx, &b->path, (size_t)b->sumT));
        TRY(dev_alloc(ctx, &b->point, (size_t)b->U));
    }
    if (ctx->dp_async) {
        // like the forward-backward: on the second stream, behind everything the main stream has queued (this batch's scoring), beside
        // the NEXT batch's scoring; in order with a forward-backward of the same batch already queued there (round 4: on the main
        // stream the 0.35 ms recursion sat between two scoring kernels -- config 3's step is score + Viterbi)
        if (!b->ev_dp) HIPCHK(ctx, hipEventCreateWithFlags(&b->ev_dp, hipEventDisableTiming));
        HIPCHK(ctx, pcl_dp_follows_main(b, after_fetch));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream_dp;                      // the launcher and its timer use ctx->stream
        const int rc = pcl_launch_viterbi(ctx, b, end_state_back ? 1 : 0);
        ctx->stream = main_stream;
        if (rc != PCL_OK) return rc;
        HIPCHK(ctx, hipEventRecord(b->ev_dp, ctx->stream_dp));
        b->dp_pending = true;
    } else {
        TRY(batch_join(b));
        TRY(pcl_launch_viterbi(ctx, b, end_state_back ? 1 : 0));
        HIPCHK(ctx, pcl_batch_mark(b));
        b->mark_is_score = false;
    }
    b->have_vit = true;
    return PCL_OK;
}

int pcl_batch_regroup(pcl_batch *b, const int32_t *row_unit, int gmm_num, int32_t *frame_unit, int32_t *frame_k) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (!row_unit || !frame_unit || !frame_k || gmm_num < 1) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_regroup: bad arguments");
    if (!b->have_vit) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_regroup: run pcl_batch_viterbi first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t *d_ru = nullptr, *d_fu = nullptr, *d_fk = nullptr;
    int rc = dev_alloc(ctx, &d_ru, (size_t)b->sumN);
    if (rc == PCL_OK) rc = dev_alloc(ctx, &d_fu, (size_t)b->sumT);
    if (rc == PCL_OK) rc = dev_alloc(ctx, &d_fk, (size_t)b->sumT);
    if (rc == PCL_OK && hipMemcpyAsync(d_ru, row_unit, (size_t)b->sumN * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PCL_ERR_HIP;
    if (rc == PCL_OK) rc = pcl_launch_regroup(ctx, b, d_ru, gmm_num, d_fu, d_fk);
    if (rc == PCL_OK && hipMemcpyAsync(frame_unit, d_fu, (size_t)b->sumT * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_ERR_HIP;
    if (rc == PCL_OK && hipMemcpyAsync(frame_k, d_fk, (size_t)b->sumT * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_ERR_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PCL_OK) rc = PCL_ERR_HIP;
    dev_free(d_ru); dev_free(d_fu); dev_free(d_fk);
    if (rc == PCL_ERR_HIP) PCL_FAIL(ctx, PCL_ERR_HIP, "pcl_batch_regroup: HIP error");
    return rc;
}

int pcl_batch_get(pcl_batch *b, int what, void *host) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    if (!host) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_get: NULL host buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double *mat = nullptr;
    switch (what) {
        case PCL_GET_B:
            if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_get: no emissions yet");
            mat = b->Bt;
            break;
        case PCL_GET_ALPHA: mat = b->alpha; break;
        case PCL_GET_BETA: mat = b->beta; break;
        case PCL_GET_LGAMMA: mat = b->lgam; break;
        default: break;
    }
    if (((what >= PCL_GET_ALPHA && what <= PCL_GET_QTRACE) || what == PCL_GET_KSAI_NZ) && !b->have_fb) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_get: run pcl_batch_forward_backward first");
    if ((what == PCL_GET_PATH || what == PCL_GET_POINT) && !b->have_vit) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_get: run pcl_batch_viterbi first");
    if (mat) {
        TRY(ensure_tmp(b));
        double *logs = nullptr;
        if ((what == PCL_GET_ALPHA || what == PCL_GET_BETA) && b->fb_linear) {
            // the scaled forward-backward keeps (mantissa, exponent) pairs; the logarithms the reference holds are made here, on demand
            TRY(dev_alloc(ctx, &logs, (size_t)b->sumNT));
            const int rc = pcl_launch_fb_to_log(ctx, b, mat, what == PCL_GET_ALPHA ? b->alpha_e : b->beta_e, logs);
            if (rc != PCL_OK) {
                dev_free(logs);
                return rc;
            }
            mat = logs;
        }
        const int rt = pcl_launch_transpose(ctx, b, mat, b->tmp, 0);
        if (logs) {
            hipStreamSynchronize(ctx->stream);
            pcl_free_synced_scope done;
            dev_free(logs);
        }
        TRY(rt);
        HIPCHK(ctx, hipMemcpyAsync(host, b->tmp, (size_t)b->sumNT * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return PCL_OK;
    }
    const void *src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case PCL_GET_KSAI: src = b->ksai; bytes = (size_t)b->sumNN * 8; break;
        case PCL_GET_KSAI_NZ:
            if (!b->nz_tmp) TRY(dev_alloc(ctx, &b->nz_tmp, (size_t)b->nnz));
            TRY(pcl_launch_ksai_gather(ctx, b, b->nz_tmp));
            src = b->nz_tmp; bytes = (size_t)b->nnz * 8;
            break;
        case PCL_GET_GAMMA: src = b->gamma_out; bytes = (size_t)b->sumN * 8; break;
        case PCL_GET_PI: src = b->pi_out; bytes = (size_t)b->sumN * 8; break;
        case PCL_GET_LOGP: src = b->logp; bytes = (size_t)b->U * 8; break;
        case PCL_GET_NPASS: src = b->npass; bytes = (size_t)b->U * 4; break;
        case PCL_GET_QTRACE: src = b->qtrace; bytes = (size_t)b->U * PCL_MAX_PASS * 8; break;
        case PCL_GET_PATH: src = b->path; bytes = (size_t)b->sumT * 4; break;
        case PCL_GET_POINT: src = b->point; bytes = (size_t)b->U * 8; break;
        default: PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_get: unknown selector %d", what);
    }
    HIPCHK(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_batch_sizes(pcl_batch *b, int64_t *sum_nt, int64_t *sum_n, int64_t *sum_t, int64_t *nnz) {
    if (!b) return PCL_ERR_INVALID;
    if (sum_nt) *sum_nt = b->sumNT;
    if (sum_n) *sum_n = b->sumN;
    if (sum_t) *sum_t = b->sumT;
    if (nnz) *nnz = b->nnz;
    return PCL_OK;
}

int pcl_batch_fetch_async(pcl_batch *b, double *logp, double *lgamma_tm, double *ksai_nz, int32_t *path, double *point) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if ((logp || lgamma_tm || ksai_nz) && !b->have_fb) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_fetch_async: run pcl_batch_forward_backward first");
    if ((path || point) && !b->have_vit) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_fetch_async: run pcl_batch_viterbi first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!b->ev_fetch) {
        HIPCHK(ctx, hipEventCreateWithFlags(&b->ev_fetch, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&b->ev_fetch_src, hipEventDisableTiming));
    }
    hipStream_t ds = ctx->stream_d2h;
    // behind what the batch has queued: the second stream's recursion (which itself waited for the batch's scoring) when one is
    // pending -- no packet on the main stream then --, else the main stream as of now
    if (b->dp_pending && pcl_fewer_markers()) {
        HIPCHK(ctx, hipStreamWaitEvent(ds, b->ev_dp, 0));
    } else {
        HIPCHK(ctx, hipEventRecord(b->ev_fetch_src, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ds, b->ev_fetch_src, 0));
        if (b->dp_pending) HIPCHK(ctx, hipStreamWaitEvent(ds, b->ev_dp, 0));
    }
    if (ksai_nz) {
        if (!b->nz_tmp) TRY(dev_alloc(ctx, &b->nz_tmp, (size_t)b->nnz));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ds;                                     // (the launcher uses ctx->stream)
        const int rc = pcl_launch_ksai_gather(ctx, b, b->nz_tmp);
        ctx->stream = main_stream;
        if (rc != PCL_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ksai_nz, b->nz_tmp, (size_t)b->nnz * 8, hipMemcpyDeviceToHost, ds));
    }
    if (logp) HIPCHK(ctx, hipMemcpyAsync(logp, b->logp, (size_t)b->U * 8, hipMemcpyDeviceToHost, ds));
    if (lgamma_tm) HIPCHK(ctx, hipMemcpyAsync(lgamma_tm, b->lgam, (size_t)b->sumNT * 8, hipMemcpyDeviceToHost, ds));
    if (path) HIPCHK(ctx, hipMemcpyAsync(path, b->path, (size_t)b->sumT * 4, hipMemcpyDeviceToHost, ds));
    if (point) HIPCHK(ctx, hipMemcpyAsync(point, b->point, (size_t)b->U * 8, hipMemcpyDeviceToHost, ds));
    HIPCHK(ctx, hipEventRecord(b->ev_fetch, ds));
    b->fetch_pending = true;
    return PCL_OK;
}

int pcl_batch_fetch_wait(pcl_batch *b) {
    if (!b) return PCL_ERR_INVALID;
    if (!b->ev_fetch) PCL_FAIL(b->ctx, PCL_ERR_STATE, "pcl_batch_fetch_wait: nothing was fetched");
    HIPCHK(b->ctx, hipEventSynchronize(b->ev_fetch));
    b->fetch_pending = false;
    return PCL_OK;
}

int pcl_clock_probe(pcl_ctx *ctx, int spin_us, double *shader_mhz) {
    if (!ctx || !shader_mhz || spin_us < 1 || spin_us > 1000000) return PCL_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unsigned long long *d = nullptr, h[4] = {0, 0, 0, 0};
    TRY(dev_alloc(ctx, &d, (size_t)4));
    int rc = pcl_launch_clock_probe(ctx, spin_us, d);
    if (rc == PCL_OK && hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream_aux) != hipSuccess) rc = PCL_ERR_HIP;
    if (rc == PCL_OK && hipStreamSynchronize(ctx->stream_aux) != hipSuccess) rc = PCL_ERR_HIP;
    {
        pcl_free_synced_scope done;                           // (the probe is the only user of d and it has finished)
        dev_free(d);
    }
    if (rc != PCL_OK) PCL_FAIL(ctx, rc, "pcl_clock_probe: HIP error");
    const double cyc = (double)(h[1] - h[0]), ref = (double)(h[3] - h[2]);
    *shader_mhz = ref > 0 ? cyc / ref * 100.0 : 0.0;          // s_memrealtime counts at 100 MHz
    return PCL_OK;
}

// ================================================================ E-step statistics
int pcl_stats_zero(pcl_ctx *ctx) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_zero: no model uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    static const bool zero_async = !(getenv("PCL_ZERO_ASYNC") && atoi(getenv("PCL_ZERO_ASYNC")) == 0);      // 0: on the main stream (rounds 1-4; A/B)
    // (a small block -- the one-unit models of the per-object drop-in route -- is cleared in microseconds where it is: the hop to another
    //  queue and back costs more than that)
    if (zero_async && ctx->stream_aux && ctx->stats_len * sizeof(double) >= ((size_t)64 << 20)) {
        // beside whatever the main stream does next (an E-step starts with the scoring of its first batch, which does not touch the block):
        // behind everything queued so far (the block's last readers), on the auxiliary stream
        if (!ctx->ev_zero) {
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_zero, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_zero_src, hipEventDisableTiming));
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev_zero_src, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_zero_src, 0));
        HIPCHK(ctx, hipMemsetAsync(ctx->stats, 0, ctx->stats_len * sizeof(double), ctx->stream_aux));
        HIPCHK(ctx, hipEventRecord(ctx->ev_zero, ctx->stream_aux));
        ctx->zero_pending = true;
    } else {
        HIPCHK(ctx, hipMemsetAsync(ctx->stats, 0, ctx->stats_len * sizeof(double), ctx->stream));
    }
    ctx->stats_fresh = true;
    if (ctx->hmm_ksai) return pcl_hmm_acc_zero(ctx);      // the per-unit transition accumulators restart at ln 0 (LHMM.py:84-85)
    return PCL_OK;
}

// what pcl_batch_accumulate needs of the batch and the context (checked before anything is queued; pcl_batch_accumulate_exchange checks
// it BEFORE it opens the pipe: a pass that cannot run must not be followed by an exchange of incomplete statistics)
static int accumulate_precheck(pcl_batch *b, int precision, const char *who) {
    pcl_ctx *ctx = b->ctx;
    if (precision != PCL_F32 && precision != PCL_F64) PCL_FAIL(ctx, PCL_ERR_INVALID, "%s: precision %d", who, precision);
    if (!b->have_post) PCL_FAIL(ctx, PCL_ERR_STATE, "%s: run pcl_batch_forward_backward (or set_posteriors) first", who);
    if (!b->have_B) PCL_FAIL(ctx, PCL_ERR_STATE, "%s: no emissions", who);
    if (!b->have_states) PCL_FAIL(ctx, PCL_ERR_STATE, "%s: pcl_batch_set_states was not called", who);
    if (ctx->F == 0) PCL_FAIL(ctx, PCL_ERR_STATE, "%s: no frames uploaded", who);
    if (ctx->FDhost != ctx->Dhost) PCL_FAIL(ctx, PCL_ERR_INVALID, "data dimension %d does not match model dimension %d", ctx->FDhost, ctx->Dhost);
    return batch_revalidate(b, who);
}

int pcl_batch_accumulate(pcl_batch *b, int precision) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    TRY(batch_join(b));
    TRY(accumulate_precheck(b, precision, "pcl_batch_accumulate"));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (precision == PCL_F64) {
        TRY(ensure_frames64(ctx));
        TRY(pcl_ensure_layouts(ctx, PCL_LAYOUT_P64));
    }
    HIPCHK(ctx, pcl_stats_join(ctx));
    const int rc = pcl_launch_accumulate(ctx, b, precision);
    ctx->stats_fresh = false;
    if (rc == PCL_OK) HIPCHK(ctx, pcl_batch_mark(b));
    return rc;                                               // (accumulate writes nothing the recursion reads: mark_is_score stays)
}

int pcl_batch_accumulate_exchange(pcl_batch *b, int precision, double c_covariance, int payload, int update_transitions, int n_chunks) {
    if (!b) return PCL_ERR_INVALID;
    pcl_ctx *ctx = b->ctx;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_exchange: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    if (payload != PCL_F64 && payload != PCL_F32) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_accumulate_exchange: payload %d", payload);
    if (n_chunks < 1) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_batch_accumulate_exchange: n_chunks %d", n_chunks);
    if (ctx->transport == 0 && ctx->nranks != 1) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_exchange: pcl_comm_init was not called");
    if (update_transitions && !ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_batch_accumulate_exchange: update_transitions without pcl_units_upload");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TRY(accumulate_precheck(b, precision, "pcl_batch_accumulate_exchange"));   // nothing is exchanged for a pass that cannot run
    TRY(pcl_pipe_begin(ctx, c_covariance, payload, n_chunks));
    int rc = pcl_batch_accumulate(b, precision);                 // releases chunks as its state groups finish
    // (a failure from here on is a HIP / allocation error in the middle of the pass: the pipe is closed -- the collectives stay
    //  matched across the ranks -- and the error is returned; the caller must treat the model as undefined)
    const int rf = pcl_pipe_finish(ctx, update_transitions);     // (always: closes the pipe)
    if (rc == PCL_OK) rc = rf;
    if (rc != PCL_OK) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_accumulate_exchange_idle(pcl_ctx *ctx, double c_covariance, int payload, int update_transitions, int n_chunks) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_accumulate_exchange_idle: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    if (payload != PCL_F64 && payload != PCL_F32) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_accumulate_exchange_idle: payload %d", payload);
    if (n_chunks < 1) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_accumulate_exchange_idle: n_chunks %d", n_chunks);
    if (ctx->transport == 0 && ctx->nranks != 1) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_accumulate_exchange_idle: pcl_comm_init was not called");
    if (update_transitions && !ctx->hmm_ksai) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_accumulate_exchange_idle: update_transitions without pcl_units_upload");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TRY(pcl_pipe_begin(ctx, c_covariance, payload, n_chunks));
    TRY(pcl_pipe_finish(ctx, update_transitions));               // every chunk, in order: the collectives the other ranks' passes release
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

int pcl_mstep(pcl_ctx *ctx, double c_covariance) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_mstep: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return pcl_launch_mstep(ctx, c_covariance);
}

int pcl_model_conditioning(pcl_ctx *ctx, float *cond, float *cond_max) {
    if (!ctx) return PCL_ERR_INVALID;
    if (ctx->cond.empty()) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_model_conditioning: no model uploaded");
    if (cond) memcpy(cond, ctx->cond.data(), ctx->cond.size() * sizeof(float));
    if (cond_max) *cond_max = ctx->cond_max;
    return PCL_OK;
}

int pcl_score_occupancy(pcl_ctx *ctx, int workgroups_per_cu) {
    if (!ctx) return PCL_ERR_INVALID;
    if (workgroups_per_cu != 0 && workgroups_per_cu != 2) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_score_occupancy: 0 (default) or 2");
    ctx->score_wgs_per_cu = workgroups_per_cu;
    return PCL_OK;
}

int pcl_model_split_info(pcl_ctx *ctx, int *n_off, int *limit) {
    if (!ctx) return PCL_ERR_INVALID;
    if (ctx->nbad.empty()) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_model_split_info: no model uploaded");
    if (n_off) memcpy(n_off, ctx->nbad.data(), ctx->nbad.size() * sizeof(int));
    if (limit) *limit = ctx->split_max;
    return PCL_OK;
}

int pcl_model_download(pcl_ctx *ctx, double *mean, double *var, double *weight) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->mean64) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_model_download: no model uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->J * ctx->M * ctx->Dhost;
    double *tmp = nullptr;
    TRY(dev_alloc(ctx, &tmp, n));
    int r = PCL_OK;
    const double *srcs[3] = {ctx->mean64, ctx->var64, ctx->w64};
    double *dsts[3] = {mean, var, weight};
    for (int k = 0; k < 3 && r == PCL_OK; ++k) {
        if (!dsts[k]) continue;
        const int inner = (k == 2) ? 1 : ctx->Dhost;
        r = pcl_launch_pack(ctx, srcs[k], inner, tmp);
        if (r == PCL_OK && hipMemcpyAsync(dsts[k], tmp, (size_t)ctx->J * ctx->M * inner * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) r = PCL_ERR_HIP;
        if (r == PCL_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = PCL_ERR_HIP;
    }
    dev_free(tmp);
    if (r != PCL_OK && ctx->err.empty()) pcl_set_error(ctx, "pcl_model_download: copy failed");
    return r;
}

int pcl_accumulate_prune(pcl_ctx *ctx, double log2_threshold) {
    if (!ctx) return PCL_ERR_INVALID;
    if (log2_threshold > 0.0) PCL_FAIL(ctx, PCL_ERR_INVALID, "pcl_accumulate_prune: threshold 2^%g > 1", log2_threshold);
    ctx->acc_prune_log2 = log2_threshold;
    return PCL_OK;
}

int pcl_stats_download(pcl_ctx *ctx, double *acc, double *alpha_acc, double *mean_acc, double *cov_acc) {
    if (!ctx) return PCL_ERR_INVALID;
    if (!ctx->stats) PCL_FAIL(ctx, PCL_ERR_STATE, "pcl_stats_download: no model uploaded");
    HIPCHK(ctx, pcl_stats_join(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // [acc | alpha | mean | cov]: only as far as the caller asks (the two moment blocks are 99 % of the bytes)
    const size_t need = cov_acc ? ctx->stats_len : mean_acc ? (size_t)(ctx->st_cov - ctx->stats) : (size_t)(ctx->st_mean - ctx->stats);
    std::vector<double> h(need);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->stats, need * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int J = ctx->J, M = ctx->M, Mp = ctx->Mpad, D = ctx->Dhost, Dd = ctx->D;
    const double *a = h.data(), *al = a + (size_t)J * Mp, *me = al + J, *co = me + (size_t)J * Mp * Dd;
    for (int j = 0; j < J; ++j) {
        if (alpha_acc) alpha_acc[j] = al[j];
        for (int m = 0; m < M; ++m) {
            if (acc) acc[(size_t)j * M + m] = a[(size_t)j * Mp + m];
            for (int d = 0; d < D; ++d) {
                if (mean_acc) mean_acc[((size_t)j * M + m) * D + d] = me[((size_t)j * Mp + m) * Dd + d];
                if (cov_acc) cov_acc[((size_t)j * M + m) * D + d] = co[((size_t)j * Mp + m) * Dd + d];
            }
        }
    }
    return PCL_OK;
}

}  // extern "C"

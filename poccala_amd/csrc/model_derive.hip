// model_derive.hip -- device-side model preparation and the GMM M-step (gfx950).
//
//   derive: from the float64 master copy (mean, var, weight) build every layout the kernels read:
//           the VALU scoring rows [s_d c_d ... k2] (f32 + f64), the per-state expansion centres and the
//           MFMA layout [m-tile][KS4][64][4] of gmm_score_mfma.hip, and the f32 means.
//   mstep:  Clustering.GMM.update_param (StatisticalModel/Clustering.py:682-693) for every state at once
//           from the resident linear statistics:
//               w = acc / alpha_acc          (= exp(acc - alpha_acc) of the log-domain reference)
//               mu = mean_acc / acc - bias
//               var = max(cov_acc / acc, c_covariance)
//           so that an EM iteration (E-step -> RCCL all-reduce -> M-step -> next E-step) never leaves the GPU.
// Both are HBM-bound streaming passes over J*M*D elements (C4: 240 M elements, a few ms).
#include "pcl_internal.h"

namespace {

constexpr double LOG2E = 1.4426950408889634074, LOG_2PI = 1.8378770664093454836;

// one workgroup per state: centre c_j[d] = (float) mean_m mu[j,m,d]
__global__ void centers_kernel(const double *__restrict__ mean64, int M, int Mpad, int D, float *__restrict__ centers) {
    const int j = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += mean64[((size_t)j * Mpad + m) * D + d];
        centers[(size_t)j * D + d] = (float)(s / M);
    }
}

// one thread per (state, mixture) of the padded grids
__global__ void derive_kernel(const double *__restrict__ mean64, const double *__restrict__ var64,
                              const double *__restrict__ w64, const float *__restrict__ centers, int J, int M, int Mpad,
                              int Mpad32, int D, int Dhost, int row, int flags, float *__restrict__ params32,
                              double *__restrict__ params64, float *__restrict__ mean32, float *__restrict__ pm32) {
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int mmax = Mpad32 > Mpad ? Mpad32 : Mpad;
    if (gid >= (long long)J * mmax) return;
    const int j = (int)(gid / mmax), m = (int)(gid % mmax);
    const int KS = D + 1, KS4 = (KS + 3) / 4, nmt = Mpad32 / 32;
    const bool real_m = m < M;
    double k2 = -INFINITY;
    if (m < Mpad) {
        double *p64 = params64 + ((size_t)j * Mpad + m) * row;
        float *p32 = params32 + ((size_t)j * Mpad + m) * row;
        double sumvar = 0.0, sumlog = 0.0;
        for (int d = 0; d < D; ++d) {
            double s = 0.0, c = 0.0, mu = 0.0;
            if (real_m && d < Dhost) {
                const size_t o = ((size_t)j * Mpad + m) * D + d;
                mu = mean64[o];
                const double vr = var64[o];
                s = sqrt(LOG2E / (2.0 * vr));
                c = -mu * s;
                sumvar += vr;
                sumlog += log(vr);
            }
            p64[2 * d] = s; p64[2 * d + 1] = c;
            p32[2 * d] = (float)s; p32[2 * d + 1] = (float)c;
            mean32[((size_t)j * Mpad + m) * D + d] = (float)mu;
        }
        if (real_m) {
            // util.py:29 (quirk Q1): -D/2 ln 2pi - 1/2 sum(var); textbook log-determinant only on request
            const double tail = (flags & PCL_MODEL_LOGDET) ? sumlog : sumvar;
            k2 = LOG2E * (log(w64[(size_t)j * Mpad + m]) - 0.5 * Dhost * LOG_2PI - 0.5 * tail);
        }
        p64[2 * D] = k2; p32[2 * D] = (float)k2;
        for (int k = 2 * D + 1; k < row; ++k) { p64[k] = 0.0; p32[k] = 0.f; }
    }
    if (m < Mpad32 && pm32) {
        const int mt = m >> 5, cl = m & 31;
        float *base = pm32 + (((size_t)j * nmt + mt) * KS4) * 64 * 4;
        double kq = 0.0;
        for (int s = 0; s < KS4 * 4; ++s) {
            double a = 0.0, b = 0.0;
            if (real_m && s < Dhost) {
                const size_t o = ((size_t)j * Mpad + m) * D + s;
                const double vr = var64[o], dm = mean64[o] - (double)centers[(size_t)j * D + s];
                a = -LOG2E / (2.0 * vr);
                b = LOG2E * dm / vr;
                kq += dm * dm / (2.0 * vr);
            }
            base[((size_t)(s >> 2) * 64 + cl) * 4 + (s & 3)] = (float)a;
            base[((size_t)(s >> 2) * 64 + 32 + cl) * 4 + (s & 3)] = (float)b;
        }
        // the constant pair (k-step D): k' on the low half-wave, 1 in the spare slot (it multiplies 0 in the
        // scoring kernel, -ref / cf in the kernels that use the slot)
        base[((size_t)(D >> 2) * 64 + cl) * 4 + (D & 3)] = real_m ? (float)(k2 - LOG2E * kq) : -INFINITY;
        base[((size_t)(D >> 2) * 64 + 32 + cl) * 4 + (D & 3)] = 1.f;
    }
}

// Clustering.GMM.update_param for every (state, mixture)
__global__ void mstep_kernel(const double *__restrict__ st_acc, const double *__restrict__ st_alpha,
                             const double *__restrict__ st_mean, const double *__restrict__ st_cov, int J, int M, int Mpad,
                             int D, int Dhost, double bias, double floor_var, double *__restrict__ mean64,
                             double *__restrict__ var64, double *__restrict__ w64) {
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (gid >= (long long)J * M) return;
    const int j = (int)(gid / M), m = (int)(gid % M);
    const size_t jm = (size_t)j * Mpad + m;
    const double a = st_acc[jm];
    w64[jm] = a / st_alpha[j];                                   // Clustering.py:685
    for (int d = 0; d < Dhost; ++d) {
        mean64[jm * D + d] = st_mean[jm * D + d] / a - bias;     // :686
        double c = st_cov[jm * D + d] / a;                       // :688
        if (c < floor_var) c = floor_var;                        // :689-692
        var64[jm * D + d] = c;
    }
}

// gather the padded device master copy into the caller's dense (J,M,D) arrays
__global__ void pack_kernel(const double *__restrict__ src, int J, int M, int Mpad, int D, int Dhost, double *__restrict__ dst) {
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (gid >= (long long)J * M * Dhost) return;
    const int d = (int)(gid % Dhost);
    const long long jm = gid / Dhost;
    const int j = (int)(jm / M), m = (int)(jm % M);
    dst[gid] = src[((size_t)j * Mpad + m) * D + d];
}

// element-type conversion of the frame matrix: d64 -> f32 (dst32) or f32 -> d64 (dst64)
__global__ void cast_kernel(const double *__restrict__ src64, float *__restrict__ f32, double *__restrict__ dst64, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (dst64) dst64[i] = (double)f32[i];
        else f32[i] = (float)src64[i];
    }
}

}  // namespace

int pcl_launch_cast(pcl_ctx *ctx, const double *src64, float *f32, double *dst64, size_t n) {
    hipLaunchKernelGGL(cast_kernel, dim3(2048), dim3(256), 0, ctx->stream, src64, f32, dst64, n);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_derive(pcl_ctx *ctx) {
    hipLaunchKernelGGL(centers_kernel, dim3(ctx->J), dim3(64), 0, ctx->stream, ctx->mean64, ctx->M, ctx->Mpad, ctx->D, ctx->centers32);
    const int mmax = ctx->Mpad32 > ctx->Mpad ? ctx->Mpad32 : ctx->Mpad;
    const long long n = (long long)ctx->J * mmax;
    hipLaunchKernelGGL(derive_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->mean64, ctx->var64, ctx->w64,
                       ctx->centers32, ctx->J, ctx->M, ctx->Mpad, ctx->Mpad32, ctx->D, ctx->Dhost, ctx->row, ctx->model_flags,
                       ctx->params32, ctx->params64, ctx->mean32, ctx->pm32);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

int pcl_launch_mstep(pcl_ctx *ctx, double floor_var) {
    const long long n = (long long)ctx->J * ctx->M;
    pcl_timer_begin(ctx, "mstep");
    hipLaunchKernelGGL(mstep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->st_acc, ctx->st_alpha, ctx->st_mean,
                       ctx->st_cov, ctx->J, ctx->M, ctx->Mpad, ctx->D, ctx->Dhost, 100.0, floor_var, ctx->mean64, ctx->var64, ctx->w64);
    int r = pcl_launch_derive(ctx);
    pcl_timer_end(ctx, "mstep");
    HIPCHK(ctx, hipGetLastError());
    return r;
}

int pcl_launch_pack(pcl_ctx *ctx, const double *src, int inner, double *dst) {
    // inner = D for (J,M,D) arrays, 1 for (J,M)
    const long long n = (long long)ctx->J * ctx->M * (inner == 1 ? 1 : ctx->Dhost);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src, ctx->J, ctx->M, ctx->Mpad,
                       inner == 1 ? 1 : ctx->D, inner == 1 ? 1 : ctx->Dhost, dst);
    HIPCHK(ctx, hipGetLastError());
    return PCL_OK;
}

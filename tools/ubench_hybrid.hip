// ubench_hybrid.hip -- do the f32 matrix pipe and the f32 VALU add up on gfx950?
// A 512-thread workgroup puts two waves on every SIMD: waves 0-3 run a dependent v_mfma_f32_32x32x2_f32 chain,
// waves 4-7 run independent v_fma_f32 / v_pk_fma_f32 chains.  Each role is timed alone and together (s_memtime per
// wave, HIP events for the launch).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
#ifndef MFMA_BF16
#define MFMA_BF16 0
#endif

template <int VMODE>   // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_fma_f32 with one exp per 39 fma
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, float a, float b, int iters_m, int iters_v) {
    const int wave = threadIdx.x >> 6;
    float s = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        f16v acc = {0};
        float av = a + threadIdx.x * 1e-6f, bv = b;
#if MFMA_BF16
        bf8v a8, b8;
        for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(av + j); b8[j] = (__bf16)(bv * (j + 1)); }
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);   // 16 x 32 cyc = the same 512 cyc
        }
#else
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
#endif
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        if (VMODE == 1) {
            f2v acc[8], x = {a, b}, y = {b, a};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f2v{(float)i, threadIdx.x * 1e-3f};
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
        } else {
            float acc[8], e = threadIdx.x * 1e-3f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x * 1e-3f;
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (VMODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(e));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += acc[i];
            s += e;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VMODE>
void run(const char *name, int iters_m, int iters_v, float *out, unsigned long long *cyc) {
    const int nb = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<VMODE><<<nb, 512>>>(out, cyc, 0.5f, 0.25f, iters_m, iters_v);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<VMODE><<<nb, 512>>>(out, cyc, 0.5f, 0.25f, iters_m, iters_v);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[b * 8 + w] / (nb * 4);
    const double fm = (double)nb * 4 * iters_m * (MFMA_BF16 ? 16 * (2.0 * 32 * 32 * 16) : 8 * (2.0 * 32 * 32 * 2));
    const double fv = (double)nb * 4 * iters_v * (VMODE == 1 ? 16 * 128 * 2.0 : 32 * 64 * 2.0);
    printf("%-34s %8.3f ms  MFMA %6.1f TF  VALU %6.1f TF  sum %6.1f | memtime ticks/wave: mfma %.0f valu %.0f\n", name, ms,
           fm / (ms * 1e-3) / 1e12, fv / (ms * 1e-3) / 1e12, (fm + fv) / (ms * 1e-3) / 1e12, cm, cv);
}

int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
    const int M = 4096;           // 8 MFMAs x 64 cyc = 512 cyc per iteration
    run<0>("mfma alone", M, 0, out, cyc);
    run<0>("v_fma_f32 alone (1 wave/SIMD)", 0, 4 * M, out, cyc);
    run<1>("v_pk_fma_f32 alone (1 wave/SIMD)", 0, 4 * M, out, cyc);
    for (int q : {2, 4, 6, 8, 10}) {
        char nm[64];
        snprintf(nm, sizeof nm, "mfma + v_fma x%d/4", q); run<0>(nm, M, q * M, out, cyc);
    }
    for (int q : {2, 4, 6, 8}) {
        char nm[64];
        snprintf(nm, sizeof nm, "mfma + v_pk_fma x%d/4", q); run<1>(nm, M, q * M, out, cyc);
    }
    for (int q : {4, 8}) {
        char nm[64];
        snprintf(nm, sizeof nm, "mfma + v_fma+exp x%d/4", q); run<2>(nm, M, q * M, out, cyc);
    }
    return 0;
}

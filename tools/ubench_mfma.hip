// ubench_mfma.hip -- f32-input MFMA issue rate on gfx950, alone and with VALU/transcendental work
// interleaved (the shape of an MFMA-based GMM scoring loop: MFMA chain + per-output max/sub/exp/add).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    f16v acc0 = {0}, acc1 = {0};
    f4v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {            // 32x32x2, two independent accumulators
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc1, 0, 0, 0);
        } else if (MODE == 1) {     // 32x32x2, ONE dependent chain
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc0, 0, 0, 0);
        } else if (MODE == 2) {     // 16x16x4, four accumulators
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c3, 0, 0, 0);
        } else if (MODE == 3) {     // 32x32x2 chain + the LSE VALU work of 2 MFMAs' worth of outputs at K=80:
            // per 40 MFMAs (one 32x32 tile) 16 outputs/lane x ~6 VALU (incl. 1 exp) -> per MFMA 2.4 VALU; use 5 per 2 MFMAs
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc0, 0, 0, 0);
            asm volatile("v_max_f32 %0, %0, %1\n v_sub_f32 %2, %2, %0\n v_exp_f32 %3, %2\n v_add_f32 %4, %4, %3\n v_fma_f32 %5, %5, %1, %3"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    s += c0[0] + c1[1] + c2[2] + c3[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double flop_per_wave_iter, float *out) {
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, 0.5f, 0.25f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<grid, block>>>(out, 0.5f, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double waves = (double)grid.x * 4;
    printf("%-44s waves/SIMD=%d %.3f ms  %.1f TFLOP/s (MFMA flops)\n", name, waves_per_simd, ms, waves * ITERS * flop_per_wave_iter / (ms * 1e-3) / 1e12);
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 3}) {
        run<0>("mfma_f32_32x32x2f32, 2 accumulators", w, 2 * 32 * 32 * 2 * 2, out);
        run<1>("mfma_f32_32x32x2f32, 1 dependent chain", w, 2 * 32 * 32 * 2 * 2, out);
        run<2>("mfma_f32_16x16x4f32, 4 accumulators", w, 4 * 16 * 16 * 4 * 2, out);
        run<3>("32x32x2 chain + 5 VALU (1 exp) per 2 MFMA", w, 2 * 32 * 32 * 2 * 2, out);
    }
    return 0;
}

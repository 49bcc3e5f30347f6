// ubench_shape.hip -- 32x32x16 vs 16x16x32 f16 MFMA on RANDOM operands under the scoring kernel's mix
// (A fragments re-read from LDS, B fragments in registers, 32 v_exp_f32 + adds per 2048 outputs per wave).
// Same FLOPs, same outputs per iteration; the question is the clock the chip holds (power) and the LDS traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// per iteration ("m-tile"): 32 mixtures x 64 frames per wave, K = 256 f16 (incl. padding)
template <int SHAPE, int MINW>
__global__ __launch_bounds__(256, MINW) void k(const uint4 *__restrict__ gA, const uint4 *__restrict__ gB, float *out, int iters) {
    __shared__ uint4 lds[16 * 64];                       // 16 KB of A fragments
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = gA[i];
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    if (SHAPE == 32) {
        // B: 2 column tiles x 8 K-steps (K = 16 each... 128 per pass-equivalent): 16 MFMAs per column tile
        h8v b[2][16];
        for (int c = 0; c < 2; ++c) for (int s = 0; s < 16; ++s) b[c][s] = __builtin_bit_cast(h8v, gB[((c * 16 + s) * 64 + lane) & 4095]);
        for (int it = 0; it < iters; ++it) {
            f16v acc[2];
            for (int c = 0; c < 2; ++c) { acc[c] = f16v{0}; acc[c][0] = s0 * 1e-30f; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const h8v a = __builtin_bit_cast(h8v, lds[(s % 11) * 64 + lane]);      // 11 distinct KB re-read, as in the kernel
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[c][s], acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(acc[c][r] * 1e-3f);
#pragma unroll
                for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                    for (int r = 0; r < w; ++r) e[r] += e[r + w];
                if (c) s1 += e[0]; else s0 += e[0];
            }
        }
    } else {
        // 16x16x32: 2 mixture sub-tiles x 4 frame sub-tiles, 8 K-steps of 32
        h8v b[4][8];
        for (int c = 0; c < 4; ++c) for (int s = 0; s < 8; ++s) b[c][s] = __builtin_bit_cast(h8v, gB[((c * 8 + s) * 64 + lane) & 4095]);
        for (int it = 0; it < iters; ++it) {
            f4v acc[2][4];
            for (int m = 0; m < 2; ++m) for (int c = 0; c < 4; ++c) { acc[m][c] = f4v{0}; acc[m][c][0] = s0 * 1e-30f; }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const h8v a = __builtin_bit_cast(h8v, lds[(m * 8 + s) * 64 + lane]);   // 16 KB re-read
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[c][s], acc[m][c], 0, 0, 0);
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float e[8];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[m * 4 + r] = __builtin_amdgcn_exp2f(acc[m][c][r] * 1e-3f);
#pragma unroll
                for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                    for (int r = 0; r < w; ++r) e[r] += e[r + w];
                if (c & 1) s1 += e[0]; else s0 += e[0];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1;
}

template <int SHAPE, int MINW>
void run(const char *name, const uint4 *A, const uint4 *B, float *out) {
    const int iters = 2048;
    dim3 grid(256 * MINW * 4), block(256);            // 4 rounds of full occupancy: ~10+ ms, long enough for the clock to settle
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SHAPE, MINW><<<grid, block>>>(A, B, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k<SHAPE, MINW><<<grid, block>>>(A, B, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = (double)grid.x * 4 * iters * 32.0 * 64 * 256 * 2;
    printf("%-36s waves/SIMD=%d  %.2f ms  %.0f TFLOP/s f16 MFMA\n", name, MINW, ms, flop / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const bool zero = argc > 1;
    const size_t n = 4096;
    uint4 *hA = (uint4 *)malloc(n * 16), *hB = (uint4 *)malloc(n * 16);
    srand(3);
    auto rh = [&]() -> unsigned { if (zero) return 0; const float v = (rand() / (float)RAND_MAX) * 2 - 1; _Float16 h = (_Float16)v; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; };
    for (size_t i = 0; i < n; ++i) {
        hA[i] = make_uint4(rh() | rh() << 16, rh() | rh() << 16, rh() | rh() << 16, rh() | rh() << 16);
        hB[i] = make_uint4(rh() | rh() << 16, rh() | rh() << 16, rh() | rh() << 16, rh() | rh() << 16);
    }
    uint4 *A, *B; float *out;
    (void)hipMalloc(&A, n * 16); (void)hipMalloc(&B, n * 16); (void)hipMalloc(&out, 256 * 3 * 4 * 256 * 4);
    (void)hipMemcpy(A, hA, n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, n * 16, hipMemcpyHostToDevice);
    printf("%s operands\n", zero ? "all-zero" : "random");
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 2>("32x32x16, 16 KB... 11 KB A per tile", A, B, out);
        run<16, 2>("16x16x32, 16 KB A per tile", A, B, out);
        run<32, 3>("32x32x16", A, B, out);
        run<16, 3>("16x16x32", A, B, out);
    }
    return 0;
}

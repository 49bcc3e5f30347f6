// ubench_split.hip -- f32-accurate products on the bf16/f16 matrix pipe by error-free splitting (gfx950).
// (1) numerics: D[32x32] = A[32x80] B[80x32] with the magnitudes of the centred GMM exponent, computed by
//     the exact-f32 MFMA chain, by a 3-way bf16 split keeping 6 of the 9 cross terms, and by a 2-way f16 split
//     keeping 3 of 4, each against float64;
// (2) rate: the split chain with the log-sum-exp VALU work beside it.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
#define K 80

__device__ inline void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
    p1 = (__bf16)x; float r = x - (float)p1;
    p2 = (__bf16)r; r -= (float)p2;
    p3 = (__bf16)r;
}
__device__ inline void split2h(float x, _Float16 &p1, _Float16 &p2) {
    p1 = (_Float16)x; p2 = (_Float16)(x - (float)p1);
}

// A row-major [32][K], B row-major [K][32]; out[mode][32][32]
__global__ void numerics(const float *A, const float *B, float *out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    // mode 0: exact f32 chain
    f16v acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[0 * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    // mode 1: bf16 x3, six terms, small ones first
    bf8v a[3][K / 16], b[3][K / 16];
    for (int s = 0; s < K / 16; ++s)
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            __bf16 p1, p2, p3;
            split3(A[r * K + k], p1, p2, p3); a[0][s][j] = p1; a[1][s][j] = p2; a[2][s][j] = p3;
            split3(B[k * 32 + r], p1, p2, p3); b[0][s][j] = p1; b[1][s][j] = p2; b[2][s][j] = p3;
        }
    const int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};
    acc = f16v{0};
    for (int t = 0; t < 6; ++t)
        for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[t]][s], b[pb[t]][s], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[1 * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    // mode 2: same, large term first
    acc = f16v{0};
    for (int t = 5; t >= 0; --t)
        for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[t]][s], b[pb[t]][s], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[2 * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    // mode 3: bf16 x3, only 3 terms (a1b1, a1b2, a2b1): 16-bit accuracy, for scale
    acc = f16v{0};
    for (int t = 3; t < 6; ++t)
        for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[t]][s], b[pb[t]][s], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[3 * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    // mode 4: f16 x2, three terms
    h8v ah[2][K / 16], bh[2][K / 16];
    for (int s = 0; s < K / 16; ++s)
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            _Float16 p1, p2;
            split2h(A[r * K + k], p1, p2); ah[0][s][j] = p1; ah[1][s][j] = p2;
            split2h(B[k * 32 + r], p1, p2); bh[0][s][j] = p1; bh[1][s][j] = p2;
        }
    acc = f16v{0};
    for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][s], bh[0][s], acc, 0, 0, 0);
    for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][s], bh[1][s], acc, 0, 0, 0);
    for (int s = 0; s < K / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][s], bh[0][s], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[4 * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// rate: per "m-tile" NT x 30 bf16 MFMAs + the LSE of NT x 16 outputs per lane (exp2 + add)
template <int NT, int LSE, int NM>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    bf8v a[6], b[NT][6];
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(threadIdx.x * 1e-3f + i + j); for (int t = 0; t < NT; ++t) b[t][i][j] = (__bf16)(0.001f * (i + j + t)); }
    }
    float s[NT] = {0};
    for (int it = 0; it < iters; ++it) {
        f16v acc[NT];
        for (int t = 0; t < NT; ++t) { acc[t] = f16v{0}; acc[t][0] = s[t] * 1e-30f; }   // loop-carried: nothing can be hoisted
#pragma unroll
        for (int q = 0; q < NM; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % 6], b[t][(q + t) % 6], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (LSE) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[t] += __builtin_amdgcn_exp2f(acc[t][i]);
            } else {
                s[t] += acc[t][0] + acc[t][7];
            }
        }
    }
    float tot = 0;
    for (int t = 0; t < NT; ++t) tot += s[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}

template <int NT, int LSE, int NM = 30>
void run_rate(const char *name, int wps, float *out) {
    const int iters = 512;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<NT, LSE, NM><<<grid, block>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) rate<NT, LSE, NM><<<grid, block>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double tiles = (double)grid.x * 4 * iters * NT;            // 32x32 (mixture, frame) tiles
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / tiles;             // SIMD-cycles per tile at 2.4 GHz
    printf("%-40s waves/SIMD=%d %.3f ms  %.0f SIMD-cyc per 32x32 tile (f32 chain: 2560)  bf16 MFMA %.0f TF  = %.0f TF algorithmic\n", name, wps, ms,
           cyc, tiles * NM * 32768.0 / (ms * 1e-3) / 1e12, tiles * 1024 * 121.0 / (ms * 1e-3) / 1e12);
}

int main() {
    std::vector<float> A(32 * K), B(K * 32);
    std::vector<double> ref(1024);
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX) * 2 - 1; };
    // k = 2d: a = -log2e/(2 var), b = x'^2 ; k = 2d+1: a = log2e mu'/var, b = x'
    for (int f = 0; f < 32; ++f)
        for (int d = 0; d < 39; ++d) { const double x = 1.5 * rnd() * 1.7; B[(2 * d) * 32 + f] = (float)(x * x); B[(2 * d + 1) * 32 + f] = (float)x; }
    for (int f = 0; f < 32; ++f) { B[78 * 32 + f] = 1.f; B[79 * 32 + f] = (float)(-40 + 10 * rnd()); }
    for (int m = 0; m < 32; ++m) {
        for (int d = 0; d < 39; ++d) { const double var = 0.5 + 0.75 * (rnd() + 1), mu = 1.5 * rnd(); A[m * K + 2 * d] = (float)(-1.4427 / (2 * var)); A[m * K + 2 * d + 1] = (float)(1.4427 * mu / var); }
        A[m * K + 78] = (float)(-60 + 20 * rnd()); A[m * K + 79] = 1.f;
    }
    double mag = 0;
    for (int m = 0; m < 32; ++m) for (int f = 0; f < 32; ++f) { double s = 0, t = 0; for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * B[k * 32 + f]; t += fabs((double)A[m * K + k] * B[k * 32 + f]); } ref[m * 32 + f] = s; mag += t / 1024; }
    float *dA, *dB, *dO;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, 5 * 1024 * 4 + 256 * 4 * 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    numerics<<<1, 64>>>(dA, dB, dO);
    std::vector<float> o(5 * 1024);
    (void)hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
    const char *names[5] = {"exact f32 MFMA chain", "bf16 x3, 6 terms, small first", "bf16 x3, 6 terms, large first", "bf16 x3, 3 terms (16-bit)", "f16 x2, 3 terms"};
    printf("mean sum of |terms| = %.1f (log2 units)\n", mag);
    for (int md = 0; md < 5; ++md) {
        double e = 0, e2 = 0;
        for (int i = 0; i < 1024; ++i) { const double d = fabs(o[md * 1024 + i] - ref[i]); e = fmax(e, d); e2 += d * d / 1024; }
        printf("%-34s max |err| = %.3g   rms = %.3g  (log2 units)\n", names[md], e, sqrt(e2));
    }
    for (int w : {1, 2}) {
        run_rate<2, 0>("30 MFMA/tile, NT=2, no LSE", w, dO + 5 * 1024);
        run_rate<2, 1>("30 MFMA/tile, NT=2, + 16 exp2/add", w, dO + 5 * 1024);
        run_rate<4, 1>("30 MFMA/tile, NT=4, + 16 exp2/add", w, dO + 5 * 1024);
        run_rate<2, 0, 16>("16 MFMA/tile, NT=2, no LSE", w, dO + 5 * 1024);
        run_rate<2, 1, 16>("16 MFMA/tile, NT=2, + 16 exp2/add", w, dO + 5 * 1024);
        run_rate<4, 1, 16>("16 MFMA/tile, NT=4, + 16 exp2/add", w, dO + 5 * 1024);
    }
    return 0;
}

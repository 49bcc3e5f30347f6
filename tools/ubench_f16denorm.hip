// does v_mfma_f32_32x32x16_f16 honour subnormal f16 inputs on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__global__ void k(float *out, float small) {
    h8v a = {0}, b = {0};
    a[0] = (_Float16)1.0f;
    b[0] = (_Float16)small;          // subnormal in f16
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)b[0]; }
    // and as the A operand
    h8v a2 = {0}, b2 = {0};
    a2[0] = (_Float16)small; b2[0] = (_Float16)1.0f;
    f16v acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc2, 0, 0, 0);
    if (threadIdx.x == 0) out[2] = acc2[0];
    // product of two subnormals-free small numbers whose product is tiny in f32 (no issue expected)
    h8v a3 = {0}, b3 = {0};
    a3[0] = (_Float16)1e-4f; b3[0] = (_Float16)1e-4f;
    f16v acc3 = {0};
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b3, acc3, 0, 0, 0);
    if (threadIdx.x == 0) out[3] = acc3[0];
}
int main() {
    float *d, h[4];
    (void)hipMalloc(&d, 16);
    for (float s : {3e-6f, 6e-8f, 1e-5f}) {
        k<<<1, 64>>>(d, s);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("small=%g: f16(small)=%g  1*small via MFMA (B side)=%g (A side)=%g ; 1e-4*1e-4=%g\n", s, h[1], h[0], h[2], h[3]);
    }
    return 0;
}

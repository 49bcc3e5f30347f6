// ubench_valu.hip -- measures what bounds the scoring kernel on gfx950: f32 FMA issue rate by
// instruction form (v_fma_f32, v_pk_fma_f32 with VGPR operands, v_pk_fma_f32 with an SGPR-pair
// operand selected through op_sel) at 1/2/4 waves per SIMD, and v_exp_f32.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 4096

template <int MODE>
__global__ void k(float *out, float a, float b, unsigned long long *clk = nullptr) {
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    float acc[16];
    float2v acc2[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = float2v{acc[2 * i], acc[2 * i + 1]};
    float2v ab = {a, b};
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            float2v a2 = {a, a}, b2 = {b, b};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        } else if (MODE == 2) {
            // y.xy = y.xy * S.lo + S.hi  with S an SGPR pair (one constant-bus read)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(acc2[i]) : "s"(ab));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i]));
        } else if (MODE == 5) {
            // the scoring kernel's mix: y = x*S.lo + S.hi (sgpr pair) ; q = y*y + q, 8 independent chains
            float2v y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(y[i]) : "v"(acc2[i]), "s"(ab));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc2[i]) : "v"(y[i]));
        } else if (MODE == 6) {
            // same mix, parameters in a VGPR pair (LDS-broadcast form)
            float2v y[8];
            float2v abv = {a, b};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(y[i]) : "v"(acc2[i]), "v"(abv));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc2[i]) : "v"(y[i]));
        } else if (MODE == 7) {
            // same mix with plain v_fma_f32 (what the compiler emits for the LDS kernel)
            float y[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y[i]) : "v"(acc[i]), "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(y[i]));
        } else if (MODE == 4) {
            // v_fma_f32 with one SGPR operand
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(b));
        }
    }
    float s = 0;
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc2[i].x + acc2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double flop_per_lane_iter, float *out) {
    // one block of 256 threads = one wave per SIMD; blocks per CU = waves_per_simd
    int cus = 256;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 2 * 4096 * sizeof(unsigned long long));
    for (int r = 0; r < 5; ++r) k<MODE><<<grid, block>>>(out, 0.999f, 1e-3f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double lanes = (double)grid.x * 256;
    double tf = lanes * ITERS * flop_per_lane_iter / (ms * 1e-3) / 1e12;
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memrealtime ticks at 100 MHz
    int ninstr = (MODE == 1 || MODE == 2) ? 8 : (MODE == 5 || MODE == 6) ? 16 : (MODE == 7) ? 32 : 16;
    double instr_per_clk_simd = (double)ITERS * ninstr * waves_per_simd / (ms * 1e-3 * ghz * 1e9);
    printf("%-36s waves/SIMD=%d  %.3f ms  %6.1f TFLOP/s  clock %.2f GHz  wave-instr/clk/SIMD=%.3f\n", name, waves_per_simd, ms, tf,
           ghz, instr_per_clk_simd);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
    // correctness of MODE 2 (op_sel on an SGPR pair)
    {
        k<2><<<1, 64>>>(out, 0.5f, 0.25f);
        k<1><<<1, 64>>>(out + 64, 0.5f, 0.25f);
        hipDeviceSynchronize();
        std::vector<float> h(128);
        hipMemcpy(h.data(), out, 128 * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; ++i) bad += (h[i] != h[64 + i]);
        printf("pk_fma sgpr-pair op_sel form vs vgpr form: %s (%g vs %g)\n", bad ? "MISMATCH" : "identical", h[3], h[67]);
    }
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 (vgpr,vgpr)", w, 16 * 2, out);
        run<4>("v_fma_f32 (sgpr,vgpr)", w, 16 * 2, out);
        run<1>("v_pk_fma_f32 (vgpr pairs)", w, 8 * 4, out);
        run<2>("v_pk_fma_f32 (sgpr pair, op_sel)", w, 8 * 4, out);
        run<3>("v_exp_f32", w, 16, out);
        run<5>("mix pk: y=x*S+S (sgpr) ; q+=y*y", w, 16 * 4, out);
        run<6>("mix pk: y=x*P+P (vgpr) ; q+=y*y", w, 16 * 4, out);
        run<7>("mix v_fma: y=x*s+c ; q+=y*y", w, 32 * 2, out);
    }
    return 0;
}

// ubench_trread.hip -- checks, with exact integer data, the LDS image and the transposed reads the f16 x2 accumulate kernel
// (gmm_accumulate_f16.hip, product 2) relies on:
//   image   per k-step s a 1-KiB block [side 2][frame 32][8 f16] (what product 1 reads with ds_read_b128, lane = side*32+frame),
//           its 64-byte units (4 frames) rotated inside each side by (side + 2 (s & 1)) so that the transposed reads of one
//           32-lane half fall on four different bank quarters;
//   reads   ds_read_b64_tr_b16: per 16-lane group a block of 4 rows (frames) x 16 columns, lane 4q+p supplies the address of
//           row q, columns 4p..4p+3; lane i receives column i of the 4 rows;
//   product S^T[c][m] = sum_f X[f][c] g[f][m] with A = X^T from the transposed reads and B = g taken straight from a 32x32
//           accumulator tile (column m on the lane, frames in the registers, k order permuted).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_trread.hip -o tools/ubench_trread.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int KS = 5, NCT = 3;

__host__ __device__ inline int img_off(int s, int side, int frame) {      // byte offset of the 16 B of (k-step, side, frame)
    const int unit = ((frame >> 2) + side + 2 * (s & 1)) & 7;
    return s * 1024 + side * 512 + unit * 64 + (frame & 3) * 16;
}

__global__ void k(const unsigned char *img, const float *g, float *out, float *out1) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[KS * 1024];
    for (int i = threadIdx.x; i < KS * 1024 / 16; i += 64) reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(img)[i];
    __syncthreads();
    const int lane = threadIdx.x, h = lane >> 5, col = lane & 31;
    // g in accumulator layout: register r <-> frame (r&3) + 8 (r>>2) + 4h, column = mixture
    f16v d;
    for (int r = 0; r < 16; ++r) d[r] = g[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + col];
    h8v gb[2];
    for (int sp = 0; sp < 2; ++sp)
        for (int j = 0; j < 8; ++j) gb[sp][j] = (_Float16)d[8 * sp + j];
    // product 1 style row reads (ds_read_b128): lane = side*32 + frame -> 8 features of k-step s; summed for a check
    float rowsum = 0.f;
    for (int s = 0; s < KS; ++s) {
        const h8v a = *reinterpret_cast<const h8v *>(lds + img_off(s, h, col));
        for (int j = 0; j < 8; ++j) rowsum += (float)a[j] * (float)(1 + j + 8 * s);
    }
    out1[lane] = rowsum;
    // transposed reads: group = lane >> 4; within it q = (lane >> 2) & 3 (row), p = lane & 3 (chunk)
    const int grp = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    for (int ct = 0; ct < NCT; ++ct) {
        f16v acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int s = 2 * ct + grp;
        if (s >= KS) s = KS - 1;
        for (int sp = 0; sp < 2; ++sp) {
            h8v a;
            for (int rd = 0; rd < 2; ++rd) {
                const int f0 = 16 * sp + 4 * h + 8 * rd;
                const int off = img_off(s, p >> 1, f0 + q) + 8 * (p & 1);
                const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + off));
                const h4v hv = __builtin_bit_cast(h4v, v);
                for (int j = 0; j < 4; ++j) a[4 * rd + j] = hv[j];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb[sp], acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) out[(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + col] = acc[r];
    }
}

int main() {
    std::vector<_Float16> X(32 * 80);           // X[f][c], c = side*40 + d
    std::vector<float> g(32 * 32);
    srand(1);
    for (auto &v : X) v = (_Float16)(float)(rand() % 17 - 8);
    for (auto &v : g) v = (float)(rand() % 9 - 4);
    std::vector<unsigned char> img(KS * 1024);
    for (int s = 0; s < KS; ++s)
        for (int side = 0; side < 2; ++side)
            for (int f = 0; f < 32; ++f)
                for (int j = 0; j < 8; ++j) reinterpret_cast<_Float16 *>(img.data() + img_off(s, side, f))[j] = X[f * 80 + side * 40 + 8 * s + j];
    unsigned char *dimg; float *dg, *dout, *dout1;
    hipMalloc(&dimg, img.size()); hipMalloc(&dg, g.size() * 4); hipMalloc(&dout, 96 * 32 * 4); hipMalloc(&dout1, 64 * 4);
    hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dimg, dg, dout, dout1);
    std::vector<float> out(96 * 32), out1(64);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(out1.data(), dout1, 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0, bad1 = 0;
    for (int lane = 0; lane < 64; ++lane) {
        float ref = 0.f;
        for (int s = 0; s < KS; ++s) for (int j = 0; j < 8; ++j) ref += (float)X[(lane & 31) * 80 + (lane >> 5) * 40 + 8 * s + j] * (float)(1 + j + 8 * s);
        if (ref != out1[lane]) ++bad1;
    }
    // row c of column tile ct: group = c >> 4 -> k-step s = 2 ct + group; i = c & 15: side = i >> 3, d = 8 s + (i & 7)
    for (int ct = 0; ct < NCT; ++ct)
        for (int c = 0; c < 32; ++c) {
            int s = 2 * ct + (c >> 4);
            if (s >= KS) s = KS - 1;
            const int i = c & 15, xc = (i >> 3) * 40 + 8 * s + (i & 7);
            for (int m = 0; m < 32; ++m) {
                float ref = 0.f;
                for (int f = 0; f < 32; ++f) ref += (float)X[f * 80 + xc] * g[f * 32 + m];
                if (ref != out[(ct * 32 + c) * 32 + m]) {
                    if (bad < 5) printf("mismatch ct %d c %d m %d: got %g want %g\n", ct, c, m, out[(ct * 32 + c) * 32 + m], ref);
                    ++bad;
                }
            }
        }
    printf("row reads: %d mismatches of 64; transposed product: %d mismatches of %d\n", bad1, bad, NCT * 32 * 32);
    return bad || bad1;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* y, double* z, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = ::exp2(x[i]); z[i] = exp(x[i] * 0.693147180559945309417232121458); }
}
int main() {
    const int n = 4096;
    double hx[n], hy[n], hz[n];
    for (int i = 0; i < n; ++i) hx[i] = -80.0 * (double)rand() / RAND_MAX;
    double *dx, *dy, *dz;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dz, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, dz, n);
    hipMemcpy(hy, dy, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hz, dz, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) { e1 = fmax(e1, fabs(hy[i] / exp2(hx[i]) - 1)); e2 = fmax(e2, fabs(hz[i] / exp2(hx[i]) - 1)); }
    printf("device exp2(double) max rel err %.3g ; exp(x ln2) max rel err %.3g\n", e1, e2);
    return 0;
}

// Calibration of the TCC FETCH_SIZE / WRITE_SIZE counters for 8-byte-per-lane (fp64) streams on
// gfx950 (the guide calibrates 16-byte-per-lane only).  Run under rocprofv3 --pmc FETCH_SIZE /
// --pmc WRITE_SIZE; each kernel moves exactly 512 MiB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read8(const double *in, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 1.2345) out[0] = s;
}
__global__ void k_read16(const double2 *in, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = in[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
__global__ void k_write8(double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (double)i;
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 8;
    double *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, 4096);
    hipMemset(a, 0, bytes);
    hipLaunchKernelGGL(k_read8, 2048, 256, 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read16, 2048, 256, 0, 0, (const double2 *)a, b, n / 2);
    hipLaunchKernelGGL(k_write8, 2048, 256, 0, 0, a, n);
    hipDeviceSynchronize();
    printf("moved %zu bytes per kernel\n", bytes);
    return 0;
}

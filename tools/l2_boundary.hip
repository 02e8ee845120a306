// Does an XCD's L2 keep what a kernel wrote across a (same-stream) kernel boundary?  Producer launch: workgroup b writes
// chunk b (CHUNK bytes).  Consumer launch: workgroup b reads chunk (b + shift) % grid with every load issued before the
// first use, and stamps the cycles of that read.  Workgroups go to the XCDs round-robin by id, so shift % 8 == 0 reads what
// the SAME XCD wrote in the launch before, any other shift what another XCD wrote.  If the L2 survived the boundary the
// first case reads at L2 speed (guide: ~29 B/cycle/CU) and the second at fabric / Infinity-Cache speed (10-14 B/cycle/CU).
// Decides whether XCD-stable placement of producer and consumer blocks can help the fused PCR steps.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_boundary.hip -o tools/l2_boundary && tools/l2_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int CHUNK_DOUBLES = 20480;      // 160 KB: what a fused-step workgroup loads
constexpr int NT = 512, PER = CHUNK_DOUBLES / NT / 2;     // double2 per lane

__global__ __launch_bounds__(NT) void k_produce(double *buf, double v) {
    double2 *p = reinterpret_cast<double2 *>(buf + (size_t)blockIdx.x * CHUNK_DOUBLES);
    for (int i = 0; i < PER; ++i) p[i * NT + threadIdx.x] = make_double2(v + i, v - i);
}
__global__ __launch_bounds__(NT) void k_consume(const double *buf, int shift, unsigned long long *cyc, int *xcc, double *sink) {
    const int src = (blockIdx.x + shift) % gridDim.x;
    const double2 *p = reinterpret_cast<const double2 *>(buf + (size_t)src * CHUNK_DOUBLES);
    double2 v[PER];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = p[i * NT + threadIdx.x];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i].x + v[i].y;
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (s == 1234.5678) sink[0] = s;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[blockIdx.x] = (int)(x & 0xf);
    }
}
int main() {
    const int grid = 252;
    double *buf, *sink; unsigned long long *cyc; int *xcc;
    CHK(hipMalloc(&buf, (size_t)grid * CHUNK_DOUBLES * 8)); CHK(hipMalloc(&sink, 8));
    CHK(hipMalloc(&cyc, grid * 8)); CHK(hipMalloc(&xcc, grid * 4));
    std::vector<unsigned long long> h(grid); std::vector<int> hx(grid), hx0(grid);
    for (int shift : {0, 8, 16, 1, 3, 4, 0}) {
        double best = 1e30, med = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k_produce, dim3(grid), dim3(NT), 0, 0, buf, (double)rep);
            hipLaunchKernelGGL(k_consume, dim3(grid), dim3(NT), 0, 0, buf, shift, cyc, xcc, sink);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost));
            if (rep == 0 && shift == 0) hx0 = hx;
            std::sort(h.begin(), h.end());
            best = std::min(best, (double)h[grid / 2]);
            med = (double)h[grid / 2];
        }
        int same = 0;
        for (int b = 0; b < grid; ++b) same += hx[(b + shift) % grid] == hx[b];
        printf("shift %2d: median read phase %.0f cycles (best of 5 medians; last %.0f) = %.1f B/cycle/CU for %d KB; producer on the consumer's XCD for %d of %d workgroups; xcc of wg 0..9:",
               shift, best, med, CHUNK_DOUBLES * 8.0 / best, CHUNK_DOUBLES * 8 / 1024, same, grid);
        for (int b = 0; b < 10; ++b) printf(" %d", hx[b]);
        printf("\n");
    }
    // the same read twice in ONE launch-pair without a producer in between: the second consumer launch re-reads what the first read
    for (int shift : {0, 1}) {
        hipLaunchKernelGGL(k_produce, dim3(grid), dim3(NT), 0, 0, buf, 1.0);
        hipLaunchKernelGGL(k_consume, dim3(grid), dim3(NT), 0, 0, buf, 0, cyc, xcc, sink);
        hipLaunchKernelGGL(k_consume, dim3(grid), dim3(NT), 0, 0, buf, shift, cyc, xcc, sink);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("re-read after a read-only launch, shift %d: median %llu cycles = %.1f B/cycle/CU\n", shift, h[grid / 2], CHUNK_DOUBLES * 8.0 / (double)h[grid / 2]);
    }
    return 0;
}

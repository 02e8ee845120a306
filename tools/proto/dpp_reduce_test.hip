// Stand-alone check + timing of a DPP wave reduction for doubles against the __shfl_down tree (ssba_device.h: wave_sum).
//   hipcc --offload-arch=gfx950 -O3 tools/proto/dpp_reduce_test.hip -o /tmp/dpp_test && /tmp/dpp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

static __device__ __forceinline__ double wave_sum_shfl(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;       // lane 0
}
template <int CTRL, int ROW_MASK> static __device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0x111, 0xf>(v);       // row_shr:1
    v += dpp_move<0x112, 0xf>(v);       // row_shr:2
    v += dpp_move<0x114, 0xf>(v);       // row_shr:4
    v += dpp_move<0x118, 0xf>(v);       // row_shr:8   -> lane 15 of every row of 16 holds the row's sum
    v += dpp_move<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;       // lane 63
}
__global__ void k_check(const double *in, double *a, double *b) {
    const double v = in[blockIdx.x * 64 + threadIdx.x];
    const double s = wave_sum_shfl(v), t = wave_sum_dpp(v);
    if (threadIdx.x == 0) a[blockIdx.x] = s;
    if (threadIdx.x == 63) b[blockIdx.x] = t;
}
template <int DPP> __global__ void k_time(const double *in, double *out, int reps) {
    double v = in[threadIdx.x], acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = DPP ? wave_sum_dpp(v + q + r) : wave_sum_shfl(v + q + r);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += x[q];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
    const int nb = 1024;
    std::vector<double> h(nb * 64);
    for (auto &x : h) x = (double)rand() / RAND_MAX - 0.5;
    double *din, *da, *db, *dout;
    hipMalloc(&din, h.size() * 8); hipMalloc(&da, nb * 8); hipMalloc(&db, nb * 8); hipMalloc(&dout, h.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    k_check<<<nb, 64>>>(din, da, db);
    std::vector<double> a(nb), b(nb);
    hipMemcpy(a.data(), da, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, nb * 8, hipMemcpyDeviceToHost);
    double worst = 0, exact_worst = 0;
    for (int i = 0; i < nb; ++i) {
        long double s = 0;
        for (int j = 0; j < 64; ++j) s += h[i * 64 + j];
        worst = fmax(worst, fabs(a[i] - b[i]));
        exact_worst = fmax(exact_worst, fabs(b[i] - (double)s));
    }
    printf("max |shfl - dpp| = %.3e   max |dpp - exact| = %.3e\n", worst, exact_worst);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dpp = 0; dpp < 2; ++dpp) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (dpp) k_time<1><<<1024, 64>>>(din, dout, 1000); else k_time<0><<<1024, 64>>>(din, dout, 1000);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.3f ms for 1024 waves x 8000 reductions\n", dpp ? "dpp " : "shfl", ms);
    }
    return 0;
}

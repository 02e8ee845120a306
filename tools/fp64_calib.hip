// Calibration microbenchmarks for fp64 on gfx950 (not part of the product path):
// FMA throughput at 1/2/4 waves per SIMD, dependent-chain latency, rsq/rcp/sqrt/div latency,
// LDS b128 read rate, trivial-kernel duration.   hipcc --offload-arch=gfx950 -O3 fp64_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(double *o) { if (o == nullptr) o[0] = 1; }

template <int NACC>
__global__ void k_fma_tp(double *out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16); per lane 1 double of A, 1 of B, 4 of C/D; 2048 flop per instruction
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma_tp(double *out, int iters, double a, double b) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{threadIdx.x * 1e-9, 1.0 * i, 0.0, 0.0};
    const double av = a + threadIdx.x * 1e-12, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// do the fp64 MFMA pipe and the fp64 VALU overlap?  Two waves per SIMD: even waves issue MFMAs, odd waves FMAs.
__global__ void k_mixed(double *out, int iters_mfma, int iters_fma, double a, double b) {
    const int wv = threadIdx.x >> 6;
    double s = 0;
    if ((wv >> 2) == 0) {       // waves 0..3: one per SIMD
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = d4{threadIdx.x * 1e-9, 1.0 * i, 0.0, 0.0};
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {                    // waves 4..7: the second wave of every SIMD
        double acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
        for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave: cycles per dependent op
template <int OP>
__global__ void k_chain(double *out, long long *cyc, int iters, double a, double b) {
    double x = 1.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) x = fma(x, a, b);
        if (OP == 1) x = __builtin_amdgcn_rsq(x) + b;
        if (OP == 2) x = __builtin_amdgcn_rcp(x) + b;
        if (OP == 3) x = sqrt(x) + b;
        if (OP == 4) x = b / x + a;
        if (OP == 5) x = __shfl(x, (threadIdx.x + 1) & 63, 64) + b;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_lds_read(double *out, int iters) {
    __shared__ double s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i;
    __syncthreads();
    double acc0 = 0, acc1 = 0;
    const double2 *s2 = reinterpret_cast<const double2 *>(s);
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double2 v = s2[(idx + j * 256) & 4095];
            acc0 += v.x; acc1 += v.y;
        }
        idx = (idx + 1) & 4095;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1;
}

// STREAM-style copy: 16 bytes per lane, read + write; bytes moved = 2 x n
__global__ void k_copy(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// the same with eight 16-byte loads in flight per lane before the first store (what the guide's 6.3 TB/s float4 copy needs:
// a single load per lane and trip leaves the memory system idle between trips)
typedef double vd2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy8(const double2 *__restrict__ src_, double2 *__restrict__ dst_, size_t n) {
    const vd2 *src = reinterpret_cast<const vd2 *>(src_);
    vd2 *dst = reinterpret_cast<vd2 *>(dst_);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 7 * stride < n; i += 8 * stride) {
        vd2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(&src[i + q * stride]);
#pragma unroll
        for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], &dst[i + q * stride]);
    }
}

template <class F> float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    double *out; long long *cyc;
    CHK(hipMalloc(&out, 1 << 26)); CHK(hipMalloc(&cyc, 64));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    printf("empty kernel (back-to-back, event avg): %.2f us\n", 1e3 * time_ms([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, 0, out); }, 200));
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        int threads = 256 * wps, blocks = 256;   // one block per CU, wps waves per SIMD
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma_tp<8>, blocks, threads, 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
        double fl = 2.0 * 8 * iters * (double)threads * blocks;
        printf("fp64 FMA throughput, %d wave(s)/SIMD, 8 indep acc: %.1f TFLOP/s (%.3f ms)\n", wps, fl / ms / 1e9, ms);
    }
    for (int wps : {1, 2}) {
        int threads = 256 * wps, blocks = 256;
        const int it2 = 4000;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_tp<4>, blocks, threads, 0, 0, out, it2, 1.0000001, 1e-9); }, 3);
        double fl = 2048.0 * 4 * it2 * (double)(threads / 64) * blocks;
        printf("fp64 MFMA 16x16x4 throughput, %d wave(s)/SIMD, 4 indep acc: %.1f TFLOP/s (%.3f ms)\n", wps, fl / ms / 1e9, ms);
    }
    {
        // alone: 4000 x 4 MFMAs take t_m, 26000 x 8 FMAs take t_f (one wave per SIMD each); together: max() if the pipes overlap, sum if shared
        float tm = time_ms([&] { hipLaunchKernelGGL(k_mixed, 256, 512, 0, 0, out, 4000, 0, 1.0000001, 1e-9); }, 3);
        float tf = time_ms([&] { hipLaunchKernelGGL(k_mixed, 256, 512, 0, 0, out, 0, 13000, 1.0000001, 1e-9); }, 3);
        float tb = time_ms([&] { hipLaunchKernelGGL(k_mixed, 256, 512, 0, 0, out, 4000, 13000, 1.0000001, 1e-9); }, 3);
        printf("fp64 MFMA wave + fp64 FMA wave on every SIMD: MFMA alone %.3f ms, FMA alone %.3f ms, together %.3f ms (sum %.3f)\n", tm, tf, tb, tm + tf);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_tp<4>, 1, 64, 0, 0, out, 4000, 1.0000001, 1e-9); }, 3);
        printf("one wave, 4 indep MFMA 16x16x4 chains: %.2f ns per MFMA (%.1f flop/ns)\n", ms * 1e6 / (4.0 * 4000), 2048.0 / (ms * 1e6 / (4.0 * 4000)));
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_tp<1>, 1, 64, 0, 0, out, 4000, 1.0000001, 1e-9); }, 3);
        printf("one wave, 1 dependent MFMA 16x16x4 chain: %.2f ns per MFMA\n", ms * 1e6 / 4000.0);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma_tp<8>, 1, 64, 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
        printf("one wave, 8 indep FMA chains: %.2f ns per wave-FMA\n", ms * 1e6 / (8.0 * iters));
        ms = time_ms([&] { hipLaunchKernelGGL(k_fma_tp<1>, 1, 64, 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
        printf("one wave, 1 dependent FMA chain: %.2f ns per FMA\n", ms * 1e6 / (1.0 * iters));
    }
    const char *names[] = {"fma", "rsq+add", "rcp+add", "sqrt+add", "div+add", "shfl+add"};
    long long h;
    auto run_chain = [&](int op) {
        switch (op) {
            case 0: hipLaunchKernelGGL(k_chain<0>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1e-9); break;
            case 1: hipLaunchKernelGGL(k_chain<1>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1.5); break;
            case 2: hipLaunchKernelGGL(k_chain<2>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1.5); break;
            case 3: hipLaunchKernelGGL(k_chain<3>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1.5); break;
            case 4: hipLaunchKernelGGL(k_chain<4>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1.5); break;
            case 5: hipLaunchKernelGGL(k_chain<5>, 1, 64, 0, 0, out, cyc, 4000, 1.0000001, 1.5); break;
        }
    };
    for (int op = 0; op < 6; ++op) {
        float ms = time_ms([&] { run_chain(op); }, 3);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("dependent %-9s: %.1f clock64 ticks/op, %.1f ns/op\n", names[op], (double)h / 4000, ms * 1e6 / 4000);
    }
    {
        // HBM: 1 GiB read + 1 GiB written per launch (far beyond the 256 MiB Infinity Cache)
        const size_t n = (size_t)1 << 26;      // double2 elements
        double2 *a, *b;
        CHK(hipMalloc(&a, n * 16)); CHK(hipMalloc(&b, n * 16));
        CHK(hipMemset(a, 0, n * 16));
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, 256 * 16, 256, 0, 0, a, b, n); }, 5);
        printf("HBM copy (1 GiB read + 1 GiB write per launch): %.0f GB/s\n", 2.0 * n * 16 / ms / 1e6);
        for (int wgs : {4, 8, 16}) {
            ms = time_ms([&] { hipLaunchKernelGGL(k_copy8, 256 * wgs, 256, 0, 0, a, b, n); }, 5);
            printf("HBM copy, 8 x 16 B in flight per lane, nontemporal, %d workgroups per CU: %.0f GB/s\n", wgs, 2.0 * n * 16 / ms / 1e6);
        }
        hipFree(a); hipFree(b);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_read, 256, 256, 0, 0, out, 4000); }, 3);
        double bytes = 256.0 * 256 * 4000 * 8 * 16;
        printf("LDS ds_read_b128: %.1f TB/s chip (%.1f B/clk/CU at 2.4 GHz)\n", bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9);
    }
    return 0;
}

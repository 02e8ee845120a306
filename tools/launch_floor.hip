// What does a dependent launch cost inside a hipGraph on this chip, and what part of it is the 3.6 KB Dev struct that every
// solver kernel takes BY VALUE?  Chains of N dependent single-wave kernels, captured once and replayed:
//   small      : (int *state)                                 -- 8 bytes of kernel arguments
//   byvalue    : (Big d, int *state), reads d.tail            -- 3.6 KB of kernel arguments, last word used
//   byvalue0   : (Big d, int *state), reads d.head            -- 3.6 KB passed, first word used
//   bypointer  : (const Big *d, int *state), reads d->tail    -- the struct lives in device memory
// each also as "+state": the kernel first reads a word the PREVIOUS kernel wrote (the solver's `terminated` test) and writes it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { long head; char pad[3640 - 16]; long tail; };
static_assert(sizeof(Big) == 3640, "same size as ssba::Dev");

__global__ void k_small(int *state, int use_state) {
    if (use_state) { if (threadIdx.x == 0) state[0] = state[0] + 1; }
}
__global__ void k_byvalue(Big d, int *state, int use_state) {
    if (d.tail == 12345) state[1] = 1;
    if (use_state) { if (threadIdx.x == 0) state[0] = state[0] + 1; }
}
__global__ void k_byvalue0(Big d, int *state, int use_state) {
    if (d.head == 12345) state[1] = 1;
    if (use_state) { if (threadIdx.x == 0) state[0] = state[0] + 1; }
}
__global__ void k_bypointer(const Big *d, int *state, int use_state) {
    if (d->tail == 12345) state[1] = 1;
    if (use_state) { if (threadIdx.x == 0) state[0] = state[0] + 1; }
}

// clock64() against wall time: spins until the counter has advanced by `cycles`
__global__ void k_spin(long long cycles, long long *out) {
    const long long t0 = clock64(), w0 = wall_clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) { out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
}

template <class F>
static double time_chain(hipStream_t s, int n, int reps, F enqueue) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return us / ((double)reps * n);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int grid = argc > 3 ? atoi(argv[3]) : 1;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *state;
    CK(hipMalloc(&state, 256));
    CK(hipMemset(state, 0, 256));
    Big h{};
    Big *dbig;
    CK(hipMalloc(&dbig, sizeof(Big)));
    CK(hipMemcpy(dbig, &h, sizeof(Big), hipMemcpyHostToDevice));
    {
        long long *out;
        CK(hipMalloc(&out, 16));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 2000000LL, out);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            long long h[2];
            CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            printf("clock64: %lld ticks in %.1f us by events (%.1f per us); wall_clock64: %lld ticks (%.1f per us)\n", h[0], 1000.0 * ms, h[0] / (1000.0 * ms), h[1], h[1] / (1000.0 * ms));
        }
    }
    printf("chain of %d dependent launches per graph, %d replays, grid %d x 64 lanes; microseconds per launch\n", n, reps, grid);
    for (int use_state = 0; use_state < 2; ++use_state) {
        const char *sfx = use_state ? "+state" : "";
        printf("  small%-7s  %6.2f\n", sfx, time_chain(s, n, reps, [&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(64), 0, s, state, use_state); }));
        printf("  byvalue%-7s%6.2f\n", sfx, time_chain(s, n, reps, [&] { hipLaunchKernelGGL(k_byvalue, dim3(grid), dim3(64), 0, s, h, state, use_state); }));
        printf("  byvalue0%-6s%6.2f\n", sfx, time_chain(s, n, reps, [&] { hipLaunchKernelGGL(k_byvalue0, dim3(grid), dim3(64), 0, s, h, state, use_state); }));
        printf("  bypointer%-5s%6.2f\n", sfx, time_chain(s, n, reps, [&] { hipLaunchKernelGGL(k_bypointer, dim3(grid), dim3(64), 0, s, dbig, state, use_state); }));
    }
    return 0;
}

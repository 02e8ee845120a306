// Stand-alone check + timing of the 144-row parallel cyclic reduction (ceres_slam_amd/csrc/ssba_wide.hip): a random SPD
// block-tridiagonal system is pushed through launch_wide_solve and compared with a dense host Cholesky; the first factor
// launch is checked on its own (Y = U^-T [L | U | r] against a host triangular solve).  Not part of the product path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ceres_slam_amd/csrc tools/wide_bench.hip ceres_slam_amd/csrc/ssba_pool.hip -o tools/wide_bench && tools/wide_bench [n]
#include "../ceres_slam_amd/csrc/ssba_wide.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace ssba;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <class T> static T *dalloc(size_t n) { T *p; CHK(hipMalloc(&p, n * sizeof(T))); CHK(hipMemset(p, 0, n * sizeof(T))); return p; }
static double maxrel(const double *a, const double *b, size_t n) {
    double e = 0, m = 0;
    for (size_t i = 0; i < n; ++i) { e = fmax(e, fabs(a[i] - b[i])); m = fmax(m, fabs(b[i])); }
    return e / fmax(m, 1e-300);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 25;
    const int B = WBD;
    const size_t blk = (size_t)B * B;
    const int N = n * B;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> hD(n * blk), hL(n * blk, 0.0), hr((size_t)N);
    for (int b = 0; b < n; ++b) {
        std::vector<double> A((size_t)B * 2 * B);
        for (auto &v : A) v = nd(rng);
        for (int i = 0; i < B; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0;
                for (int k = 0; k < 2 * B; ++k) s += A[(size_t)i * 2 * B + k] * A[(size_t)j * 2 * B + k];
                s = s / (2 * B) + (i == j ? 1.0 : 0.0);
                hD[b * blk + (size_t)i * B + j] = hD[b * blk + (size_t)j * B + i] = s;
            }
        if (b > 0) for (size_t i = 0; i < blk; ++i) hL[b * blk + i] = 0.03 * nd(rng);
    }
    for (auto &v : hr) v = nd(rng);

    Dev d;
    memset(&d, 0, sizeof d);
    d.st = dalloc<State>(1);
    d.nfree = n * WSP;
    d.x0 = dalloc<double>((size_t)N);
    d.dbg = dalloc<unsigned long long>(256);        // (-DWD_STAMPS: every launch of k_wd_factor stamps into it)
    WideSys w;
    memset(&w, 0, sizeof w);
    w.n = n;
    for (int s2 = 1; s2 < n; s2 <<= 1) ++w.steps;
    w.off_L = (uint64_t)n * blk; w.off_rhs = 2 * (uint64_t)n * blk; w.count = w.off_rhs + (uint64_t)N;
    w.xw = dalloc<double>(w.count);
    w.U = dalloc<double>(n * blk); w.YL = dalloc<double>(n * blk); w.YU = dalloc<double>(n * blk); w.yr = dalloc<double>((size_t)N);
    WideSys *dw = dalloc<WideSys>(1);
    CHK(hipMemcpy(dw, &w, sizeof w, hipMemcpyHostToDevice));
    d.wide = dw;
    auto upload = [&] {
        CHK(hipMemcpy(w.xw, hD.data(), n * blk * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(w.xw + w.off_L, hL.data(), n * blk * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(w.xw + w.off_rhs, hr.data(), (size_t)N * 8, hipMemcpyHostToDevice));
        CHK(hipMemset(d.st, 0, sizeof(State)));
    };
    if (configure_wide()) { printf("configure failed\n"); return 1; }
    Launcher L;
    L.wide = w;
    CHK(hipStreamCreate(&L.stream));
    setvbuf(stdout, nullptr, _IONBF, 0);

    // host references: Cholesky of block 0's D (lower G), forward solves
    auto chol = [&](const double *A, std::vector<double> &G) {
        G.assign(blk, 0.0);
        for (int j = 0; j < B; ++j) {
            double s = A[(size_t)j * B + j];
            for (int k = 0; k < j; ++k) s -= G[(size_t)j * B + k] * G[(size_t)j * B + k];
            const double g = sqrt(s);
            G[(size_t)j * B + j] = g;
            for (int i = j + 1; i < B; ++i) {
                double v = A[(size_t)i * B + j];
                for (int k = 0; k < j; ++k) v -= G[(size_t)i * B + k] * G[(size_t)j * B + k];
                G[(size_t)i * B + j] = v / g;
            }
        }
    };
    // ---- first factor launch on its own ----
    upload();
    LAUNCH(KC_BCR_FACTOR, k_wd_factor<0>, dim3(n * 3), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, 0, 3);
    CHK(hipStreamSynchronize(L.stream));
    CHK(hipGetLastError());
    {
        std::vector<double> oYL(n * blk), oYU(n * blk), oyr((size_t)N), G, Y(blk), R(blk);
        CHK(hipMemcpy(oYL.data(), w.YL, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oYU.data(), w.YU, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oyr.data(), w.yr, (size_t)N * 8, hipMemcpyDeviceToHost));
        double eL = 0, eU = 0, er = 0;
        for (int b = 0; b < n; ++b) {
            chol(&hD[b * blk], G);
            auto fwd = [&](const double *Rm, int ncol, double *Ym) {
                for (int c = 0; c < ncol; ++c)
                    for (int i = 0; i < B; ++i) {
                        double v = Rm[(size_t)i * ncol + c];
                        for (int k = 0; k < i; ++k) v -= G[(size_t)i * B + k] * Ym[(size_t)k * ncol + c];
                        Ym[(size_t)i * ncol + c] = v / G[(size_t)i * B + i];
                    }
            };
            if (b >= 1) { fwd(&hL[b * blk], B, Y.data()); eL = fmax(eL, maxrel(&oYL[b * blk], Y.data(), blk)); }
            if (b + 1 < n) {
                for (int i = 0; i < B; ++i) for (int j = 0; j < B; ++j) R[(size_t)i * B + j] = hL[(b + 1) * blk + (size_t)j * B + i];
                fwd(R.data(), B, Y.data());
                eU = fmax(eU, maxrel(&oYU[b * blk], Y.data(), blk));
            }
            std::vector<double> yv(B);
            fwd(&hr[(size_t)b * B], 1, yv.data());
            er = fmax(er, maxrel(&oyr[(size_t)b * B], yv.data(), B));
        }
        printf("factor step 0: max rel err YL %.2e YU %.2e yr %.2e\n", eL, eU, er);
    }
    // ---- whole solve against a dense host solve ----
    upload();
    launch_wide_solve(L, d);
    CHK(hipStreamSynchronize(L.stream));
    CHK(hipGetLastError());
    std::vector<double> x((size_t)N), xr(hr);
    CHK(hipMemcpy(x.data(), d.x0, (size_t)N * 8, hipMemcpyDeviceToHost));
    State hst;
    CHK(hipMemcpy(&hst, d.st, sizeof hst, hipMemcpyDeviceToHost));
    if (n <= 48) {
        std::vector<double> A((size_t)N * N, 0.0);
        for (int b = 0; b < n; ++b)
            for (int i = 0; i < B; ++i)
                for (int j = 0; j < B; ++j) {
                    A[(size_t)(b * B + i) * N + b * B + j] = hD[b * blk + (size_t)i * B + j];
                    if (b > 0) {
                        A[(size_t)(b * B + i) * N + (b - 1) * B + j] = hL[b * blk + (size_t)i * B + j];
                        A[(size_t)((b - 1) * B + j) * N + b * B + i] = hL[b * blk + (size_t)i * B + j];
                    }
                }
        const int hb = 2 * B;       // band
        for (int j = 0; j < N; ++j) {
            double sdiag = A[(size_t)j * N + j];
            for (int k = std::max(0, j - hb); k < j; ++k) sdiag -= A[(size_t)j * N + k] * A[(size_t)j * N + k];
            const double gjj = sqrt(sdiag);
            A[(size_t)j * N + j] = gjj;
            for (int i = j + 1; i < std::min(N, j + hb + 1); ++i) {
                double v = A[(size_t)i * N + j];
                for (int k = std::max(0, i - hb); k < j; ++k) v -= A[(size_t)i * N + k] * A[(size_t)j * N + k];
                A[(size_t)i * N + j] = v / gjj;
            }
        }
        for (int i = 0; i < N; ++i) { double v = xr[i]; for (int k = std::max(0, i - hb); k < i; ++k) v -= A[(size_t)i * N + k] * xr[k]; xr[i] = v / A[(size_t)i * N + i]; }
        for (int i = N - 1; i >= 0; --i) { double v = xr[i]; for (int k = i + 1; k < std::min(N, i + hb + 1); ++k) v -= A[(size_t)k * N + i] * xr[k]; xr[i] = v / A[(size_t)i * N + i]; }
        printf("n = %d blocks (%d steps): solution against a dense host solve: max rel err %.2e (step_failed %d)\n", n, w.steps, maxrel(x.data(), xr.data(), (size_t)N), hst.step_failed);
    }
    // ---- timing ----
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time_us = [&](auto fn, int reps) {
        fn();
        CHK(hipStreamSynchronize(L.stream));
        CHK(hipEventRecord(e0, L.stream));
        for (int i = 0; i < reps; ++i) fn();
        CHK(hipEventRecord(e1, L.stream));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        return 1000.0 * ms / reps;
    };
    upload();
    printf("factor (step 0, 3 workgroups per block): %.2f us / launch\n", time_us([&] { LAUNCH(KC_BCR_FACTOR, k_wd_factor<0>, dim3(n * 3), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, 0, 3); }, 20));
#ifdef WD_STAMPS
    {
        LAUNCH(KC_BCR_FACTOR, k_wd_factor<0>, dim3(n * 3), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, 0, 3);
        LAUNCH(KC_BCR_FACTOR, k_wd_factor<0>, dim3(n * 3), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, 0, 3);
        CHK(hipDeviceSynchronize());
        unsigned long long h[128];
        CHK(hipMemcpy(h, d.dbg, sizeof h, hipMemcpyDeviceToHost));
        for (int wv = 0; wv < 2; ++wv) {
            printf("stamps of wave %d of a middle work-group (shader cycles after entry): load %llu | first tile factored %llu | barrier %llu\n", wv,
                   h[64 * wv + 1] - h[64 * wv], h[64 * wv + 2] - h[64 * wv], h[64 * wv + 3] - h[64 * wv]);
            for (int k = 0; k < WNT; ++k)
                printf("   k = %d: panel done %llu, barrier %llu, trailing / next factor done %llu, barrier %llu\n", k, h[64 * wv + 4 + 4 * k] - h[64 * wv],
                       h[64 * wv + 5 + 4 * k] - h[64 * wv], h[64 * wv + 6 + 4 * k] - h[64 * wv], h[64 * wv + 7 + 4 * k] - h[64 * wv]);
            if (wv) printf("   stores begin %llu\n", h[64 * wv + 40] - h[64 * wv]);
        }
    }
#endif
    printf("reduce (step 0):                         %.2f us / launch\n", time_us([&] { LAUNCH(KC_BCR_REDUCE, k_wd_reduce, dim3(n * WR_WG_PER_BLOCK), dim3(256), 0, d, 0); }, 20));
    upload();
    printf("factor (decoupled last step):            %.2f us / launch\n", time_us([&] { LAUNCH(KC_BCR_FACTOR, k_wd_factor<1>, dim3(n), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, w.steps, 1); }, 20));
    printf("whole solve (%d launches, data degenerates over repeats: timing only): %.1f us\n", 2 * w.steps + 1, time_us([&] { launch_wide_solve(L, d); }, 10));
    return 0;
}

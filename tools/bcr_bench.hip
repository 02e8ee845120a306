// Stand-alone check + timing of the matrix-core factor / reduce kernels of the reduced-camera solve
// (ceres_slam_amd/csrc/ssba_bcr_mfma.hip) on synthetic block-tridiagonal data: the kernels are driven through a
// hand-filled Dev (parallel cyclic reduction plan of n blocks), their outputs are compared with a host Cholesky, and
// launches are timed with HIP events.  Not part of the product path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ceres_slam_amd/csrc tools/bcr_bench.hip -o tools/bcr_bench && tools/bcr_bench [n]
#include "../ceres_slam_amd/csrc/ssba_bcr_mfma.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace ssba;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void chol(const double *A, double *G) {      // lower, row-major BD x BD
    memset(G, 0, sizeof(double) * BD * BD);
    for (int j = 0; j < BD; ++j) {
        double s = A[j * BD + j];
        for (int k = 0; k < j; ++k) s -= G[j * BD + k] * G[j * BD + k];
        const double g = sqrt(s);
        G[j * BD + j] = g;
        for (int i = j + 1; i < BD; ++i) {
            double v = A[i * BD + j];
            for (int k = 0; k < j; ++k) v -= G[i * BD + k] * G[j * BD + k];
            G[i * BD + j] = v / g;
        }
    }
}
static void fwd(const double *G, const double *B, double *Y, int ncol) {   // Y = G^-1 B, B row-major BD x ncol
    for (int c = 0; c < ncol; ++c)
        for (int i = 0; i < BD; ++i) {
            double v = B[i * ncol + c];
            for (int k = 0; k < i; ++k) v -= G[i * BD + k] * Y[k * ncol + c];
            Y[i * ncol + c] = v / G[i * BD + i];
        }
}
static double maxrel(const double *a, const double *b, size_t n) {
    double e = 0, m = 0;
    for (size_t i = 0; i < n; ++i) { e = fmax(e, fabs(a[i] - b[i])); m = fmax(m, fabs(b[i])); }
    return e / fmax(m, 1e-300);
}

template <class T> static T *dalloc(size_t n) { T *p; CHK(hipMalloc(&p, n * sizeof(T))); CHK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 84;
    const size_t blk = (size_t)BD * BD;
    std::mt19937_64 rng(42);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> hD(n * blk), hL(n * blk), hLb(n * blk), hLbT(n * blk), hr(n * BD);
    for (int b = 0; b < n; ++b) {
        std::vector<double> A(BD * 2 * BD);
        for (auto &v : A) v = nd(rng);
        for (int i = 0; i < BD; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0;
                for (int k = 0; k < 2 * BD; ++k) s += A[i * 2 * BD + k] * A[j * 2 * BD + k];
                s = s / (2 * BD) + (i == j ? 1.0 : 0.0);
                hD[b * blk + i * BD + j] = hD[b * blk + j * BD + i] = s;
            }
    }
    for (auto &v : hL) v = 0.3 * nd(rng);
    for (auto &v : hLb) v = 0.3 * nd(rng);
    for (auto &v : hLbT) v = 0.3 * nd(rng);
    for (auto &v : hr) v = nd(rng);

    Dev d;
    memset(&d, 0, sizeof d);
    d.st = dalloc<State>(1);
    d.dbg = dalloc<unsigned long long>(8192);
    d.n_levels = 1;
    d.lev[0].n = n;
    d.lev[0].D = dalloc<double>(n * blk); d.lev[0].L = dalloc<double>(n * blk); d.lev[0].r = dalloc<double>(n * BD);
    d.pcr.level = 0; d.pcr.n = n; d.pcr.steps = 0;
    for (int s2 = 1; s2 < n; s2 <<= 1) ++d.pcr.steps;
    d.pcr.Lbuf = dalloc<double>(n * blk); d.pcr.LbufT = dalloc<double>(n * blk);
    d.pcr.YL = dalloc<double>(n * blk); d.pcr.YU = dalloc<double>(n * blk); d.pcr.yr = dalloc<double>(n * BD);
    auto upload = [&] {
        CHK(hipMemcpy(d.lev[0].D, hD.data(), n * blk * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d.lev[0].L, hL.data(), n * blk * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d.lev[0].r, hr.data(), n * BD * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d.pcr.Lbuf, hLb.data(), n * blk * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d.pcr.LbufT, hLbT.data(), n * blk * 8, hipMemcpyHostToDevice));
    };
    upload();
    setvbuf(stdout, nullptr, _IONBF, 0);
    d.x0 = dalloc<double>((size_t)n * BD);
    d.nfree = n * SBP;
    {
        std::vector<int> pos(n);
        for (int i = 0; i < n; ++i) pos[i] = i;
        int *dpos = dalloc<int>(n);
        CHK(hipMemcpy(dpos, pos.data(), n * sizeof(int), hipMemcpyHostToDevice));
        d.lev[0].pos = dpos;
    }
    if (configure_bcr_mf()) { printf("configure failed\n"); return 1; }
    Launcher L;
    CHK(hipStreamCreate(&L.stream));

    std::vector<double> G(n * blk), gYL(n * blk), gYU(n * blk), gyr(n * BD), oYL(n * blk), oYU(n * blk), oyr(n * BD);
    for (int b = 0; b < n; ++b) chol(&hD[b * blk], &G[b * blk]);
    // ---- factor, step lev = 1 (operands Lbuf / LbufT, no transposed storage) ----
    for (int lev : {1, 0}) {
        const int s = 1 << lev, last = n - 1;
        launch_bcr_factor_mf(L, d, n, lev, 0, 2, true);
        CHK(hipStreamSynchronize(L.stream));
        CHK(hipMemcpy(oYL.data(), d.pcr.YL, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oYU.data(), d.pcr.YU, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oyr.data(), d.pcr.yr, n * BD * 8, hipMemcpyDeviceToHost));
        double eL = 0, eU = 0, er = 0;
        std::vector<double> Bm(blk), Y(blk), yv(BD);
        for (int b = 0; b < n; ++b) {
            const bool hasL = b - s >= 0, hasU = b + s <= last;
            if (hasL) {
                for (int i = 0; i < BD; ++i)
                    for (int j = 0; j < BD; ++j)
                        Bm[i * BD + j] = lev == 0 ? ((b & 1) == 0 ? hL[b * blk + j * BD + i] : hL[b * blk + i * BD + j]) : hLb[b * blk + i * BD + j];
                fwd(&G[b * blk], Bm.data(), Y.data(), BD);
                eL = fmax(eL, maxrel(&oYL[b * blk], Y.data(), blk));
                if (lev == 1) memcpy(&gYL[b * blk], Y.data(), blk * 8);
            }
            if (hasU) {
                const int u = lev == 0 ? b + 1 : b + s;
                for (int i = 0; i < BD; ++i)
                    for (int j = 0; j < BD; ++j)
                        Bm[i * BD + j] = lev == 0 ? ((b & 1) == 0 ? hL[u * blk + j * BD + i] : hL[u * blk + i * BD + j]) : hLbT[u * blk + i * BD + j];
                fwd(&G[b * blk], Bm.data(), Y.data(), BD);
                eU = fmax(eU, maxrel(&oYU[b * blk], Y.data(), blk));
                if (lev == 1) memcpy(&gYU[b * blk], Y.data(), blk * 8);
            }
            fwd(&G[b * blk], &hr[b * BD], yv.data(), 1);
            er = fmax(er, maxrel(&oyr[b * BD], yv.data(), BD));
            if (lev == 1) memcpy(&gyr[b * BD], yv.data(), BD * 8);
        }
        printf("factor lev %d: max rel err YL %.2e YU %.2e yr %.2e\n", lev, eL, eU, er);
    }
    // ---- reduce, step lev = 1, from the factor outputs of lev 1 ----
    {
        const int lev = 1, s = 2, last = n - 1;
        launch_bcr_factor_mf(L, d, n, lev, 0, 2, true);
        launch_bcr_reduce_mf(L, d, n, 2, lev, 2);
        CHK(hipStreamSynchronize(L.stream));
        std::vector<double> oD(n * blk), orr(n * BD), oLb(n * blk), oLbT(n * blk);
        CHK(hipMemcpy(oD.data(), d.lev[0].D, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(orr.data(), d.lev[0].r, n * BD * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oLb.data(), d.pcr.Lbuf, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(oLbT.data(), d.pcr.LbufT, n * blk * 8, hipMemcpyDeviceToHost));
        double eD = 0, er = 0, eL = 0, eLT = 0;
        std::vector<double> rD(blk), rr(BD), rL(blk), rLT(blk);
        for (int e = 0; e < n; ++e) {
            const int prev = e - s, next = e + s;
            const bool hasPrev = prev >= 0, hasNext = next <= last;
            memcpy(rD.data(), &hD[e * blk], blk * 8);
            memcpy(rr.data(), &hr[e * BD], BD * 8);
            for (int i = 0; i < BD; ++i) {
                for (int j = 0; j < BD; ++j) {
                    double v = 0;
                    if (hasPrev) for (int k = 0; k < BD; ++k) v += gYU[prev * blk + k * BD + i] * gYU[prev * blk + k * BD + j];
                    if (hasNext) for (int k = 0; k < BD; ++k) v += gYL[next * blk + k * BD + i] * gYL[next * blk + k * BD + j];
                    rD[i * BD + j] -= v;
                }
                double v = 0;
                if (hasPrev) for (int k = 0; k < BD; ++k) v += gYU[prev * blk + k * BD + i] * gyr[prev * BD + k];
                if (hasNext) for (int k = 0; k < BD; ++k) v += gYL[next * blk + k * BD + i] * gyr[next * BD + k];
                rr[i] -= v;
            }
            eD = fmax(eD, maxrel(&oD[e * blk], rD.data(), blk));
            er = fmax(er, maxrel(&orr[e * BD], rr.data(), BD));
            if (hasPrev && prev - s >= 0) {
                for (int i = 0; i < BD; ++i)
                    for (int j = 0; j < BD; ++j) {
                        double v = 0;
                        for (int k = 0; k < BD; ++k) v += gYU[prev * blk + k * BD + i] * gYL[prev * blk + k * BD + j];
                        rL[i * BD + j] = -v;
                        rLT[j * BD + i] = -v;
                    }
                eL = fmax(eL, maxrel(&oLb[e * blk], rL.data(), blk));
                eLT = fmax(eLT, maxrel(&oLbT[e * blk], rLT.data(), blk));
            }
        }
        printf("reduce lev 1: max rel err D %.2e r %.2e Lbuf %.2e LbufT %.2e\n", eD, er, eL, eLT);
        upload();
    }
    // ---- top step: G in place ----
    {
        launch_bcr_factor_mf(L, d, n, d.pcr.steps, 1, 2, false);
        CHK(hipStreamSynchronize(L.stream));
        std::vector<double> oD(n * blk), orr(n * BD);
        CHK(hipMemcpy(oD.data(), d.lev[0].D, n * blk * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(orr.data(), d.lev[0].r, n * BD * 8, hipMemcpyDeviceToHost));
        double eG = 0, er = 0;
        std::vector<double> yv(BD);
        for (int b = 0; b < n; ++b) {
            for (int i = 0; i < BD; ++i)
                for (int j = 0; j <= i; ++j) {
                    const double ref = i == j ? 1.0 / G[b * blk + i * BD + j] : G[b * blk + i * BD + j];
                    eG = fmax(eG, fabs(oD[b * blk + i * BD + j] - ref));
                }
            fwd(&G[b * blk], &hr[b * BD], yv.data(), 1);
            er = fmax(er, maxrel(&orr[b * BD], yv.data(), BD));
        }
        printf("factor top: max abs err G %.2e, rel err yr %.2e\n", eG, er);
        upload();
    }
    // ---- whole chains: factor + reduce per step (classic) against one launch per step (fused plan, PcrFused) ----
    {
        const double cpl = 0.04;        // weak couplings: the block-tridiagonal matrix stays positive definite
        std::vector<double> cL(n * blk);
        for (auto &v : cL) v = cpl * nd(rng);
        for (int q = 0; q < 2; ++q) {
            d.pcrf.Dpp[q] = dalloc<double>(n * blk); d.pcrf.rpp[q] = dalloc<double>(n * BD);
            d.pcrf.GLL[q] = dalloc<double>(n * blk); d.pcrf.GUU[q] = dalloc<double>(n * blk);
            d.pcrf.GUL[q] = dalloc<double>(n * blk); d.pcrf.GULT[q] = dalloc<double>(n * blk);
            d.pcrf.gL[q] = dalloc<double>(n * BD); d.pcrf.gU[q] = dalloc<double>(n * BD);
        }
        d.pcrf.on = 1;
        auto upload2 = [&] {
            CHK(hipMemcpy(d.lev[0].D, hD.data(), n * blk * 8, hipMemcpyHostToDevice));
            CHK(hipMemcpy(d.lev[0].L, cL.data(), n * blk * 8, hipMemcpyHostToDevice));
            CHK(hipMemcpy(d.lev[0].r, hr.data(), n * BD * 8, hipMemcpyHostToDevice));
            CHK(hipMemset(d.st, 0, sizeof(State)));
        };
        auto chain = [&](bool fused) {
            for (int q = 0; q < d.pcr.steps; ++q) {
                if (fused) launch_pcr_fused_step(L, d, n, q);
                else { launch_bcr_factor_mf(L, d, n, q, 0, 2, true); launch_bcr_reduce_mf(L, d, n, 2, q, 2); }
            }
            if (fused) launch_pcr_fused_top(L, d, n, d.pcr.steps, 1);
            else launch_bcr_factor_mf(L, d, n, d.pcr.steps, 1, 2, false, false, 1);
        };
        std::vector<double> xc((size_t)n * BD), xf((size_t)n * BD);
        State hst;
        upload2(); chain(false); CHK(hipStreamSynchronize(L.stream));
        CHK(hipMemcpy(xc.data(), d.x0, xc.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&hst, d.st, sizeof hst, hipMemcpyDeviceToHost));
        printf("classic chain: step_failed %d\n", hst.step_failed);
        upload2(); chain(true); CHK(hipStreamSynchronize(L.stream));
        CHK(hipMemcpy(xf.data(), d.x0, xf.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&hst, d.st, sizeof hst, hipMemcpyDeviceToHost));
        printf("fused chain:   step_failed %d\n", hst.step_failed);
        printf("solution, fused against classic: max rel diff %.2e\n", maxrel(xf.data(), xc.data(), xf.size()));
        if (n <= 16) {      // dense host solve of the block-tridiagonal system
            const int N = n * BD;
            std::vector<double> A((size_t)N * N, 0.0), x(hr.begin(), hr.begin() + N);
            for (int b = 0; b < n; ++b)
                for (int i = 0; i < BD; ++i)
                    for (int j = 0; j < BD; ++j) {
                        A[(size_t)(b * BD + i) * N + b * BD + j] = hD[b * blk + i * BD + j];
                        if (b > 0) {        // S[b, b-1]: stored transposed for even b
                            const double v = (b & 1) == 0 ? cL[b * blk + j * BD + i] : cL[b * blk + i * BD + j];
                            A[(size_t)(b * BD + i) * N + (b - 1) * BD + j] = v;
                            A[(size_t)((b - 1) * BD + j) * N + b * BD + i] = v;
                        }
                    }
            for (int j = 0; j < N; ++j) {       // in-place Cholesky (lower) + two triangular solves
                double sdiag = A[(size_t)j * N + j];
                for (int k = 0; k < j; ++k) sdiag -= A[(size_t)j * N + k] * A[(size_t)j * N + k];
                const double gjj = sqrt(sdiag);
                A[(size_t)j * N + j] = gjj;
                for (int i = j + 1; i < N; ++i) {
                    double v = A[(size_t)i * N + j];
                    for (int k = 0; k < j; ++k) v -= A[(size_t)i * N + k] * A[(size_t)j * N + k];
                    A[(size_t)i * N + j] = v / gjj;
                }
            }
            for (int i = 0; i < N; ++i) { double v = x[i]; for (int k = 0; k < i; ++k) v -= A[(size_t)i * N + k] * x[k]; x[i] = v / A[(size_t)i * N + i]; }
            for (int i = N - 1; i >= 0; --i) { double v = x[i]; for (int k = i + 1; k < N; ++k) v -= A[(size_t)k * N + i] * x[k]; x[i] = v / A[(size_t)i * N + i]; }
            printf("solution against a dense host solve: classic %.2e, fused %.2e (max rel)\n", maxrel(xc.data(), x.data(), N), maxrel(xf.data(), x.data(), N));
        }
        // per-launch times on VALID data (a chain leaves G in place of D: every run starts from a fresh upload)
        for (int fused = 0; fused < 2; ++fused) {
            std::vector<hipEvent_t> ev;
            auto mark = [&] { hipEvent_t e; CHK(hipEventCreate(&e)); CHK(hipEventRecord(e, L.stream)); ev.push_back(e); };
            double best_total = 1e30;
            std::vector<float> best;
            for (int rep = 0; rep < 5; ++rep) {
                upload2();
                CHK(hipStreamSynchronize(L.stream));
                ev.clear();
                mark();
                for (int q = 0; q < d.pcr.steps; ++q) {
                    if (fused) { launch_pcr_fused_step(L, d, n, q); mark(); }
                    else { launch_bcr_factor_mf(L, d, n, q, 0, 2, true); mark(); launch_bcr_reduce_mf(L, d, n, 2, q, 2); mark(); }
                }
                if (fused) launch_pcr_fused_top(L, d, n, d.pcr.steps, 1);
                else launch_bcr_factor_mf(L, d, n, d.pcr.steps, 1, 2, false, false, 1);
                mark();
                CHK(hipStreamSynchronize(L.stream));
                std::vector<float> ms(ev.size() - 1);
                float tot;
                CHK(hipEventElapsedTime(&tot, ev.front(), ev.back()));
                for (size_t i = 0; i + 1 < ev.size(); ++i) CHK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
                if (tot < best_total) { best_total = tot; best = ms; }
                for (auto e : ev) CHK(hipEventDestroy(e));
            }
            printf("%s chain, eager with an event after every launch: %.1f us total; per launch:", fused ? "fused  " : "classic", 1000.0 * best_total);
            for (float v : best) printf(" %.1f", 1000.0 * v);
            printf("\n");
            CHK(hipMemcpy(&hst, d.st, sizeof hst, hipMemcpyDeviceToHost));
            if (hst.step_failed) printf("  (step_failed set!)\n");
        }
#ifdef SSBA_STAMPS
        {
            upload2();
            for (int q = 0; q < 3 && q < d.pcr.steps; ++q) launch_pcr_fused_step(L, d, n, q);
            CHK(hipStreamSynchronize(L.stream));
            CHK(hipMemset(d.dbg, 0, 8192 * 8));
            launch_pcr_fused_step(L, d, n, 3);
            CHK(hipStreamSynchronize(L.stream));
            std::vector<unsigned long long> stp(8192), stq(8192);
            CHK(hipMemcpy(stp.data(), d.dbg, 8192 * 8, hipMemcpyDeviceToHost));
            {   // is the workgroup -> XCD placement the same from launch to launch?  (slot 5 of the per-workgroup stamps)
                launch_pcr_fused_step(L, d, n, 4);
                CHK(hipStreamSynchronize(L.stream));
                CHK(hipMemcpy(stq.data(), d.dbg, 8192 * 8, hipMemcpyDeviceToHost));
                const int nwg = n * (n <= 85 ? 3 : 2);
                int same = 0, rr = 0;
                for (int g = 0; g < nwg; ++g) {
                    same += stp[4096 + 8 * g + 5] == stq[4096 + 8 * g + 5];
                    rr += (int)(stp[4096 + 8 * g + 5] - 1000) == ((int)(stp[4096 + 5] - 1000) + g) % 8;
                }
                printf("XCD placement: %d of %d workgroups on the same XCD in two consecutive launches; %d of %d follow (xcd of workgroup 0 + id) %% 8; "
                       "workgroup 0 on XCD %d then %d\n", same, nwg, rr, nwg, (int)(stp[4096 + 5] - 1000), (int)(stq[4096 + 5] - 1000));
                // where do the workgroups of a block run, and do the producers of a block's operands (blocks e - h, e + h of the
                // launch before) run on the XCD that consumes them?  xcd_map on the host: workgroup id -> (block, part)
                auto host_map = [&](int id, int nn, int ny, int &e, int &y) {
                    const int N = nn * ny, c = id & 7;
                    const int q = (N + 7) >> 3, r = (N + 7) & 7;
                    int idx = (id >> 3) + c * q - std::max(0, c - (r + 1));
                    e = 0; y = 0;
                    for (int k = 0; k < 8; ++k) {
                        const int sk = ((nn - k + 7) >> 3) * ny;
                        if (idx < sk) { const int bq = idx / ny; y = idx - bq * ny; e = k + 8 * bq; return; }
                        idx -= sk;
                    }
                };
                const int ny = n <= 85 ? 3 : 2;
                std::vector<int> xa(n * ny, -1), xb(n * ny, -1);         // XCD of (block, part) in launch 3 / launch 4
                for (int g = 0; g < nwg && g < 500; ++g) {
                    int e, y;
                    host_map(g, n, ny, e, y);
                    xa[e * ny + y] = (int)(stp[4096 + 8 * g + 5] - 1000);
                    xb[e * ny + y] = (int)(stq[4096 + 8 * g + 5] - 1000);
                }
                int together = 0, counted = 0, fed = 0, feeds = 0;
                for (int e = 0; e < n; ++e) {
                    if (xa[e * ny] < 0 || xb[e * ny] < 0) continue;
                    ++counted;
                    bool one = true;
                    for (int y = 1; y < ny; ++y) one = one && xb[e * ny + y] == xb[e * ny];
                    together += one;
                    // launch 4 (stride 16) assembles block e from the Gram products blocks e - 8 and e + 8 formed in launch 3
                    for (int h : {-8, 8}) {
                        const int b = e + h;
                        if (b < 0 || b >= n || xa[b * ny] < 0) continue;
                        for (int y = 0; y < ny; ++y) {          // every part of the producer writes a share of the Gram tiles
                            ++feeds;
                            fed += xa[b * ny + y] == xb[e * ny];
                        }
                    }
                }
                printf("XCD placement by block: %d of %d blocks have all %d workgroups on one XCD (launch 4); %d of %d (producer part, consumer) "
                       "pairs of launch 3 -> launch 4 (operands e -/+ 8) sit on the same XCD\n", together, counted, ny, fed, feeds);
                printf("XCD of workgroup id 0..23 in launch 3:");
                for (int g = 0; g < 24 && g < nwg; ++g) printf(" %d", (int)(stp[4096 + 8 * g + 5] - 1000));
                printf("\n");
            }
            {
                const int nwg = n * (n <= 85 ? 3 : n <= 128 ? 2 : 1);
                unsigned long long lo = ~0ull;
                for (int g = 0; g < nwg; ++g) lo = std::min(lo, stp[4096 + 8 * g]);
                printf("fused step 3, every workgroup (thread 0) on the 100 MHz clock, us since the first entry: entry loaded factored staged done\n");
                std::vector<std::pair<double, int>> order;
                for (int g = 0; g < nwg; ++g) order.push_back({(stp[4096 + 8 * g + 4] - lo) / 100.0, g});
                std::sort(order.begin(), order.end());
                for (int i = 0; i < nwg; i += (i < 8 || i >= nwg - 24) ? 1 : 16) {
                    const int g = order[i].second;
                    printf("  wg %3d (id %% 8 = %d, XCC_ID %d):", g, g & 7, (int)(stp[4096 + 8 * g + 5] - 1000));
                    for (int k = 0; k < 5; ++k) printf(" %6.2f", (double)(long long)(stp[4096 + 8 * g + k] - lo) / 100.0);
                    printf("\n");
                }
            }
            for (int w = 0; w < 8; ++w) {
                printf("fused step 3, wave %d stamps (cycles since wave 0's first):", w);
                for (int i = 0; i < 40; ++i) if (stp[w * 64 + i]) printf(" [%d]%lld", i, (long long)(stp[w * 64 + i] - stp[0]));
                printf("\n");
            }
        }
#endif
        upload();
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
    printf("n = %d blocks\n", n);
    printf("factor (coupled, lev 1):   %.2f us / launch\n", time_us([&] { launch_bcr_factor_mf(L, d, n, 1, 0, 2, true); }, 50));
    printf("factor (coupled, lev 0):   %.2f us / launch\n", time_us([&] { launch_bcr_factor_mf(L, d, n, 0, 0, 2, true); }, 50));
    printf("factor (decoupled, top=0): %.2f us / launch\n", time_us([&] { launch_bcr_factor_mf(L, d, n, d.pcr.steps, 0, 2, false); }, 50));
    launch_bcr_factor_mf(L, d, n, 1, 0, 2, true);
    printf("reduce (lev 1):            %.2f us / launch\n", time_us([&] { launch_bcr_reduce_mf(L, d, n, 2, 1, 2); }, 50));
#ifdef SSBA_STAMPS
    {
        upload();
        CHK(hipMemset(d.dbg, 0, 8192 * 8));
        launch_bcr_factor_mf(L, d, n, 1, 0, 2, true);
        CHK(hipStreamSynchronize(L.stream));
        std::vector<unsigned long long> st(8192);
        CHK(hipMemcpy(st.data(), d.dbg, 8192 * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            printf("wave %d stamps (cycles since first):", w);
            for (int i = 0; i < 40; ++i) if (st[w * 64 + i]) printf(" [%d]%lld", i, (long long)(st[w * 64 + i] - st[0]));
            printf("\n");
        }
        CHK(hipMemset(d.dbg, 0, 8192 * 8));
        launch_bcr_reduce_mf(L, d, n, 2, 1, 2);
        CHK(hipStreamSynchronize(L.stream));
        CHK(hipMemcpy(st.data(), d.dbg, 8192 * 8, hipMemcpyDeviceToHost));
        for (int y = 0; y < 3; ++y)
            for (int w = 0; w < 4; ++w) {
                const unsigned long long *q = &st[2048 + w * 64 + 16 * y];
                printf("reduce y %d wave %d:", y, w);
                for (int i = 1; i < 13; ++i) if (q[i]) printf(" [%d]%lld", i, (long long)(q[i] - q[0]));
                printf("\n");
            }
    }
#endif
    return 0;
}

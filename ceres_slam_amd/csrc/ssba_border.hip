// Arrowhead solve for the border of free shared blocks (config 3: light, Phong parameters, textures).
//
//   [S_pp S_pb; S_pb^T S_bb] [dp; db] = [-g_p^; -g_b^]
//
// S_pp is the block-tridiagonal reduced camera system that ssba_bcr.hip has just factored level by
// level (G in D, YL = G^-1 L in L, YU = G^-1 L_{i+1}^T) while solving x0 = S_pp^-1 (-g_p^).  The kernels
// here push the nb <= NBP columns of S_pb through the SAME factors (forward: yb = G^-1 B on odd blocks,
// B' = B_e - YU^T yb - YL^T yb on even blocks; backward: X_i = G^-T (yb - YL X_{i-1} - YU X_{i+1})),
// giving Zb = S_pp^-1 S_pb, then form and solve the small border system
//   (S_bb + D_b^2 - S_pb^T Zb) db = -g_b^ - S_pb^T x0 ,   dp = x0 - Zb db .
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssba_launch.h"
#include "ssba_types.h"

namespace ssba {

constexpr int LDT = BD + 1;            // padded LDS stride: column and row sweeps both stay conflict-light
constexpr int MR_THREADS = 1024;       // 16 waves x 2 columns = NBP
constexpr int UPD_THREADS = 576;       // 72 rows x 8 column groups of 4

static __device__ __forceinline__ double bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Staging copies with ALL global reads of a lane in flight before its first LDS store (a rolled copy loop waits for a
// full memory round trip per iteration: 9 + 4 of them per operand pair in the first version of k_bcrm_upd).
template <int NT> static __device__ __forceinline__ void stage_pad(double *dst, const double *__restrict__ src, bool transpose) {
    constexpr int NLD = (BD * BD / 2 + NT - 1) / NT;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * NT;
        v[q] = e < BD * BD / 2 ? s2[e] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * NT;
        if (e >= BD * BD / 2) continue;
        const int r = (2 * e) / BD, c = 2 * e - r * BD;
        if (transpose) { dst[c * LDT + r] = v[q].x; dst[(c + 1) * LDT + r] = v[q].y; }
        else { dst[r * LDT + c] = v[q].x; dst[r * LDT + c + 1] = v[q].y; }
    }
}
template <int N, int NT> static __device__ __forceinline__ void stage_flat(double *dst, const double *__restrict__ src) {
    constexpr int NLD = (N / 2 + NT - 1) / NT;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    double2 v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * NT;
        v[q] = e < N / 2 ? s2[e] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * NT;
        if (e < N / 2) d2[e] = v[q];
    }
}

// odd blocks: yb = G^-1 B in place (forward substitution, column sweep; lanes = rows, a wave = 2 columns)
// which = 2: parallel plan (ssba_bcr.hip), step `lev` of it (top: the last, decoupled one): every block, its factor of
// that step, columns read from Bb and written to yB (the block's own columns stay for k_bcrm_upd)
__global__ __launch_bounds__(MR_THREADS) void k_bcrm_fwd(Dev d, int lev, int top, int which) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    extern __shared__ __align__(16) double lds[];
    double *Gt = lds;   // Gt[k][i] = G[i][k]
    const BcrLevel &L = d.lev[which == 2 ? d.pcr.level : lev];
    const int blk = which == 2 ? (int)blockIdx.x : (top ? 0 : 2 * blockIdx.x + 1);
    const double *Gsrc = which == 2 && !top ? d.pcr.Gs + ((size_t)lev * L.n + blk) * BD * BD : L.D + (size_t)blk * BD * BD;
    stage_pad<MR_THREADS>(Gt, Gsrc, true);
    const double *Bin = which == 2 ? d.pcr.Bb + (size_t)blk * BD * NBP : L.B + (size_t)blk * BD * NBP;
    double *B = which == 2 ? d.pcr.yB + (size_t)blk * BD * NBP : L.B + (size_t)blk * BD * NBP;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, c0 = 2 * w, c1 = c0 + 1;
    const bool hiRow = lane < BD - 64;
    double lo0 = Bin[lane * NBP + c0], lo1 = Bin[lane * NBP + c1];
    double hi0 = hiRow ? Bin[(64 + lane) * NBP + c0] : 0.0, hi1 = hiRow ? Bin[(64 + lane) * NBP + c1] : 0.0;
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        const double ginv = Gt[k * LDT + k];
        const double x0 = bcast(lo0, k) * ginv, x1 = bcast(lo1, k) * ginv;
        const double gl = (lane > k) ? Gt[k * LDT + lane] : 0.0;
        const double gh = hiRow ? Gt[k * LDT + 64 + lane] : 0.0;
        lo0 = (lane == k) ? x0 : lo0 - gl * x0;
        lo1 = (lane == k) ? x1 : lo1 - gl * x1;
        hi0 -= gh * x0;
        hi1 -= gh * x1;
    }
#pragma unroll
    for (int k = 64; k < BD; ++k) {
        const int kk = k - 64;
        const double ginv = Gt[k * LDT + k];
        const double x0 = bcast(hi0, kk) * ginv, x1 = bcast(hi1, kk) * ginv;
        const double gh = (hiRow && lane > kk) ? Gt[k * LDT + 64 + lane] : 0.0;
        hi0 = (lane == kk) ? x0 : hi0 - gh * x0;
        hi1 = (lane == kk) ? x1 : hi1 - gh * x1;
    }
    B[lane * NBP + c0] = lo0;
    B[lane * NBP + c1] = lo1;
    if (hiRow) { B[(64 + lane) * NBP + c0] = hi0; B[(64 + lane) * NBP + c1] = hi1; }
}

// even blocks: B'(m) = B(e) - YU(e-1)^T yb(e-1) - YL(e+1)^T yb(e+1), e = 2m
__global__ __launch_bounds__(UPD_THREADS) void k_bcrm_upd(Dev d, int lev, int which) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    extern __shared__ __align__(16) double lds[];
    double *sA = lds, *sB = lds + BD * BD, *ya = lds + 2 * BD * BD, *yb = ya + BD * NBP;
    if (which == 2) {
        // parallel plan, stride s = 2^lev: B(e) -= YU(e-s)^T yB(e-s) + YL(e+s)^T yB(e+s), in place (own block only)
        const BcrLevel &P = d.lev[d.pcr.level];
        const int e = blockIdx.x, s = 1 << lev, prev = e - s, next = e + s;
        const bool hasPrev = prev >= 0, hasNext = next < P.n;
        if (!hasPrev && !hasNext) return;
        const size_t so = (size_t)lev * P.n;
        const int t = threadIdx.x, r = t % BD, cg = (t / BD) * 4;
        double *Be = d.pcr.Bb + ((size_t)e * BD + r) * NBP + cg;
        double acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = Be[c];       // own block: fetched under the staging, not after the barrier
        if (hasPrev) {
            stage_flat<BD * BD, UPD_THREADS>(sA, d.pcr.YU + (so + prev) * BD * BD);
            stage_flat<BD * NBP, UPD_THREADS>(ya, d.pcr.yB + (size_t)prev * BD * NBP);
        }
        if (hasNext) {
            stage_flat<BD * BD, UPD_THREADS>(sB, d.pcr.YL + (so + next) * BD * BD);
            stage_flat<BD * NBP, UPD_THREADS>(yb, d.pcr.yB + (size_t)next * BD * NBP);
        }
        __syncthreads();
        if (hasPrev)
            for (int k = 0; k < BD; ++k) {
                const double a = sA[k * BD + r];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] -= a * ya[k * NBP + cg + c];
            }
        if (hasNext)
            for (int k = 0; k < BD; ++k) {
                const double a = sB[k * BD + r];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] -= a * yb[k * NBP + cg + c];
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) Be[c] = acc[c];
        return;
    }
    const BcrLevel &L = d.lev[lev];
    const BcrLevel &N = d.lev[lev + 1];
    const int m = blockIdx.x, e = 2 * m;
    const bool hasPrev = e >= 1, hasNext = e + 1 < L.n;
    const int t = threadIdx.x, r = t % BD, cg = (t / BD) * 4;
    double acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = L.B[((size_t)e * BD + r) * NBP + cg + c];
    if (hasPrev) {
        stage_flat<BD * BD, UPD_THREADS>(sA, L.YU + (size_t)(m - 1) * BD * BD);
        stage_flat<BD * NBP, UPD_THREADS>(ya, L.B + (size_t)(e - 1) * BD * NBP);
    }
    if (hasNext) {
        stage_flat<BD * BD, UPD_THREADS>(sB, L.L + (size_t)(e + 1) * BD * BD);
        stage_flat<BD * NBP, UPD_THREADS>(yb, L.B + (size_t)(e + 1) * BD * NBP);
    }
    __syncthreads();
    if (hasPrev)
        for (int k = 0; k < BD; ++k) {
            const double a = sA[k * BD + r];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] -= a * ya[k * NBP + cg + c];
        }
    if (hasNext)
        for (int k = 0; k < BD; ++k) {
            const double a = sB[k * BD + r];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] -= a * yb[k * NBP + cg + c];
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) N.B[((size_t)m * BD + r) * NBP + cg + c] = acc[c];
}

// odd blocks, top-down: X_i = G^-T (yb_i - YL_i X_{i-1} - YU_i X_{i+1}); X lives in d.Zb at level-0 positions
// which = 2 (with top): the decoupled blocks of the parallel plan, X = G^-T yB for every block
// ride_r (which = 2): column NBP - 1 -- padding of the border -- carries the right-hand side of the reduced system instead: read
// from the level's r, solved by the same sweep (k_bcr_backsub's arithmetic, lane for lane), written to x0
__global__ __launch_bounds__(MR_THREADS) void k_bcrm_bwd(Dev d, int lev, int top, int which, int ride_r) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    extern __shared__ __align__(16) double lds[];
    double *sL = lds, *sU = lds + BD * LDT, *xm = lds + 2 * BD * LDT, *xp = xm + BD * NBP;
    const BcrLevel &L = d.lev[which == 2 ? d.pcr.level : lev];
    const int blk = which == 2 ? (int)blockIdx.x : (top ? 0 : 2 * blockIdx.x + 1);
    const bool hasU = !top && (blk + 1 < L.n);
    if (!top) {
        stage_pad<MR_THREADS>(sL, L.L + (size_t)blk * BD * BD, false);
        stage_flat<BD * NBP, MR_THREADS>(xm, d.Zb + (((size_t)(blk - 1) << lev) * BD) * NBP);
    }
    if (hasU) {
        stage_pad<MR_THREADS>(sU, L.YU + (size_t)blockIdx.x * BD * BD, false);
        stage_flat<BD * NBP, MR_THREADS>(xp, d.Zb + (((size_t)(blk + 1) << lev) * BD) * NBP);
    }
    const double *B = which == 2 ? d.pcr.yB + (size_t)blk * BD * NBP : L.B + (size_t)blk * BD * NBP;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, c0 = 2 * w, c1 = c0 + 1;
    const bool hiRow = lane < BD - 64;
    const bool rr = ride_r && which == 2 && c1 == NBP - 1;      // (wave-uniform: the last wave's second column)
    const double *rv = L.r + (size_t)blk * BD;
    double lo0 = B[lane * NBP + c0], lo1 = rr ? rv[lane] : B[lane * NBP + c1];
    double hi0 = hiRow ? B[(64 + lane) * NBP + c0] : 0.0, hi1 = hiRow ? (rr ? rv[64 + lane] : B[(64 + lane) * NBP + c1]) : 0.0;
    __syncthreads();
    if (!top) {
        for (int k = 0; k < BD; ++k) {
            const double a = sL[lane * LDT + k], ah = hiRow ? sL[(64 + lane) * LDT + k] : 0.0;
            const double v0 = xm[k * NBP + c0], v1 = xm[k * NBP + c1];
            lo0 -= a * v0; lo1 -= a * v1; hi0 -= ah * v0; hi1 -= ah * v1;
        }
        if (hasU)
            for (int k = 0; k < BD; ++k) {
                const double a = sU[lane * LDT + k], ah = hiRow ? sU[(64 + lane) * LDT + k] : 0.0;
                const double v0 = xp[k * NBP + c0], v1 = xp[k * NBP + c1];
                lo0 -= a * v0; lo1 -= a * v1; hi0 -= ah * v0; hi1 -= ah * v1;
            }
    }
    __syncthreads();
    double *sG = lds;   // rows of G over the YL staging area
    stage_pad<MR_THREADS>(sG, L.D + (size_t)blk * BD * BD, false);
    __syncthreads();
#pragma unroll
    for (int k = BD - 1; k >= 64; --k) {
        const int kk = k - 64;
        const double ginv = sG[k * LDT + k];
        const double x0 = bcast(hi0, kk) * ginv, x1 = bcast(hi1, kk) * ginv;
        const double gl = sG[k * LDT + lane];
        const double gh = (lane < kk) ? sG[k * LDT + 64 + lane] : 0.0;
        lo0 -= gl * x0; lo1 -= gl * x1;
        hi0 = (lane == kk) ? x0 : hi0 - gh * x0;
        hi1 = (lane == kk) ? x1 : hi1 - gh * x1;
    }
#pragma unroll 4
    for (int k = 63; k >= 0; --k) {
        const double ginv = sG[k * LDT + k];
        const double x0 = bcast(lo0, k) * ginv, x1 = bcast(lo1, k) * ginv;
        const double gl = (lane < k) ? sG[k * LDT + lane] : 0.0;
        lo0 = (lane == k) ? x0 : lo0 - gl * x0;
        lo1 = (lane == k) ? x1 : lo1 - gl * x1;
    }
    double *X = d.Zb + ((which == 2 ? (size_t)L.pos[blk] : ((size_t)blk << lev)) * BD) * NBP;
    X[lane * NBP + c0] = lo0;
    if (hiRow) X[(64 + lane) * NBP + c0] = hi0;
    if (rr) {
        double *xi = d.x0 + (size_t)d.chain0 * BD + (size_t)L.pos[blk] * BD;
        xi[lane] = lo1;
        if (hiRow) xi[64 + lane] = hi1;
    } else {
        X[lane * NBP + c1] = lo1;
        if (hiRow) X[(64 + lane) * NBP + c1] = hi1;
    }
}

// partial sums of S_pb^T [Zb | x0] over a slice of pose rows: thread (a, b) of the NBP x NBP result.  The slice goes
// through LDS in chunks of GRAM_ROWS rows (bulk loads, all in flight) -- the first version read its two operands from
// global memory row by row: ~95 dependent round trips per workgroup, 31 us for 0.2 MFLOP.
constexpr int GRAM_ROWS = 96;
__global__ __launch_bounds__(1024) void k_border_gram(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double sS[GRAM_ROWS * NBP], sZ[GRAM_ROWS * NBP], sx[GRAM_ROWS];
    const int t = threadIdx.x, a = t / NBP, b = t - a * NBP;
    const int rows = d.nf_pad * 6, per = (rows + d.n_gram - 1) / d.n_gram;
    const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    double acc = 0.0, av = 0.0;
    for (int c0 = r0; c0 < r1; c0 += GRAM_ROWS) {
        const int n = min(GRAM_ROWS, r1 - c0);
        double vs[3], vz[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = t + q * 1024;
            vs[q] = e < n * NBP ? d.Spb[(size_t)c0 * NBP + e] : 0.0;
            vz[q] = e < n * NBP ? d.Zb[(size_t)c0 * NBP + e] : 0.0;
        }
        const double xv = t < n ? d.x0[c0 + t] : 0.0;
        __syncthreads();            // the previous chunk has been consumed
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = t + q * 1024;
            if (e < GRAM_ROWS * NBP) { sS[e] = vs[q]; sZ[e] = vz[q]; }
        }
        if (t < GRAM_ROWS) sx[t] = xv;
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            const double sv = sS[i * NBP + a];
            acc += sv * sZ[i * NBP + b];
            if (b == 0) av += sv * sx[i];
        }
    }
    double *out = d.part_g + (size_t)blockIdx.x * (NBP * NBP + NBP);
    out[t] = acc;
    if (b == 0) out[NBP * NBP + a] = av;
}

// border system: T = S_bb + D_b^2 - S_pb^T Zb, rb = -g_b^ - S_pb^T x0, Cholesky solve -> delta_b (one wave
// does the factorisation out of LDS)
__global__ __launch_bounds__(1024) void k_border_solve(Dev d) {
    State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double T[NBP * (NBP + 1)];
    __shared__ double rb[NBP];
    const int t = threadIdx.x, a = t / NBP, b = t - a * NBP, nb = d.nb;
    {
        // partial sums in list order, GF loads in flight at a time (r04: 8 -- eight dependent trips for the 64 partials, most of
        // this launch's 34 us; the order of the additions is the same); the transpose comes from LDS
        constexpr int GF = 16;
        __shared__ double G[NBP * (NBP + 1)];
        double g = 0.0, gv = 0.0;
        for (int q0 = 0; q0 < d.n_gram; q0 += GF) {
            double x[GF], y[GF];
#pragma unroll
            for (int u = 0; u < GF; ++u) x[u] = q0 + u < d.n_gram ? d.part_g[(size_t)(q0 + u) * (NBP * NBP + NBP) + a * NBP + b] : 0.0;
            if (b == 0) {
#pragma unroll
                for (int u = 0; u < GF; ++u) y[u] = q0 + u < d.n_gram ? d.part_g[(size_t)(q0 + u) * (NBP * NBP + NBP) + NBP * NBP + a] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < GF; ++u) g += x[u];
            if (b == 0) {
#pragma unroll
                for (int u = 0; u < GF; ++u) gv += y[u];
            }
        }
        G[a * (NBP + 1) + b] = g;
        __syncthreads();
        const double gt = G[b * (NBP + 1) + a];
        // symmetrised against rounding (only the lower triangle is read below)
        double v = 0.5 * ((d.bsys[BS_SBB + a * NBP + b] - g) + (d.bsys[BS_SBB + b * NBP + a] - gt));
        if (a == b) {
            if (a < nb) {
                const double s = d.bsys[BS_S + a], s2 = s * s;
                const double radius = st.opt.strategy ? 1.0 / st.mu : st.radius;
                v += fmin(fmax(d.bsys[BS_H + a] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (radius * s2);
            } else {
                v = 1.0;
            }
        } else if (a >= nb || b >= nb) {
            v = 0.0;
        }
        T[a * (NBP + 1) + b] = v;
        if (b == 0) rb[a] = a < nb ? -d.bsys[BS_RHS + a] - gv : 0.0;
    }
    __syncthreads();
    if (t >= 64) return;
    // One wave, lane = row, the row in REGISTERS: a right-looking Cholesky whose pivot column travels by v_readlane
    // (the first version kept T in LDS and paid three LDS operations per multiply-add and two fences per column: 88 us
    // for 32 columns).  Same arithmetic, same order: entry (i, c) loses l_ij l_cj for j ascending.
    const int i = t & (NBP - 1);   // lanes >= NBP mirror a row and are ignored
    double row[NBP];
#pragma unroll
    for (int c = 0; c < NBP; ++c) row[c] = T[i * (NBP + 1) + c];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const double piv = bcast(row[j], j);
        if (!(piv > 0.0) || !isfinite(piv)) bad = true;
        const double rs = 1.0 / sqrt(piv);
        const double lij = row[j] * rs;       // meaningful on lanes i >= j
        row[j] = lij;
#pragma unroll
        for (int c = j + 1; c < NBP; ++c) row[c] -= lij * bcast(lij, c);      // meaningful on lanes i >= c
    }
    if (bad) { if (t == 0) st.step_failed = 1; return; }
    if (t < NBP) {
#pragma unroll
        for (int c = 0; c < NBP; ++c) T[i * (NBP + 1) + c] = row[c];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // forward and backward substitution (lane i owns rb[i])
    double v = t < NBP ? rb[i] : 0.0;
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const double yj = bcast(v, j) / bcast(row[j], j);
        if (t == j) v = yj;
        else if (t > j && t < NBP) v -= row[j] * yj;
    }
    for (int j = NBP - 1; j >= 0; --j) {
        const double xj = bcast(v, j) / T[j * (NBP + 1) + j];
        if (t == j) v = xj;
        else if (t < j) v -= T[j * (NBP + 1) + t] * xj;
    }
    if (t < NBP) d.bsys[BS_DB + t] = t < nb ? v : 0.0;
}

// dp = x0 - Zb db
__global__ __launch_bounds__(256) void k_border_apply(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.nf_pad * 6) return;
    double v = d.x0[i];
    for (int c = 0; c < d.nb; ++c) v -= d.Zb[(size_t)i * NBP + c] * d.bsys[BS_DB + c];
    if (d.cb && i >= d.nchain * 6 && i < d.nfree * 6) v = d.bsys[BS_DB + (i - d.nchain * 6)];     // closure border: these poses ARE the border
    d.x0[i] = v;
}

// sharded problems: the Jacobi scale of the border columns from diag H_bb summed over the ranks (single GPU:
// k_ph_border_reduce does it where it forms the sums)
__global__ void k_border_scale(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.iteration != 0) return;
    const int c = threadIdx.x;
    if (c < d.nb) d.bsys[BS_S + c] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(d.bsys[BS_H + c])) : 1.0;
}
void launch_border_scale(Launcher &L, const Dev &d) { LAUNCH(KC_SMALL, k_border_scale, dim3(1), dim3(64), 0, d); }

static constexpr size_t SH_FWD = (size_t)BD * LDT * sizeof(double);
static constexpr size_t SH_UPD = (size_t)(2 * BD * BD + 2 * BD * NBP) * sizeof(double);
static constexpr size_t SH_BWD = (size_t)(2 * BD * LDT + 2 * BD * NBP) * sizeof(double);

int configure_border() {
    if (hipFuncSetAttribute((const void *)k_bcrm_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_FWD) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcrm_upd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_UPD) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcrm_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_BWD) != hipSuccess) return -1;
    return 0;
}

// Z = S_pp^-1 B for the NBP columns in d.Spb, through the level factors launch_bcr has left in place
void launch_bcr_multi_rhs(Launcher &L, const Dev &d) {
    const int nl = d.n_levels;
    hipMemcpyAsync(d.lev[0].B, d.Spb, (size_t)d.Nsb * BD * NBP * sizeof(double), hipMemcpyDeviceToDevice, L.stream);
    for (int l = 0; l + 1 < nl; ++l) {
        const int n = d.lev[l].n;
        LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(n / 2), dim3(MR_THREADS), SH_FWD, d, l, 0, 0);
        LAUNCH(KC_BORDER, k_bcrm_upd, dim3((n + 1) / 2), dim3(UPD_THREADS), SH_UPD, d, l, 0);
    }
    LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(1), dim3(MR_THREADS), SH_FWD, d, nl - 1, 1, 0);
    LAUNCH(KC_BORDER, k_bcrm_bwd, dim3(1), dim3(MR_THREADS), SH_BWD, d, nl - 1, 1, 0, 0);
    for (int l = nl - 2; l >= 0; --l)
        LAUNCH(KC_BORDER, k_bcrm_bwd, dim3(d.lev[l].n / 2), dim3(MR_THREADS), SH_BWD, d, l, 0, 0, 0);
}

// after launch_bcr: x0 = S_pp^-1 (-g_p^) and the level factors are in place
void launch_border_solve(Launcher &L, const Dev &d) {
    // rode: launch_bcr has taken the border columns through the forward part already (yB = G^-1 B per block and level
    // sit where k_bcrm_fwd would have left them); only the backward part is left
    const bool rode = bcr_border_rides(d);
    if (d.pcr.level >= 0 && d.pcr.keep) {
        // the solve ran the parallel plan: the border columns follow through the kept factors of every step
        const int n = d.pcr.n;
        if (!rode) {
            hipMemcpyAsync(d.pcr.Bb, d.Spb, (size_t)n * BD * NBP * sizeof(double), hipMemcpyDeviceToDevice, L.stream);
            for (int q = 0; q < d.pcr.steps; ++q) {
                LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(n), dim3(MR_THREADS), SH_FWD, d, q, 0, 2);
                LAUNCH(KC_BORDER, k_bcrm_upd, dim3(n), dim3(UPD_THREADS), SH_UPD, d, q, 2);
            }
            LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(n), dim3(MR_THREADS), SH_FWD, d, d.pcr.steps, 1, 2);
        }
        LAUNCH(KC_BORDER, k_bcrm_bwd, dim3(n), dim3(MR_THREADS), SH_BWD, d, 0, 1, 2, rode && bcr_rhs_rides_in_bwd(d) ? 1 : 0);
        launch_border_finish(L, d);
        return;
    }
    const int nl = d.n_levels;
    if (!rode) {
        hipMemcpyAsync(d.lev[0].B, d.Spb, (size_t)d.Nsb * BD * NBP * sizeof(double), hipMemcpyDeviceToDevice, L.stream);
        for (int l = 0; l + 1 < nl; ++l) {
            const int n = d.lev[l].n;
            LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(n / 2), dim3(MR_THREADS), SH_FWD, d, l, 0, 0);
            LAUNCH(KC_BORDER, k_bcrm_upd, dim3((n + 1) / 2), dim3(UPD_THREADS), SH_UPD, d, l, 0);
        }
        LAUNCH(KC_BORDER, k_bcrm_fwd, dim3(1), dim3(MR_THREADS), SH_FWD, d, nl - 1, 1, 0);
    }
    LAUNCH(KC_BORDER, k_bcrm_bwd, dim3(1), dim3(MR_THREADS), SH_BWD, d, nl - 1, 1, 0, 0);
    for (int l = nl - 2; l >= 0; --l)
        LAUNCH(KC_BORDER, k_bcrm_bwd, dim3(d.lev[l].n / 2), dim3(MR_THREADS), SH_BWD, d, l, 0, 0, 0);
    launch_border_finish(L, d);
}

// with Z = S_pp^-1 S_pb in Zb: the border system, its solve, and the correction of the pose step
void launch_border_finish(Launcher &L, const Dev &d) {
    LAUNCH(KC_BORDER, k_border_gram, dim3(d.n_gram), dim3(1024), 0, d);
    LAUNCH(KC_SMALL, k_border_solve, dim3(1), dim3(1024), 0, d);
    LAUNCH(KC_SMALL, k_border_apply, dim3((d.nf_pad * 6 + 255) / 256), dim3(256), 0, d);
}

}  // namespace ssba

// Block cyclic reduction (BCR) of the block-tridiagonal reduced camera system S on gfx950.
//
// S has Nsb diagonal blocks D_i (BD x BD, BD = 72 = 12 poses) and couplings L_i = S[i, i-1].
// Level l eliminates its odd blocks (all in parallel, one workgroup each):
//   k_bcr_factor  (odd i):  D_i = G G^T ; YL = G^-1 L_i ; YU = G^-1 L_{i+1}^T ; yr = G^-1 r_i
//   k_bcr_reduce  (even e): D' = D_e - YU(e-1)^T YU(e-1) - YL(e+1)^T YL(e+1)
//                           L' = -YU(e-1)^T YL(e-1) ;  r' = r_e - YU(e-1)^T yr(e-1) - YL(e+1)^T yr(e+1)
//   k_bcr_backsub (odd i):  x_i = G^-T (yr - YL x_{i-1} - YU x_{i+1})      (top-down)
// and recurses on the even blocks; the last level (one block) is a plain Cholesky solve.
// From the first level with <= PCR_MAX_BLOCKS blocks on, the same kernels run PARALLEL cyclic reduction (which = 2,
// PcrPlan): at stride 2^k every block is factored and folds in both neighbours at +-2^k, so after log2(n) steps the
// blocks are decoupled and one factor + solve over all of them finishes -- no back-substitution sweep over those levels.
// G (with 1/G_kk on its diagonal), YL and yr overwrite D_i, L_i and r_i in place.
// Coupling blocks with an EVEN index are only ever consumed transposed (as L_{i+1}^T of the
// odd block before them), so they are stored transposed at every level: the factor kernel
// then stages all three operands with plain row-major copies (no LDS transpose).
//
// The factor kernel is latency bound (72 dependent pivots); it keeps every 6x6 tile of
// [D | L | U^T | r] in the registers of one lane for the whole factorisation ("owner
// computes") and stages only the current block column / block row through LDS, so LDS
// carries operands (2 reads per 6 FMAs) and never accumulators.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "ssba_launch.h"
#include "ssba_types.h"

namespace ssba {

constexpr int NB = 6;                       // tile edge
constexpr int NBLK = BD / NB;               // 12 block rows
constexpr int NCB = (2 * BD + 1 + NB - 1) / NB;   // 25 column blocks of the right-hand sides
constexpr int NT_A = NBLK * (NBLK - 1) / 2; // 66 strictly-lower tiles of D (diagonal tiles: see below)
constexpr int NT = NT_A + NBLK * NCB;       // 366 register-resident tiles
constexpr int TILE_THREADS = 384;           // waves 0..5 own tiles
constexpr int FACT_THREADS = 448;           // wave 6 factors the diagonal tiles one step ahead

// tile table, sorted by the step in which a tile becomes final so that whole waves retire early
__constant__ uint32_t c_tile[NT];       // type | rb << 8 | cb << 16; type 0 = tile of D (rb >= cb), 1 = tile of the right-hand sides

int upload_bcr_tables(hipStream_t s) {
    struct T { int fin, type, rb, cb; };
    std::vector<T> v;
    for (int rb = 0; rb < NBLK; ++rb)
        for (int cb = 0; cb < rb; ++cb) v.push_back({cb, 0, rb, cb});
    for (int rb = 0; rb < NBLK; ++rb)
        for (int cb = 0; cb < NCB; ++cb) v.push_back({rb, 1, rb, cb});
    std::stable_sort(v.begin(), v.end(), [](const T &a, const T &b) { return a.fin < b.fin; });
    uint32_t packed[NT];
    for (int i = 0; i < NT; ++i) packed[i] = (uint32_t)v[i].type | ((uint32_t)v[i].rb << 8) | ((uint32_t)v[i].cb << 16);
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tile), packed, sizeof packed, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}

// 1/sqrt(a) and 1/a to fp64 accuracy: hardware estimate + Newton steps.  IEEE sqrt / divide
// cost ~150 / ~110 dependent cycles on gfx950 (tools/fp64_calib.hip), the estimates ~20.
__device__ __forceinline__ double rsqrt_nr(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}
// one Newton step: the hardware estimate is good to ~2^-26, so this reaches ~2^-50, plenty
// for elimination multipliers (the outputs are scaled by the two-step rsqrt above)
__device__ __forceinline__ double rcp_nr(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}

#ifdef SSBA_STAMPS
#define STAMP(base, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) d.dbg[(base) + (i)] = clock64(); } while (0)
#define STAMPT(tid, base, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == (tid)) d.dbg[(base) + (i)] = clock64(); } while (0)
#else
#define STAMP(base, i) do { } while (0)
#define STAMPT(tid, base, i) do { } while (0)
#endif

constexpr int LDA = BD + 2;            // 74: LDS row stride of D / G (16-byte aligned rows)
constexpr int RCOLS = NCB * NB;        // 150 right-hand-side columns incl. padding
constexpr int LDR = RCOLS + 2;         // 152
constexpr int FACT_LDS_DOUBLES = BD * LDA + BD * LDR;

// One workgroup per odd block.  Everything is staged into LDS with one coalesced sweep,
// each lane then owns one 6x6 tile of [D | L | U^T | r] in registers for the whole
// factorisation; finished tiles go back to LDS (they are the operands of later updates and
// the kernel's output), and one coalesced sweep writes G, YL, YU, yr.
__global__ __launch_bounds__(FACT_THREADS) void k_bcr_factor(Dev d, int lev, int top, int which) {
    State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    extern __shared__ __align__(16) double lds[];
    double *A = lds;                 // BD x LDA
    double *R = lds + BD * LDA;      // BD x LDR : [ L_i (72) | L_{i+1}^T (72) | r_i | pad ]
    __shared__ int sBad;
    __shared__ double sDiag[36];
    // operands and destinations.  Cyclic reduction (which = 0 / 1): odd block blk of level `lev`, everything in place.
    // Parallel cyclic reduction (which = 2): block blockIdx.x of the plan's level at stride 2^lev; D and r stay (the
    // reduce kernel updates them in place), the products go to the plan's buffers; the last step (top) is in place.
    const double *Dg, *Lg, *Ug, *rin;
    double *oD, *oYL, *oYU, *orr;
    bool hasL, hasU, trL = false, trU = false;
    double *saveU = nullptr;      // pinned plans: own copy of the coupling to the pinned last block (see PcrPlan)
    if (which >= 2) {
        const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
        const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[d.pcr.level];
        const int blk = blockIdx.x, s = 1 << lev, last = B.n - 1;      // top: lev = steps, so 2^lev >= n
        if (P.pin0 && lev == 0 && blk == 1) {
            // from the next step on block 1 carries its coupling to the pinned block 0 (odd index: stored untransposed)
            const double2 *s2 = reinterpret_cast<const double2 *>(B.L + (size_t)BD * BD);
            double2 *d2 = reinterpret_cast<double2 *>(P.Lbuf + (size_t)BD * BD);
            for (int e = threadIdx.x; e < BD * BD / 2; e += FACT_THREADS) d2[e] = s2[e];
        }
        if ((P.pin0 && blk == 0) || (P.pin1 && blk == last)) return;    // a pinned block is never eliminated
        hasL = blk - s >= 0 || (P.pin0 && blk > 0);
        hasU = blk + s <= last || (P.pin1 && blk < last);
        Dg = B.D + (size_t)blk * BD * BD;
        rin = B.r + (size_t)blk * BD;
        if (lev == 0) {      // the level's own couplings: even-indexed ones are stored transposed
            Lg = B.L + (size_t)blk * BD * BD;
            Ug = B.L + (size_t)(hasU ? blk + 1 : blk) * BD * BD;
            trL = trU = (blk & 1) == 0;
        } else {
            Lg = P.Lbuf + (size_t)blk * BD * BD;
            Ug = (P.pin1 && blk + s > last) ? P.Ubuf + (size_t)blk * BD * BD : P.LbufT + (size_t)(hasU ? blk + s : blk) * BD * BD;
        }
        // the pinned last block folds this block in now and moves on: keep the coupling to it
        if (P.pin1 && blk + s == last) saveU = P.Ubuf + (size_t)blk * BD * BD;
        const size_t so = P.keep ? (size_t)lev * B.n + blk : (size_t)blk;      // per-step slots when the border follows
        oD = top ? B.D + (size_t)blk * BD * BD : (P.keep ? P.Gs + so * BD * BD : nullptr);
        oYL = hasL ? P.YL + so * BD * BD : nullptr;
        oYU = hasU ? P.YU + so * BD * BD : nullptr;
        orr = top ? B.r + (size_t)blk * BD : P.yr + (size_t)blk * BD;
    } else {
        const BcrLevel &L = d.lev[lev];
        const int blk = top ? 0 : 2 * blockIdx.x + 1;
        hasL = !top;
        hasU = !top && (blk + 1 < L.n);
        Dg = L.D + (size_t)blk * BD * BD;
        Lg = L.L + (size_t)blk * BD * BD;
        Ug = L.L + (size_t)(hasU ? blk + 1 : blk) * BD * BD;
        rin = L.r + (size_t)blk * BD;
        oD = L.D + (size_t)blk * BD * BD;
        oYL = hasL ? L.L + (size_t)blk * BD * BD : nullptr;
        oYU = top ? nullptr : L.YU + (size_t)blockIdx.x * BD * BD;
        orr = L.r + (size_t)blk * BD;
    }
    const int t = threadIdx.x;
    const bool has_tile = t < NT;
    const uint32_t tile = has_tile ? c_tile[t] : 2u;       // one load, consumed after the bulk load below is under way
    if (t == 0) sBad = 0;
    STAMP(lev * 64, 0);

    // ---- bulk load (all global reads issued back to back) --------------------------
    {
        const double2 *D2 = reinterpret_cast<const double2 *>(Dg);
        const double2 *L2 = reinterpret_cast<const double2 *>(Lg);
        const double2 *U2 = reinterpret_cast<const double2 *>(Ug);
        // all 18 reads of a lane are in flight before the first LDS store (a rolled loop waits for every round trip)
        constexpr int NLD = (BD * BD / 2 + FACT_THREADS - 1) / FACT_THREADS;
        double2 dvv[NLD], lvv[NLD], uvv[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = t + q * FACT_THREADS;
            const bool in = e < BD * BD / 2;
            dvv[q] = in ? D2[e] : make_double2(0.0, 0.0);
            lvv[q] = (in && hasL) ? L2[e] : make_double2(0.0, 0.0);
            uvv[q] = (in && hasU) ? U2[e] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = t + q * FACT_THREADS;
            if (e >= BD * BD / 2) continue;
            const int r = (2 * e) / BD, c = 2 * e - r * BD;
            const double2 dv = dvv[q], lv = lvv[q], uv = uvv[q];
            A[r * LDA + c] = dv.x; A[r * LDA + c + 1] = dv.y;
            if (!trL) { R[r * LDR + c] = lv.x; R[r * LDR + c + 1] = lv.y; }
            else { R[c * LDR + r] = lv.x; R[(c + 1) * LDR + r] = lv.y; }
            if (!trU) { R[r * LDR + BD + c] = uv.x; R[r * LDR + BD + c + 1] = uv.y; }   // even block: stored as L^T
            else { R[c * LDR + BD + r] = uv.x; R[(c + 1) * LDR + BD + r] = uv.y; }
            if (saveU) {
                if (!trU) reinterpret_cast<double2 *>(saveU)[e] = uv;
                else { saveU[c * BD + r] = uv.x; saveU[(c + 1) * BD + r] = uv.y; }
            }
        }
        if (t < BD) {
            R[t * LDR + 2 * BD] = rin[t];
#pragma unroll
            for (int c = 2 * BD + 1; c < LDR; ++c) R[t * LDR + c] = 0.0;
        }
    }
    const int type = (int)(tile & 255u), rb = (int)((tile >> 8) & 255u), cb = (int)(tile >> 16);
    __syncthreads();
    STAMP(lev * 64, 1);

    // Per-lane view of the tile so that D tiles and right-hand-side tiles run ONE code path
    // (no divergence inside a wave): the register tile T[x][z] is the D tile itself
    // (x = row, z = column) or the TRANSPOSE of the right-hand-side tile (x = column, z = row).
    //   own tile:        T[x][z]  <->  own[x*osx + z*osz]
    //   panel operand P: P[x][q]  (type 0: G[rb][kb] rows;      type 1: Y[kb][cb] columns)
    //   panel operand Q: Q[z][q]  (type 0: G[cb][kb] rows;      type 1: G[rb][kb] rows)
    //   update:          T[x][z] -= sum_q P[x][q] Q[z][q]
    //   finalisation:    T[x][:] <- solve against the diagonal tile, identical recurrence
    double *own;
    int osx, osz;
    if (type == 1) { own = R + (rb * 6) * LDR + cb * 6; osx = 1; osz = LDR; }
    else { own = A + (rb * 6) * LDA + cb * 6; osx = LDA; osz = 1; }
    const int fin = (type == 1) ? rb : cb;          // step at which this tile becomes final
    double acc[36];
    if (has_tile) {
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int z = 0; z < 6; ++z) acc[6 * x + z] = own[x * osx + z * osz];
    }
    // Diagonal tiles have no owner: wave 6 rebuilds tile (kd,kd) left-looking from the finished
    // panels of block row kd (they live in LDS), factors it (6 dependent pivots on reciprocals,
    // square roots only scale the outputs) and stores L with 1/L_jj on the diagonal.  It does so
    // for step kb+1 while waves 0..5 run the trailing update of step kb, which takes the
    // ~1.2k-cycle pivot chain off the critical path.
    // The left-looking sum over the columns of panels 0 .. kd-2 does not need panel kd-1: wave 6 forms it ahead of time
    // (diag_pre, while waves 0..5 finalise panel kd-1), so that only the six newest columns are left on the serial chain
    // diagonal tile -> forward substitution of its panel -> next diagonal tile, which is what a launch waits for.
    double dpart = 0.0;
    auto diag_pre = [&](int kd) {
        const int e = t - TILE_THREADS;
        if (e < 36) {
            const int i = e / 6, j = e - i * 6;
            const double *ri = A + (kd * 6 + i) * LDA, *rj = A + (kd * 6 + j) * LDA;
            const int cend = (kd - 1) * 6;
            double v0 = ri[kd * 6 + j], v1 = 0.0, v2 = 0.0, v3 = 0.0;
            for (int c = 0; c + 4 <= cend; c += 4) {
                const double2 a0 = *reinterpret_cast<const double2 *>(ri + c), a1 = *reinterpret_cast<const double2 *>(ri + c + 2);
                const double2 b0 = *reinterpret_cast<const double2 *>(rj + c), b1 = *reinterpret_cast<const double2 *>(rj + c + 2);
                v0 -= a0.x * b0.x; v1 -= a0.y * b0.y; v2 -= a1.x * b1.x; v3 -= a1.y * b1.y;
            }
            if (cend > 0 && (cend & 2)) {   // cend is even: a remainder of two columns when kd - 1 is odd
                const int c = cend - 2;
                v0 -= ri[c] * rj[c]; v1 -= ri[c + 1] * rj[c + 1];
            }
            dpart = (v0 + v1) + (v2 + v3);
        }
    };
    auto diag_step = [&](int kd) {
        const int e = t - TILE_THREADS;
        if (e < 36) {
            const int i = e / 6, j = e - i * 6;
            double v0 = dpart, v1 = 0.0;
            if (kd > 0) {       // the columns of panel kd - 1
                const double2 *ri = reinterpret_cast<const double2 *>(A + (kd * 6 + i) * LDA + (kd - 1) * 6);
                const double2 *rj = reinterpret_cast<const double2 *>(A + (kd * 6 + j) * LDA + (kd - 1) * 6);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double2 a = ri[q], b = rj[q];
                    v0 -= a.x * b.x; v1 -= a.y * b.y;
                }
            }
            sDiag[e] = v0 + v1;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (e == 0) {
            double a[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) a[i][j] = sDiag[6 * i + j];
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double sv = a[j][j];
                if (!(sv > 0.0) || !isfinite(sv)) { bad = true; sv = 1.0; a[j][j] = 1.0; }
                const double rc = rcp_nr(sv);
#pragma unroll
                for (int c = j + 1; c < 6; ++c) {
                    const double w = a[c][j] * rc;
#pragma unroll
                    for (int i = c; i < 6; ++i) a[i][c] -= a[i][j] * w;
                }
            }
            if (bad) sBad = 1;
            // unscaled columns (L D^1/2 form): the 36 lanes scale them in parallel below
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) sDiag[6 * i + j] = a[i][j];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (e < 36) {
            const int i = e / 6, j = e - i * 6;
            if (j <= i) {
                // G tile: L = (unscaled column j) / sqrt(pivot j) below the diagonal, 1/L_jj ON it
                const double rs = rsqrt_nr(sDiag[7 * j]);
                A[(kd * 6 + i) * LDA + kd * 6 + j] = (j < i) ? sDiag[6 * i + j] * rs : rs;
            }
        }
    };
    if (t >= TILE_THREADS) { diag_pre(0); diag_step(0); }

    for (int kb = 0; kb < NBLK; ++kb) {
        __syncthreads();                      // (1) L(kb) is in LDS
        STAMP(lev * 64, 2 + 3 * kb);
        if (sBad) {
            if (t == 0) st.step_failed = 1;
            return;
        }
        // (2) block column kb of G and block row kb of Y become final: forward substitution of
        //     each line T[x][:] against the diagonal tile (right-looking, 6 independent lines)
        if (t >= TILE_THREADS && kb + 1 < NBLK) diag_pre(kb + 1);
        if (has_tile && fin == kb) {
            double l[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) l[i][j] = A[(kb * 6 + i) * LDA + kb * 6 + j];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    const double v = acc[6 * x + c] * l[c][c];
                    acc[6 * x + c] = v;
#pragma unroll
                    for (int c2 = c + 1; c2 < 6; ++c2) acc[6 * x + c2] -= v * l[c2][c];
                }
            }
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int z = 0; z < 6; ++z) own[x * osx + z * osz] = acc[6 * x + z];
        }
        __syncthreads();
        STAMP(lev * 64, 3 + 3 * kb);
        // (3) trailing update of every tile that is not final yet
        STAMPT(340, 1024 + lev * 64, 2 * kb);
        STAMPT(384, 2048 + lev * 64, 2 * kb);
        if (has_tile && fin > kb) {
            const double *Pp, *Qp;
            int psx, psq;
            if (type == 1) { Pp = R + (kb * 6) * LDR + cb * 6; psx = 1; psq = LDR; }
            else { Pp = A + (rb * 6) * LDA + kb * 6; psx = LDA; psq = 1; }
            Qp = A + ((type == 1 ? rb : cb) * 6) * LDA + kb * 6;
            double pv[36], qv[36];
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    pv[6 * x + q] = Pp[x * psx + q * psq];
                    qv[6 * x + q] = Qp[x * LDA + q];
                }
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int z = 0; z < 6; ++z) {
                    double v = acc[6 * x + z];
#pragma unroll
                    for (int q = 0; q < 6; ++q) v -= pv[6 * x + q] * qv[6 * z + q];
                    acc[6 * x + z] = v;
                }
        }
        else if (t >= TILE_THREADS && kb + 1 < NBLK) diag_step(kb + 1);
        STAMPT(340, 1024 + lev * 64, 2 * kb + 1);
        STAMPT(384, 2048 + lev * 64, 2 * kb + 1);
        // the next step's first barrier orders these LDS reads before the next panel writes
        STAMP(lev * 64, 4 + 3 * kb);
    }
    __syncthreads();
    STAMP(lev * 64, 40);
    // ---- bulk store ----------------------------------------------------------------
    {
        double2 *D2 = reinterpret_cast<double2 *>(oD);
        double2 *L2 = reinterpret_cast<double2 *>(oYL);
        double2 *U2 = reinterpret_cast<double2 *>(oYU);
        for (int e = t; e < BD * BD / 2; e += FACT_THREADS) {
            const int r = (2 * e) / BD, c = 2 * e - r * BD;
            if (oD) D2[e] = make_double2(A[r * LDA + c], A[r * LDA + c + 1]);
            if (oYL) L2[e] = make_double2(R[r * LDR + c], R[r * LDR + c + 1]);
            if (oYU) U2[e] = make_double2(R[r * LDR + BD + c], R[r * LDR + BD + c + 1]);
        }
        if (t < BD) orr[t] = R[t * LDR + 2 * BD];
    }
    STAMP(lev * 64, 41);
}

// C -= A^T B for BD x BD operands: 3-way split over k (432 lanes = 3 x 144 tiles of 6x6),
// partial sums combined through LDS in a fixed order.
constexpr int RED_THREADS = 448;
constexpr int KSPLIT = 3;
constexpr int KCH = BD / KSPLIT;   // 24

// copy of one block, all reads of a lane in flight before its first store (nthreads is a launch constant: >= 448)
__device__ __forceinline__ void stage_block(double *dst, const double *__restrict__ src, int nthreads) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    constexpr int NLD = (BD * BD / 2 + 447) / 448;     // 6
    double2 v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * nthreads;
        v[q] = e < BD * BD / 2 ? s2[e] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = threadIdx.x + q * nthreads;
        if (e < BD * BD / 2) d2[e] = v[q];
    }
}

__device__ __forceinline__ void tile_mac_k(double *acc, const double *sA, const double *sB, int k0, int k1, int tr, int tc) {
    for (int k = k0; k < k1; ++k) {
        double a[6], b[6];
        const double2 *pa = reinterpret_cast<const double2 *>(sA + k * BD + tr * 6);   // 16-byte aligned
        const double2 *pb = reinterpret_cast<const double2 *>(sB + k * BD + tc * 6);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double2 va = pa[i], vb = pb[i];
            a[2 * i] = va.x; a[2 * i + 1] = va.y;
            b[2 * i] = vb.x; b[2 * i + 1] = vb.y;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[6 * i + j] += a[i] * b[j];
    }
}
__device__ __forceinline__ void tile_mac(double *acc, const double *sA, const double *sB, int g, int tr, int tc) {
    tile_mac_k(acc, sA, sB, g * KCH, (g + 1) * KCH, tr, tc);
}

// grid = (n_next, 2): y = 0 -> D' and r' ; y = 1 -> L'.  Both operand blocks are staged
// into LDS up front (one global round trip), r' is computed from the staged copies.
__global__ __launch_bounds__(RED_THREADS) void k_bcr_reduce(Dev d, int lev, int which) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    extern __shared__ __align__(16) double lds[];
    double *sA = lds, *sB = lds + BD * BD;
    __shared__ double sya[BD], syb[BD];
    const int t = threadIdx.x;
    if (which >= 2) {
        // parallel cyclic reduction, stride s = 2^lev: block e folds in BOTH neighbours e -+ s (D and r in place),
        //   D_e -= YU(e-s)^T YU(e-s) + YL(e+s)^T YL(e+s) ;  r_e -= YU(e-s)^T yr(e-s) + YL(e+s)^T yr(e+s)
        //   L'_e = -YU(e-s)^T YL(e-s)   (coupling to e - 2s, or to the pinned first block), stored untransposed and transposed
        //   U'_e = -YL(e+s)^T YU(e+s)   only where it is a coupling to the pinned last block (blockIdx.y = 2; see PcrPlan)
        const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
        const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[d.pcr.level];
        const int e = blockIdx.x, s = 1 << lev, prev = e - s, next = e + s, last = B.n - 1;
        const int lo = P.pin0 ? 1 : 0, hi = P.pin1 ? last - 1 : last;       // the blocks that get eliminated
        const bool hasPrev = prev >= lo, hasNext = next <= hi;
        const size_t so = P.keep ? (size_t)lev * B.n : 0;
        const bool act = t < KSPLIT * 144;
        const int g = t / 144, tt = t - g * 144;
        const int tr = tt / 12, tc = tt - tr * 12;
        double acc[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = 0.0;
        double *out, *outT = nullptr;
        double rbase = 0.0;
        double dbase[36];
        if (blockIdx.y == 0) {
            // D_e -= YU^T YU + YL^T YL is symmetric: the 78 upper 6x6 tiles, a 4-way split over k (312 lanes), mirrored stores
            if (!hasPrev && !hasNext) return;
            out = B.D + (size_t)e * BD * BD;
            constexpr int SYM_SPLIT = 4, SYM_KCH = BD / SYM_SPLIT, SYM_TILES = 78;
            const bool sact = t < SYM_SPLIT * SYM_TILES;
            const int sg = t / SYM_TILES, stt = t - sg * SYM_TILES;
            int sr = 0, rem = stt;
            while (rem >= 12 - sr) { rem -= 12 - sr; ++sr; }
            const int sc = sr + rem;
            if (sact && sg == 0) {      // the tile of D this lane updates at the end: fetched now, under the products
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) dbase[6 * i + j] = out[(size_t)(sr * 6 + i) * BD + sc * 6 + j];
            }
            if (hasPrev) stage_block(sA, P.YU + (so + prev) * BD * BD, RED_THREADS);
            if (hasNext) stage_block(sB, P.YL + (so + next) * BD * BD, RED_THREADS);
            if (t < BD) {
                sya[t] = hasPrev ? P.yr[(size_t)prev * BD + t] : 0.0;
                syb[t] = hasNext ? P.yr[(size_t)next * BD + t] : 0.0;
                rbase = B.r[(size_t)e * BD + t];
            }
            __syncthreads();
            if (sact) {
                if (hasPrev) tile_mac_k(acc, sA, sA, sg * SYM_KCH, (sg + 1) * SYM_KCH, sr, sc);
                if (hasNext) tile_mac_k(acc, sB, sB, sg * SYM_KCH, (sg + 1) * SYM_KCH, sr, sc);
            }
            if (t < BD) {
                double v0 = 0.0, v1 = 0.0;
                if (hasPrev) for (int k = 0; k < BD; ++k) v0 += sA[k * BD + t] * sya[k];
                if (hasNext) for (int k = 0; k < BD; ++k) v1 += sB[k * BD + t] * syb[k];
                B.r[(size_t)e * BD + t] = rbase - v0 - v1;
            }
            __syncthreads();
            double *spart = lds;      // 3 x 78 x 36 doubles = 67 392 B <= the operand area
            if (sact && sg > 0) {
#pragma unroll
                for (int i = 0; i < 36; ++i) spart[((sg - 1) * SYM_TILES + stt) * 36 + i] = acc[i];
            }
            __syncthreads();
            if (sact && sg == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const double sum = ((acc[6 * i + j] + spart[stt * 36 + 6 * i + j]) + spart[(SYM_TILES + stt) * 36 + 6 * i + j]) +
                                           spart[(2 * SYM_TILES + stt) * 36 + 6 * i + j];
                        const double v = dbase[6 * i + j] - sum;
                        out[(size_t)(sr * 6 + i) * BD + sc * 6 + j] = v;
                        if (sr != sc) out[(size_t)(sc * 6 + j) * BD + sr * 6 + i] = v;
                    }
            }
            return;
        } else if (blockIdx.y == 1) {
            // the folded block e - s must have a coupling on its far side: a block at e - 2s, or the pinned first block
            if (!hasPrev || !(prev - s >= 0 || P.pin0)) return;
            out = P.Lbuf + (size_t)e * BD * BD;
            outT = P.LbufT + (size_t)e * BD * BD;
            stage_block(sA, P.YU + (so + prev) * BD * BD, RED_THREADS);
            stage_block(sB, P.YL + (so + prev) * BD * BD, RED_THREADS);
            __syncthreads();
            if (act) tile_mac(acc, sA, sB, g, tr, tc);
            __syncthreads();
        } else {
            // e + s is folded and its far side is the pinned last block, not e + 2s: that coupling has no transposed
            // twin in the pinned block's own row, so it is computed here.  (e + 2s == last: the twin is LbufT[last].)
            if (!P.pin1 || (P.pin0 && e == 0) || !hasNext || next + s <= last) return;
            out = P.Ubuf + (size_t)e * BD * BD;
            stage_block(sA, P.YL + (so + next) * BD * BD, RED_THREADS);
            stage_block(sB, P.YU + (so + next) * BD * BD, RED_THREADS);
            __syncthreads();
            if (act) tile_mac(acc, sA, sB, g, tr, tc);
            __syncthreads();
        }
        double *part = lds;
        if (act && g > 0) {
#pragma unroll
            for (int i = 0; i < 36; ++i) part[((g - 1) * 144 + tt) * 36 + i] = acc[i];
        }
        __syncthreads();
        if (act && g == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double sum = (acc[6 * i + j] + part[tt * 36 + 6 * i + j]) + part[(144 + tt) * 36 + 6 * i + j];
                    const size_t o = (size_t)(tr * 6 + i) * BD + tc * 6 + j;
                    out[o] = -sum;
                    if (outT) outT[(size_t)(tc * 6 + j) * BD + tr * 6 + i] = -sum;
                }
        }
        return;
    }
    const BcrLevel &L = d.lev[lev];
    const BcrLevel &N = d.lev[lev + 1];
    const int m = blockIdx.x, e = 2 * m;
    if (L.pin && m == L.n / 2) {
        // pinned end of a partitioned chain (old index n-1, odd): carried over unchanged as the new last block;
        // its coupling to the new block before it is the old L[n-1] (no fill-in: old n-2 is its direct neighbour)
        const int src = L.n - 1;
        if (blockIdx.y == 0) {
            stage_block(N.D + (size_t)m * BD * BD, L.D + (size_t)src * BD * BD, RED_THREADS);
            if (t < BD) N.r[(size_t)m * BD + t] = L.r[(size_t)src * BD + t];
        } else {
            const double *sl = L.L + (size_t)src * BD * BD;    // odd index: stored untransposed
            double *dl = N.L + (size_t)m * BD * BD;
            const bool tr = (m & 1) == 0;                        // even-indexed couplings are stored transposed
            for (int i = t; i < BD * BD; i += RED_THREADS) {
                const int r = i / BD, c = i - r * BD;
                dl[tr ? c * BD + r : i] = sl[i];
            }
        }
        return;
    }
    const bool act = t < KSPLIT * 144;
    const int g = t / 144, tt = t - g * 144;
    const int tr = tt / 12, tc = tt - tr * 12;
    const bool hasPrev = e - 1 >= 0, hasNext = e + 1 < L.n && !(L.pin && e + 1 == L.n - 1);   // a pinned end is not eliminated
    const int tp = (e - 2) / 2;     // YU slot of odd block e-1
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    double *out;
    const double *base = nullptr;
    double rbase = 0.0;
    double dbase[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) dbase[i] = 0.0;
    if (blockIdx.y == 0) {
        // D' = D_e - YU^T YU - YL^T YL is symmetric: the 78 upper 6x6 tiles, a 4-way split over k, mirrored stores
        out = N.D + (size_t)m * BD * BD;
        base = L.D + (size_t)e * BD * BD;
        constexpr int SYM_SPLIT = 4, SYM_KCH = BD / SYM_SPLIT, SYM_TILES = 78;
        const bool sact = t < SYM_SPLIT * SYM_TILES;
        const int sg = t / SYM_TILES, stt = t - sg * SYM_TILES;
        int sr = 0, rem = stt;
        while (rem >= 12 - sr) { rem -= 12 - sr; ++sr; }
        const int sc = sr + rem;
        if (sact && sg == 0) {      // the tile of D this lane finishes: fetched now, under the products
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) dbase[6 * i + j] = base[(size_t)(sr * 6 + i) * BD + sc * 6 + j];
        }
        if (hasPrev) stage_block(sA, L.YU + (size_t)tp * BD * BD, RED_THREADS);
        if (hasNext) stage_block(sB, L.L + (size_t)(e + 1) * BD * BD, RED_THREADS);
        if (t < BD) {
            sya[t] = hasPrev ? L.r[(size_t)(e - 1) * BD + t] : 0.0;
            syb[t] = hasNext ? L.r[(size_t)(e + 1) * BD + t] : 0.0;
            rbase = L.r[(size_t)e * BD + t];
        }
        __syncthreads();
        if (sact) {
            if (hasPrev) tile_mac_k(acc, sA, sA, sg * SYM_KCH, (sg + 1) * SYM_KCH, sr, sc);
            if (hasNext) tile_mac_k(acc, sB, sB, sg * SYM_KCH, (sg + 1) * SYM_KCH, sr, sc);
        }
        // r' = r_e - YU(e-1)^T yr(e-1) - YL(e+1)^T yr(e+1): lanes 0..71
        if (t < BD) {
            double v0 = 0.0, v1 = 0.0;
            if (hasPrev) for (int k = 0; k < BD; ++k) v0 += sA[k * BD + t] * sya[k];
            if (hasNext) for (int k = 0; k < BD; ++k) v1 += sB[k * BD + t] * syb[k];
            N.r[(size_t)m * BD + t] = rbase - v0 - v1;
        }
        __syncthreads();
        double *spart = lds;
        if (sact && sg > 0) {
#pragma unroll
            for (int i = 0; i < 36; ++i) spart[((sg - 1) * SYM_TILES + stt) * 36 + i] = acc[i];
        }
        __syncthreads();
        if (sact && sg == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double sum = ((acc[6 * i + j] + spart[stt * 36 + 6 * i + j]) + spart[(SYM_TILES + stt) * 36 + 6 * i + j]) +
                                       spart[(2 * SYM_TILES + stt) * 36 + 6 * i + j];
                    const double v = dbase[6 * i + j] - sum;
                    out[(size_t)(sr * 6 + i) * BD + sc * 6 + j] = v;
                    if (sr != sc) out[(size_t)(sc * 6 + j) * BD + sr * 6 + i] = v;
                }
        }
        return;
    } else {
        if (m == 0) return;   // L'[0] does not exist
        out = N.L + (size_t)m * BD * BD;
        stage_block(sA, L.YU + (size_t)tp * BD * BD, RED_THREADS);
        stage_block(sB, L.L + (size_t)(e - 1) * BD * BD, RED_THREADS);
        __syncthreads();
        if (act) tile_mac(acc, sA, sB, g, tr, tc);
        __syncthreads();
    }
    // combine the k-split partials in LDS (fixed order), then out = base - sum
    double *part = lds;   // 2 x 144 x 36 doubles = 82,944 B: fits the operand area exactly
    if (act && g > 0) {
#pragma unroll
        for (int i = 0; i < 36; ++i) part[((g - 1) * 144 + tt) * 36 + i] = acc[i];
    }
    __syncthreads();
    if (act && g == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double s = (acc[6 * i + j] + part[tt * 36 + 6 * i + j]) + part[(144 + tt) * 36 + 6 * i + j];
                const size_t o = (size_t)(tr * 6 + i) * BD + tc * 6 + j;
                // an even-indexed coupling block of the next level is stored transposed
                const size_t ow = (blockIdx.y == 1 && (m & 1) == 0) ? (size_t)(tc * 6 + j) * BD + tr * 6 + i : o;
                out[ow] = -s;
            }
    }
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// x_i = G^-T (yr - YL x_{i-1} - YU x_{i+1}); x lives in d.x0 at level-0 block positions.
// G, YL, YU are staged into LDS with one coalesced sweep.
constexpr int BS_THREADS = 512;
__global__ __launch_bounds__(BS_THREADS) void k_bcr_backsub(Dev d, int lev, int top, int which) {
    const State &st = *d.st;
    const int dead = st.terminated | st.step_failed | st.dl_reuse;      // tested once the staging reads are in flight (a cold read)
    extern __shared__ __align__(16) double lds[];
    double *sG = lds, *sL = lds + BD * BD, *sU = lds + 2 * BD * BD;
    __shared__ double sv[BD], sxm[BD], sxp[BD];
    // which = 2 / 3: last step of the parallel cyclic reduction -- every block of the plan's level is decoupled from the
    // others; in a chain with pinned ends (partitioned solve) it still has its couplings to those (x known by now)
    const bool pcr = which >= 2;
    const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
    const BcrLevel &L = which == 3 ? d.slev[0] : d.lev[pcr ? d.pcr.level : lev];
    const int blk = pcr ? (int)blockIdx.x : (top ? 0 : 2 * blockIdx.x + 1);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (pcr && ((P.pin0 && blk == 0) || (P.pin1 && blk == L.n - 1))) return;
    const bool hasL = pcr ? (P.pin0 != 0) : !top;
    const bool hasU = pcr ? (P.pin1 != 0) : (!top && (blk + 1 < L.n));
    const int iL = pcr ? 0 : blk - 1, iU = pcr ? L.n - 1 : blk + 1;
    double *xb = which == 3 ? d.xsep : d.x0 + (size_t)d.chain0 * BD;     // solution at level-0 block positions
    double *xi = xb + (size_t)L.pos[blk] * BD;
    stage_block(sG, L.D + (size_t)blk * BD * BD, BS_THREADS);
    if (hasL) stage_block(sL, (pcr ? P.YL : L.L) + (size_t)blk * BD * BD, BS_THREADS);
    if (hasU) stage_block(sU, pcr ? P.YU + (size_t)blk * BD * BD : L.YU + (size_t)blockIdx.x * BD * BD, BS_THREADS);
    if (t < BD) {
        sxm[t] = hasL ? xb[(size_t)L.pos[iL] * BD + t] : 0.0;
        sxp[t] = hasU ? xb[(size_t)L.pos[iU] * BD + t] : 0.0;
        sv[t] = L.r[(size_t)blk * BD + t];
    }
    if (dead) return;
    __syncthreads();
    // v = yr - YL x_{i-1} - YU x_{i+1}: 8 waves x 9 rows, lanes stride the row
    for (int r = w * 9; r < w * 9 + 9; ++r) {
        double p = 0.0;
        if (hasL) {
            p += sL[r * BD + lane] * sxm[lane];
            if (lane < BD - 64) p += sL[r * BD + 64 + lane] * sxm[64 + lane];
        }
        if (hasU) {
            p += sU[r * BD + lane] * sxp[lane];
            if (lane < BD - 64) p += sU[r * BD + 64 + lane] * sxp[64 + lane];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
        if (lane == 0) sv[r] -= p;
    }
    __syncthreads();
    if (w != 0) return;
    // wave 0: solve G^T x = v by a column sweep from the bottom; the diagonal of G holds 1/G_kk.
    double lo = sv[lane];
    double hi = lane < BD - 64 ? sv[64 + lane] : 0.0;
#pragma unroll
    for (int k = BD - 1; k >= 64; --k) {
        const double xk = lane_bcast(hi, k - 64) * sG[k * BD + k];
        const double gl = sG[k * BD + lane];
        const double gh = (lane < k - 64) ? sG[k * BD + 64 + lane] : 0.0;
        lo -= gl * xk;
        hi = (lane == k - 64) ? xk : hi - gh * xk;
    }
#pragma unroll
    for (int k = 63; k >= 0; --k) {
        const double xk = lane_bcast(lo, k) * sG[k * BD + k];
        const double gl = (lane < k) ? sG[k * BD + lane] : 0.0;
        lo = (lane == k) ? xk : lo - gl * xk;
    }
    xi[lane] = lo;
    if (lane < BD - 64) xi[64 + lane] = hi;
}

// blocks eliminated at a level: all odd ones, except the pinned end of a partitioned chain
static int n_odd(const BcrLevel &lv, int pinned) { return pinned ? (lv.n - 1) / 2 : lv.n / 2; }

// The matrix-core kernels of ssba_bcr_mfma.hip are the production path; SSBA_BCR_LEGACY=1 selects the register-tile
// kernels above (same contract; tests compare the two).
static bool bcr_legacy() {
    static const bool v = [] { const char *e = getenv("SSBA_BCR_LEGACY"); return e && e[0] == '1'; }();
    return v;
}
// coupled: the blocks still have L / U operands (false for the decoupled last step of an unpinned plan)
static void launch_factor(Launcher &L, const Dev &d, int nblocks, int lev, int top, int which, bool coupled, bool ride = false, int solve = 0) {
    if (bcr_legacy()) LAUNCH(KC_BCR_FACTOR, k_bcr_factor, dim3(nblocks), dim3(FACT_THREADS), (size_t)FACT_LDS_DOUBLES * sizeof(double), d, lev, top, which);
    else launch_bcr_factor_mf(L, d, nblocks, lev, top, which, coupled, ride, solve);
}
static void launch_reduce(Launcher &L, const Dev &d, int nblocks, int ny, int lev, int which, bool ride = false) {
    if (bcr_legacy()) LAUNCH(KC_BCR_REDUCE, k_bcr_reduce, dim3(nblocks, ny), dim3(RED_THREADS), (size_t)2 * BD * BD * sizeof(double), d, lev, which);
    else launch_bcr_reduce_mf(L, d, nblocks, ny, lev, which, ride);
}
// The border columns (free shared blocks of config 3, closure border) go through the forward part of the solve INSIDE
// the matrix-core factor / reduce launches -- two more right-hand-side tiles -- instead of a forward + update launch per
// level afterwards (ssba_border.hip).  SSBA_BORDER_SWEEPS=1 keeps the separate sweeps (A/B, tests).
bool bcr_border_rides(const Dev &d) {
    const char *e = getenv("SSBA_BORDER_SWEEPS");        // read per call: tests switch it between handles
    return d.nb > 0 && !d.part && !d.dense && !bcr_legacy() && !(e && e[0] == '1');
}

// the decoupled last step of a plan that covers the whole chain can solve its blocks AND update their poses
static bool bcr_fused_solve(const Dev &d) {
    const char *nf = getenv("SSBA_NO_FUSED_SOLVE");
    return !d.part && d.pcr.level >= 0 && !bcr_border_rides(d) && d.nb == 0 && !bcr_legacy() && !(nf && nf[0] == '1');
}
// the fused plan (PcrFused buffers allocated: single GPU, no border columns); SSBA_NO_PCR_FUSED=1 keeps factor + reduce launches (A/B, tests)
static bool bcr_fused_steps(const Dev &d) {
    const char *e = getenv("SSBA_NO_PCR_FUSED");        // read per call: tests switch it between handles
    return d.pcrf.on && !d.part && !d.pcr.keep && d.nb == 0 && !bcr_legacy() && !(e && e[0] == '1');
}
// Border columns riding through the parallel plan: the right-hand side of the decoupled last step is solved as column NBP - 1 (a
// padding column while nb < NBP) of the border columns' backward sweep -- k_bcrm_bwd does for 32 columns what k_bcr_backsub does for
// one, lane for lane in the same order -- so k_bcr_backsub's launch (11 us) is not needed.  SSBA_NO_RHS_RIDE=1 keeps it (A/B, tests).
bool bcr_rhs_rides_in_bwd(const Dev &d) {
    const char *e = getenv("SSBA_NO_RHS_RIDE");
    return bcr_border_rides(d) && d.nb < NBP && d.pcr.level >= 0 && d.pcr.keep && !d.pcr.pin0 && !d.pcr.pin1 && !(e && e[0] == '1');
}
bool bcr_updates_poses(const Dev &d) { return bcr_fused_solve(d) && d.pcr.level == 0 && d.n_pf == 0; }

void launch_bcr(Launcher &L, const Dev &d, bool allow_pcr, bool fuse_update) {
    const size_t sh_backsub = (size_t)3 * BD * BD * sizeof(double);
    const int nl = d.n_levels;
    const bool ride = allow_pcr && bcr_border_rides(d);
    const bool spb_rides = L.spb_rides;     // k_ph_spb_assemble has written the border columns in place (launch_ph_schur of this iteration)
    L.spb_rides = false;
    if (!d.part && allow_pcr && d.pcr.level >= 0) {
        // cyclic reduction down to the plan's level, parallel cyclic reduction of what is left (no back-substitution
        // sweep over those levels: log2(n) x (factor + reduce) + one decoupled solve), back-substitution of the rest
        const int k = d.pcr.level, n = d.pcr.n;
        if (ride && !spb_rides)       // (a plan with border columns always starts at level 0: ssba_finalize)
            hipMemcpyAsync(d.pcr.Bb, d.Spb, (size_t)n * BD * NBP * sizeof(double), hipMemcpyDeviceToDevice, L.stream);
        for (int l = 0; l < k; ++l) {
            const int nn = d.lev[l].n;
            launch_factor(L, d, nn / 2, l, 0, 0, true);
            launch_reduce(L, d, (nn + 1) / 2, 2, l, 0);
        }
        const bool fsolve = bcr_fused_solve(d);
        const int solve = fsolve ? (fuse_update && bcr_updates_poses(d) ? 2 : 1) : 0;
        if (bcr_fused_steps(d)) {
            // one launch per step: the factorisation of a block also forms the Gram products the next step assembles its
            // operands from (ssba_bcr_mfma.hip, PcrFused) -- steps + 1 launches instead of 2 x steps + 1
            for (int q = 0; q < d.pcr.steps; ++q) launch_pcr_fused_step(L, d, n, q);
            launch_pcr_fused_top(L, d, n, d.pcr.steps, solve);
        } else {
        for (int q = 0; q < d.pcr.steps; ++q) {
            launch_factor(L, d, n, q, 0, 2, true, ride);
            launch_reduce(L, d, n, 2, q, 2, ride);
        }
        // the decoupled last step solves its blocks itself (matrix-core kernels, no border columns): no k_bcr_backsub launch
        launch_factor(L, d, n, d.pcr.steps, 1, 2, false, ride, solve);
        }
        if (!fsolve && !(ride && bcr_rhs_rides_in_bwd(d))) LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(n), dim3(BS_THREADS), sh_backsub, d, k, 1, 2);
        for (int l = k - 1; l >= 0; --l)
            LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(d.lev[l].n / 2), dim3(BS_THREADS), sh_backsub, d, l, 0, 0);
        return;
    }
    if (!d.part) {
        if (ride && !spb_rides) hipMemcpyAsync(d.lev[0].B, d.Spb, (size_t)d.Nsb * BD * NBP * sizeof(double), hipMemcpyDeviceToDevice, L.stream);
        for (int l = 0; l + 1 < nl; ++l) {
            const int n = d.lev[l].n;
            launch_factor(L, d, n / 2, l, 0, 0, true, ride);
            launch_reduce(L, d, (n + 1) / 2, 2, l, 0, ride);
        }
        launch_factor(L, d, 1, nl - 1, 1, 0, false, ride);
        LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(1), dim3(BS_THREADS), sh_backsub, d, nl - 1, 1, 0);
        for (int l = nl - 2; l >= 0; --l)
            LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(d.lev[l].n / 2), dim3(BS_THREADS), sh_backsub, d, l, 0, 0);
        return;
    }
    // Partitioned solve, forward part: eliminate the interior of this rank's chain.  Plain levels (the pinned end of an
    // even-length level is carried over) down to the plan's level, then parallel cyclic reduction with pinned ends: the
    // pinned rows end up holding this rank's share of the separator system, every other block its factor and its
    // couplings to the pinned ones.  launch_bcr_separators() continues after the separator exchange.
    const int k = d.pcr.level, n = d.pcr.n;
    for (int l = 0; l < k; ++l) {
        launch_factor(L, d, n_odd(d.lev[l], d.lev[l].pin), l, 0, 0, true);
        launch_reduce(L, d, d.lev[l + 1].n, 2, l, 0);
    }
    const char *fe = getenv("SSBA_NO_PCR_FUSED");
    if (d.pcrf.on && !bcr_legacy() && !(fe && fe[0] == '1')) {
        // one launch per step, with the couplings to the pinned ends kept (PcrFused::Lkeep / Ukeep); the last launch
        // factors the interior blocks with those couplings as right-hand sides and leaves the pinned rows in place
        for (int q = 0; q < d.pcr.steps; ++q) launch_pcr_fused_step(L, d, n, q, 2);
        launch_pcr_fused_top(L, d, n, d.pcr.steps, 0, 2);
        return;
    }
    for (int q = 0; q < d.pcr.steps; ++q) {
        launch_factor(L, d, n, q, 0, 2, true);
        launch_reduce(L, d, n, d.pcr.pin1 ? 3 : 2, q, 2);
    }
    launch_factor(L, d, n, d.pcr.steps, 1, 2, d.pcr.pin0 || d.pcr.pin1);
}

// separator system (the blocks shared by neighbouring ranks, summed over the ranks): parallel cyclic reduction,
// replicated on every rank; then the back-substitution of this rank's chain interior
void launch_bcr_separators(Launcher &L, const Dev &d) {
    const size_t sh_backsub = (size_t)3 * BD * BD * sizeof(double);
    const int ns = d.n_sep;
    const char *fe = getenv("SSBA_NO_PCR_FUSED");
    if (d.spcrf.on && !bcr_legacy() && !(fe && fe[0] == '1')) {
        // one launch per step, the decoupled last step solves its blocks itself: steps + 1 launches instead of 2 steps + 2
        for (int q = 0; q < d.spcr.steps; ++q) launch_pcr_fused_step(L, d, ns, q, 3);
        launch_pcr_fused_top(L, d, ns, d.spcr.steps, 1, 3);
    } else {
    for (int q = 0; q < d.spcr.steps; ++q) {
        launch_factor(L, d, ns, q, 0, 3, true);
        launch_reduce(L, d, ns, 2, q, 3);
    }
    launch_factor(L, d, ns, d.spcr.steps, 1, 3, false);
    LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(ns), dim3(BS_THREADS), sh_backsub, d, 0, 1, 3);
    }
    launch_sep_scatter(L, d);       // x0 at the separator poses <- separator solution
    LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(d.pcr.n), dim3(BS_THREADS), sh_backsub, d, d.pcr.level, 1, 2);
    for (int l = d.pcr.level - 1; l >= 0; --l)
        LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(n_odd(d.lev[l], d.lev[l].pin)), dim3(BS_THREADS), sh_backsub, d, l, 0, 0);
}

int configure_kernels() {
    const int sh_reduce = (int)(2 * BD * BD * sizeof(double));
    if (hipFuncSetAttribute((const void *)k_bcr_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, sh_reduce) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FACT_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcr_backsub, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * BD * BD * sizeof(double))) != hipSuccess) return -1;
    return configure_bcr_mf();
}

}  // namespace ssba

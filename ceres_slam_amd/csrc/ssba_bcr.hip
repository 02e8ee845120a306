// Block cyclic reduction (BCR) of the block-tridiagonal reduced camera system S on gfx950.
//
// S has Nsb diagonal blocks D_i (BD x BD, BD = 72 = 12 poses) and couplings L_i = S[i, i-1].
// Level l eliminates its odd blocks (all in parallel, one workgroup each):
//   k_bcr_factor  (odd i):  D_i = G G^T ; YL = G^-1 L_i ; YU = G^-1 L_{i+1}^T ; yr = G^-1 r_i
//   k_bcr_reduce  (even e): D' = D_e - YU(e-1)^T YU(e-1) - YL(e+1)^T YL(e+1)
//                           L' = -YU(e-1)^T YL(e-1) ;  r' = r_e - YU(e-1)^T yr(e-1) - YL(e+1)^T yr(e+1)
//   k_bcr_backsub (odd i):  x_i = G^-T (yr - YL x_{i-1} - YU x_{i+1})      (top-down)
// and recurses on the even blocks; the last level (one block) is a plain Cholesky solve.
// G (with 1/G_kk on its diagonal), YL and yr overwrite D_i, L_i and r_i in place.
//
// The factor kernel is latency bound (72 dependent pivots); it keeps every 6x6 tile of
// [D | L | U^T | r] in the registers of one lane for the whole factorisation ("owner
// computes") and stages only the current block column / block row through LDS, so LDS
// carries operands (2 reads per 6 FMAs) and never accumulators.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "ssba_launch.h"
#include "ssba_types.h"

namespace ssba {

constexpr int NB = 6;                       // tile edge
constexpr int NBLK = BD / NB;               // 12 block rows
constexpr int NCB = (2 * BD + 1 + NB - 1) / NB;   // 25 column blocks of the right-hand sides
constexpr int NT_A = NBLK * (NBLK + 1) / 2; // 78 lower tiles of D
constexpr int NT = NT_A + NBLK * NCB;       // 378 tiles in all
constexpr int FACT_THREADS = 384;

// tile table, sorted by the step in which a tile becomes final so that whole waves retire early
__constant__ uint8_t c_tile_type[NT];   // 0 = tile of D (rb >= cb), 1 = tile of the right-hand sides
__constant__ uint8_t c_tile_rb[NT];
__constant__ uint8_t c_tile_cb[NT];

int upload_bcr_tables(hipStream_t s) {
    struct T { int fin, type, rb, cb; };
    std::vector<T> v;
    for (int rb = 0; rb < NBLK; ++rb)
        for (int cb = 0; cb <= rb; ++cb) v.push_back({cb, 0, rb, cb});
    for (int rb = 0; rb < NBLK; ++rb)
        for (int cb = 0; cb < NCB; ++cb) v.push_back({rb, 1, rb, cb});
    std::stable_sort(v.begin(), v.end(), [](const T &a, const T &b) { return a.fin < b.fin; });
    uint8_t ty[NT], rb[NT], cb[NT];
    for (int i = 0; i < NT; ++i) { ty[i] = (uint8_t)v[i].type; rb[i] = (uint8_t)v[i].rb; cb[i] = (uint8_t)v[i].cb; }
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tile_type), ty, NT, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tile_rb), rb, NT, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tile_cb), cb, NT, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}

// 1/sqrt(a) to fp64 accuracy: hardware estimate + two Newton steps (no IEEE sqrt/div chain)
__device__ __forceinline__ double rsqrt_nr(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}

__global__ __launch_bounds__(FACT_THREADS) void k_bcr_factor(Dev d, int lev, int top) {
    State &st = *d.st;
    if (st.terminated || st.step_failed) return;
    __shared__ double sM[24];               // L^-1 of the current diagonal tile (21 used)
    __shared__ double sPA[NBLK * 36];       // current block column of G, one tile per block row
    __shared__ double sPY[NCB * 36];        // current block row of Y, tiles stored transposed
    __shared__ int sBad;
    const BcrLevel &L = d.lev[lev];
    const int blk = top ? 0 : 2 * blockIdx.x + 1;
    const bool hasL = !top, hasU = !top && (blk + 1 < L.n);
    double *Dg = L.D + (size_t)blk * BD * BD;
    double *Lg = L.L + (size_t)blk * BD * BD;
    const double *Ug = hasU ? L.L + (size_t)(blk + 1) * BD * BD : nullptr;
    double *YUg = top ? nullptr : L.YU + (size_t)blockIdx.x * BD * BD;
    double *rg = L.r + (size_t)blk * BD;
    const int t = threadIdx.x;
    const bool has_tile = t < NT;
    const int type = has_tile ? c_tile_type[t] : 2;
    const int rb = has_tile ? c_tile_rb[t] : 0, cb = has_tile ? c_tile_cb[t] : 0;
    if (t == 0) sBad = 0;

    double acc[36];
    if (type == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[6 * i + j] = Dg[(size_t)(rb * 6 + i) * BD + cb * 6 + j];
    } else if (type == 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int c = cb * 6 + j, r = rb * 6 + i;
                double v = 0.0;
                if (c < BD) v = hasL ? Lg[(size_t)r * BD + c] : 0.0;
                else if (c < 2 * BD) v = hasU ? Ug[(size_t)(c - BD) * BD + r] : 0.0;   // L_{i+1}^T
                else if (c == 2 * BD) v = rg[r];
                acc[6 * i + j] = v;
            }
    }
    __syncthreads();

    for (int kb = 0; kb < NBLK; ++kb) {
        // (1) owner of the diagonal tile: 6x6 Cholesky and its inverse
        if (type == 0 && rb == kb && cb == kb) {
            double l[6][6], m[6][6];
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double s = acc[6 * j + j];
#pragma unroll
                for (int q = 0; q < j; ++q) s -= l[j][q] * l[j][q];
                if (!(s > 0.0) || !isfinite(s)) { bad = true; s = 1.0; }
                const double r = rsqrt_nr(s);
                l[j][j] = s * r;
                m[j][j] = r;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double v = acc[6 * i + j];
#pragma unroll
                    for (int q = 0; q < j; ++q) v -= l[i][q] * l[j][q];
                    l[i][j] = v * r;
                }
            }
            // M = L^-1 (lower)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int q = j; q < i; ++q) v -= l[i][q] * m[q][j];
                    m[i][j] = v * m[i][i];
                }
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) sM[n++] = m[i][j];
            if (bad) sBad = 1;
            // G tile: L below the diagonal, 1/L_kk ON the diagonal (back-substitution multiplies)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    Dg[(size_t)(kb * 6 + i) * BD + kb * 6 + j] = (j < i) ? l[i][j] : (j == i ? m[i][i] : 0.0);
        }
        __syncthreads();
        if (sBad) {
            if (t == 0) st.step_failed = 1;
            return;
        }
        // (2) block column kb of G and block row kb of Y become final
        if (type == 0 && cb == kb && rb > kb) {
            double M[21];
#pragma unroll
            for (int i = 0; i < 21; ++i) M[i] = sM[i];
            double x[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int c = 0; c < 6; ++c) {   // X' = X M^T : x'[i][c] = sum_{q<=c} x[i][q] M[c][q]
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q <= c; ++q) v += acc[6 * i + q] * M[c * (c + 1) / 2 + q];
                    x[6 * i + c] = v;
                }
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[i] = x[i]; sPA[rb * 36 + i] = x[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int c = 0; c < 6; ++c) Dg[(size_t)(rb * 6 + i) * BD + kb * 6 + c] = x[6 * i + c];
        } else if (type == 1 && rb == kb) {
            double M[21];
#pragma unroll
            for (int i = 0; i < 21; ++i) M[i] = sM[i];
            double y[36];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) {   // Y' = M Y : y'[r][j] = sum_{q<=r} M[r][q] y[q][j]
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q <= r; ++q) v += M[r * (r + 1) / 2 + q] * acc[6 * q + j];
                    y[6 * r + j] = v;
                }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    sPY[cb * 36 + j * 6 + r] = y[6 * r + j];      // transposed for the update below
                    const int c = cb * 6 + j, row = kb * 6 + r;
                    if (c < BD) { if (hasL) Lg[(size_t)row * BD + c] = y[6 * r + j]; }
                    else if (c < 2 * BD) { if (YUg) YUg[(size_t)row * BD + (c - BD)] = y[6 * r + j]; }
                    else if (c == 2 * BD) rg[row] = y[6 * r + j];
                }
        }
        __syncthreads();
        // (3) trailing update: acc -= G[rb][kb] * B^T with B = G[cb][kb] (D tiles) or Y[kb][cb]^T
        const bool upd = (type == 0 && cb > kb) || (type == 1 && rb > kb);
        if (upd) {
            const double *a = sPA + rb * 36;
            const double *b = (type == 0) ? sPA + cb * 36 : sPY + cb * 36;
            double av[36], bv[36];
#pragma unroll
            for (int i = 0; i < 36; ++i) { av[i] = a[i]; bv[i] = b[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double v = acc[6 * i + j];
#pragma unroll
                    for (int q = 0; q < 6; ++q) v -= av[6 * i + q] * bv[6 * j + q];
                    acc[6 * i + j] = v;
                }
        }
        // no barrier here: the next step's first barrier orders these LDS reads before the
        // next panel writes; sM is only rewritten after every reader passed barrier (2)
    }
}

// C -= A^T B for BD x BD operands: 3-way split over k (432 lanes = 3 x 144 tiles of 6x6),
// partial sums combined through LDS in a fixed order.
constexpr int RED_THREADS = 448;
constexpr int KSPLIT = 3;
constexpr int KCH = BD / KSPLIT;   // 24

__device__ __forceinline__ void stage_block(double *dst, const double *__restrict__ src) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (int e = threadIdx.x; e < BD * BD / 2; e += RED_THREADS) d2[e] = s2[e];
}

__device__ __forceinline__ void tile_mac(double *acc, const double *sA, const double *sB, int g, int tr, int tc) {
    for (int k = g * KCH; k < (g + 1) * KCH; ++k) {
        double a[6], b[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { a[i] = sA[k * BD + tr * 6 + i]; b[i] = sB[k * BD + tc * 6 + i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[6 * i + j] += a[i] * b[j];
    }
}

// grid = (n_next, 2): y = 0 -> D' and r' ; y = 1 -> L'
__global__ __launch_bounds__(RED_THREADS) void k_bcr_reduce(Dev d, int lev) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed) return;
    extern __shared__ double lds[];
    double *sA = lds, *sB = lds + BD * BD;
    const BcrLevel &L = d.lev[lev];
    const BcrLevel &N = d.lev[lev + 1];
    const int m = blockIdx.x, e = 2 * m;
    const int t = threadIdx.x;
    const bool act = t < KSPLIT * 144;
    const int g = t / 144, tt = t - g * 144;
    const int tr = tt / 12, tc = tt - tr * 12;
    const bool hasPrev = e >= 2 || e - 1 >= 0, hasNext = e + 1 < L.n;
    const int tp = (e - 2) / 2;     // YU slot of odd block e-1
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    double *out;
    if (blockIdx.y == 0) {
        out = N.D + (size_t)m * BD * BD;
        if (e - 1 >= 0) {
            stage_block(sA, L.YU + (size_t)tp * BD * BD);
            __syncthreads();
            if (act) tile_mac(acc, sA, sA, g, tr, tc);
            __syncthreads();
        }
        if (hasNext) {
            stage_block(sA, L.L + (size_t)(e + 1) * BD * BD);
            __syncthreads();
            if (act) tile_mac(acc, sA, sA, g, tr, tc);
            __syncthreads();
        }
        // r' (72 lanes, operands straight from L2)
        if (t < BD) {
            double v = L.r[(size_t)e * BD + t];
            if (e - 1 >= 0) {
                const double *YU = L.YU + (size_t)tp * BD * BD, *yr = L.r + (size_t)(e - 1) * BD;
                for (int k = 0; k < BD; ++k) v -= YU[(size_t)k * BD + t] * yr[k];
            }
            if (hasNext) {
                const double *YL = L.L + (size_t)(e + 1) * BD * BD, *yr = L.r + (size_t)(e + 1) * BD;
                for (int k = 0; k < BD; ++k) v -= YL[(size_t)k * BD + t] * yr[k];
            }
            N.r[(size_t)m * BD + t] = v;
        }
    } else {
        if (m == 0) return;   // L'[0] does not exist
        out = N.L + (size_t)m * BD * BD;
        stage_block(sA, L.YU + (size_t)tp * BD * BD);
        stage_block(sB, L.L + (size_t)(e - 1) * BD * BD);
        __syncthreads();
        if (act) tile_mac(acc, sA, sB, g, tr, tc);
        __syncthreads();
    }
    (void)hasPrev;
    // combine the k-split partials in LDS (fixed order), then out = base - sum
    double *part = lds;   // 2 x 144 x 36 doubles = 82,944 B: fits the operand area exactly
    if (act && g > 0) {
#pragma unroll
        for (int i = 0; i < 36; ++i) part[((g - 1) * 144 + tt) * 36 + i] = acc[i];
    }
    __syncthreads();
    if (act && g == 0) {
        const double *base = (blockIdx.y == 0) ? L.D + (size_t)e * BD * BD : nullptr;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double s = (acc[6 * i + j] + part[tt * 36 + 6 * i + j]) + part[(144 + tt) * 36 + 6 * i + j];
                const size_t o = (size_t)(tr * 6 + i) * BD + tc * 6 + j;
                out[o] = (base ? base[o] : 0.0) - s;
            }
    }
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// x_i = G^-T (yr - YL x_{i-1} - YU x_{i+1}); x lives in d.x0 at level-0 block positions.
__global__ __launch_bounds__(256) void k_bcr_backsub(Dev d, int lev, int top) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed) return;
    __shared__ double sG[BD * BD];
    __shared__ double sv[BD];
    __shared__ double sxm[BD], sxp[BD];
    const BcrLevel &L = d.lev[lev];
    const int blk = top ? 0 : 2 * blockIdx.x + 1;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const double *G = L.D + (size_t)blk * BD * BD;
    double *xi = d.x0 + ((size_t)blk << lev) * BD;
    const bool hasU = !top && (blk + 1 < L.n);
    {
        const double2 *s2 = reinterpret_cast<const double2 *>(G);
        double2 *d2 = reinterpret_cast<double2 *>(sG);
        for (int e = t; e < BD * BD / 2; e += 256) d2[e] = s2[e];
    }
    if (t < BD) {
        sxm[t] = top ? 0.0 : d.x0[((size_t)(blk - 1) << lev) * BD + t];
        sxp[t] = hasU ? d.x0[((size_t)(blk + 1) << lev) * BD + t] : 0.0;
    }
    __syncthreads();
    // v = yr - YL x_{i-1} - YU x_{i+1}: each wave owns 18 rows, lanes stride the row
    const double *YL = L.L + (size_t)blk * BD * BD;
    const double *YU = top ? nullptr : L.YU + (size_t)blockIdx.x * BD * BD;
    for (int r = w * 18; r < w * 18 + 18; ++r) {
        double p = 0.0;
        if (!top) {
            p += YL[(size_t)r * BD + lane] * sxm[lane];
            if (lane < BD - 64) p += YL[(size_t)r * BD + 64 + lane] * sxm[64 + lane];
            if (hasU) {
                p += YU[(size_t)r * BD + lane] * sxp[lane];
                if (lane < BD - 64) p += YU[(size_t)r * BD + 64 + lane] * sxp[64 + lane];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
        if (lane == 0) sv[r] = L.r[(size_t)blk * BD + r] - p;
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

void launch_bcr(Launcher &L, const Dev &d) {
    const size_t sh_reduce = (size_t)2 * BD * BD * sizeof(double);
    const int nl = d.n_levels;
    for (int l = 0; l + 1 < nl; ++l) {
        const int n = d.lev[l].n;
        LAUNCH(KC_BCR_FACTOR, k_bcr_factor, dim3(n / 2), dim3(FACT_THREADS), 0, d, l, 0);
        LAUNCH(KC_BCR_REDUCE, k_bcr_reduce, dim3((n + 1) / 2, 2), dim3(RED_THREADS), sh_reduce, d, l);
    }
    LAUNCH(KC_BCR_FACTOR, k_bcr_factor, dim3(1), dim3(FACT_THREADS), 0, d, nl - 1, 1);
    LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(1), dim3(256), 0, d, nl - 1, 1);
    for (int l = nl - 2; l >= 0; --l)
        LAUNCH(KC_BCR_BACKSUB, k_bcr_backsub, dim3(d.lev[l].n / 2), dim3(256), 0, d, l, 0);
}

int configure_kernels() {
    const int sh_reduce = (int)(2 * BD * BD * sizeof(double));
    if (hipFuncSetAttribute((const void *)k_bcr_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, sh_reduce) != hipSuccess) return -1;
    return 0;
}

}  // namespace ssba

// General-structure path of the solve for gfx950: problems whose landmark tracks are longer than the TW slots of the
// window layout, whose pose co-visibility is not banded (loop closures), or that carry what has no place in that
// layout (a second residual block on one (pose, landmark) pair, per-block stiffness, relative-pose blocks) cannot use
// the block-tridiagonal reduced system of ssba_kernels.hip / ssba_bcr.hip.  They keep every other kernel
// (linearisation, back-substitution, trust-region control: templates on the observation layout) and swap the
// middle of the iteration for
//   k_dn_wy      per observation: W = J_p^T J_l and Y = W C^-1 (C = H_ll + damping), stored once (HBM stream);
//                with lighting terms k_ph_dn_wy (ssba_phong_solver.hip) stores the 6x6 versions
//   k_dn_schur   one work-group per 6x6 block (a <= b) of S = H_pp - sum_l Y_a W_b^T: the block's observation pairs
//                (host-built list) spread over the lanes, fixed-order reduction (no float atomics), plus the J_a^T J_b
//                of relative-pose blocks; writes the lower triangle of the dense matrix
//   k_dn_rhs     one wave per free pose: reduced gradient g_p - sum Y g_l, stored as an extra row of the matrix
//   k_dn_finish  Jacobi scale at iteration 0, LM damping on the diagonal
//   k_dn_potrf / k_dn_trsm / k_dn_syrk   right-looking blocked Cholesky (DN_BS = 64); the extra rows (the right-hand
//                side, the columns of S_pb for free shared blocks, unit vectors for a covariance block) are
//                forward-solved as part of the factorisation
//   k_dn_bwd     right-looking block back-substitution with L^T, one sweep per extra row -> pose step x0, S_pp^-1 S_pb
// The factorisation skips zero 64x64 blocks: ssba_finalize runs a symbolic Cholesky at block granularity, so a
// banded problem costs O(n b^2) and a loop closure only fills the block rows between its two ends.
// What Ceres does here (SPARSE_SCHUR / DENSE_SCHUR on the reduced camera matrix, schur_complement_solver.cc)
// is the same elimination order: landmarks first, then one Cholesky of S.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "ssba_types.h"
#include "ssba_launch.h"
#include "ssba_device.h"
#include "ssba_posefactor_device.h"

namespace ssba {

constexpr int TP = DN_BS + 1;   // padded LDS row: stride 65 doubles keeps column accesses conflict-free

__device__ __forceinline__ int dn_tri21(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

__global__ __launch_bounds__(256) void k_dn_wy(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= d.n_obs) return;
    const uint32_t k = d.dn_obs_pose[e];
    if (d.pose_free[k] < 0) return;      // rows of constant poses are never read
    const int l = (int)d.dn_obs_lm[e];
    double h[6], dmp[3], Ci[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
    landmark_damping(d, st, l, h, dmp);
    if (!inv3_spd(h, dmp, Ci)) {
        d.st->step_failed = 1;
#pragma unroll
        for (int c = 0; c < 6; ++c) Ci[c] = 0.0;
    }
    const double *T = d.poses + (size_t)k * 12;
    ObsLin o;
    double Sk[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) Sk[c] = d.dn_Sobs ? d.dn_Sobs[(size_t)e * 9 + c] : d.S[c];
    obs_linearize_S(d, Sk, T, d.pts[l], d.pts[(size_t)d.Lpad + l], d.pts[2 * (size_t)d.Lpad + l], d.dn_u[e], d.dn_v[e], d.dn_d[e], o);
    double Jp[18], Jl[9];
    jac_pose(o, Jp);
    jac_point(o, T, Jl);
    double *W = d.dn_W + (size_t)e * 18, *Y = d.dn_Y + (size_t)e * 18;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) w[c] = Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c] + Jp[12 + a] * Jl[6 + c];
        W[3 * a + 0] = w[0]; W[3 * a + 1] = w[1]; W[3 * a + 2] = w[2];
        Y[3 * a + 0] = w[0] * Ci[0] + w[1] * Ci[1] + w[2] * Ci[2];
        Y[3 * a + 1] = w[0] * Ci[1] + w[1] * Ci[3] + w[2] * Ci[4];
        Y[3 * a + 2] = w[0] * Ci[2] + w[1] * Ci[4] + w[2] * Ci[5];
    }
}

// One work-group per block.  Every thread takes every 256th observation pair of the block and accumulates the whole
// 6x6 product Y_a W_b^T in registers (the loads of different pairs are independent: the kernel is bound by the
// latency of the two gathers per pair, so pairs are spread over as many lanes as possible); the 36 partial sums are
// then reduced over the wave by shuffles and over the four waves through LDS, both in a fixed order.
template <int LD> __global__ __launch_bounds__(256) void k_dn_schur(Dev d) {      // LD = 3 (position) or 6 (position + normal)
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double part[4][36];
    const int blk = blockIdx.x;
    const uint32_t a = d.dn_blk_a[blk], b = d.dn_blk_b[blk];
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    for (uint32_t i = d.dn_blk_start[blk] + threadIdx.x; i < d.dn_blk_start[blk + 1]; i += 256) {
        const double2 *Y2 = reinterpret_cast<const double2 *>(d.dn_Y + (size_t)d.dn_pair_a[i] * (6 * LD));
        const double2 *W2 = reinterpret_cast<const double2 *>(d.dn_W + (size_t)d.dn_pair_b[i] * (6 * LD));
        double y[6 * LD], w[6 * LD];
#pragma unroll
        for (int q = 0; q < 3 * LD; ++q) {
            const double2 yv = Y2[q], wv = W2[q];
            y[2 * q] = yv.x; y[2 * q + 1] = yv.y; w[2 * q] = wv.x; w[2 * q + 1] = wv.y;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = acc[6 * r + c];
#pragma unroll
                for (int m = 0; m < LD; ++m) v += y[LD * r + m] * w[LD * c + m];
                acc[6 * r + c] = v;
            }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) part[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x >= 36) return;
    const int el = threadIdx.x, r = el / 6, c = el - r * 6;
    double v = -(part[0][el] + part[1][el] + part[2][el] + part[3][el]);
    if (d.dn_blk_rf)      // relative-pose blocks coupling the two poses of this block: J_a^T J_b
        for (uint32_t q = d.dn_blk_rf_start[blk]; q < d.dn_blk_rf_start[blk + 1]; ++q) {
            const uint32_t ent = d.dn_blk_rf[q];
            const int e1 = (int)(ent & 0x7FFFFFFFu), e2 = pf_partner(d, e1);
            const bool swapped = (ent >> 31) != 0;       // first pose of the block is the factor's SECOND pose
            const int k1 = d.free_pose[swapped ? b : a], k2 = d.free_pose[swapped ? a : b];
            const double *T1 = d.poses + (size_t)k1 * 12, *T2 = d.poses + (size_t)k2 * 12;
            double rr[6], J1[36], J2[36];
            int dim;
            pf_evaluate(d, e1, T1, T2, rr, J1, &dim);
            pf_evaluate(d, e2, T2, T1, rr, J2, &dim);
            const double *Ja = swapped ? J2 : J1, *Jb = swapped ? J1 : J2;
            for (int m = 0; m < 6; ++m) v += Ja[6 * m + r] * Jb[6 * m + c];
        }
    const size_t lda = (size_t)d.dn_pad;
    if (a == b) {
        if (c < r) return;    // (r, c) with r <= c stands for the symmetric pair; stored at (row 6a+c, col 6a+r)
        v += d.hpp[(size_t)d.free_pose[a] * 21 + dn_tri21(r, c)];
    }
    d.dn_S[((size_t)b * 6 + c) * lda + (size_t)a * 6 + r] = v;     // block (b, a) of the lower triangle = block (a, b)^T
}

template <int LD> __global__ __launch_bounds__(256) void k_dn_rhs(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= d.nfree) return;
    const int k = d.free_pose[f];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = d.dn_pose_start[k] + lane; i < d.dn_pose_start[k + 1]; i += 64) {
        const uint32_t e = d.dn_pose_obs[i];
        const int l = (int)d.dn_obs_lm[e];
        double g[LD];
#pragma unroll
        for (int m = 0; m < LD; ++m) g[m] = d.gl[(size_t)m * d.Lpad + l];
        const double *Y = d.dn_Y + (size_t)e * (6 * LD);
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int m = 0; m < LD; ++m) acc[c] += Y[LD * c + m] * g[m];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const size_t i = (size_t)f * 6 + c;
            const double g = d.gp[(size_t)k * 6 + c];
            d.dn_S[(size_t)d.dn_pad * d.dn_pad + i] = -(g - acc[c]);       // right-hand side = -reduced gradient
            d.xv[d.off_gp + i] = g;
            d.xv[d.off_hdiag + i] = d.hpp[(size_t)k * 21 + dn_tri21(c, c)];
        }
    }
}

__global__ __launch_bounds__(256) void k_dn_finish(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.dn_pad) return;
    double *Dd = d.dn_S + (size_t)i * d.dn_pad + i;
    if (i < d.n_dn) {
        const double h = d.xv[d.off_hdiag + i];
        if (st.iteration == 0) d.sp[i] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
        const double s = d.sp[i], s2 = s * s;
        *Dd += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    } else {
        *Dd = 1.0;     // padding up to the block size
    }
}

// ---- blocked Cholesky, lower triangle, row-major with stride dn_pad; block row dn_pad / DN_BS is the rhs ------
static __device__ __forceinline__ double lane_value(double v, int lane) {      // lane is uniform (unrolled loops)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Diagonal block: one wave, lane r keeps row r of the 64x64 tile in registers; the pivot column travels by
// readlane (no LDS, no barrier).  The chain of 64 dependent columns is what bounds this kernel.
__global__ __launch_bounds__(64) void k_dn_potrf(Dev d, int j) {
    State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    const size_t lda = (size_t)d.dn_pad;
    const int r = threadIdx.x;
    double *A = d.dn_S + ((size_t)j * DN_BS + r) * lda + (size_t)j * DN_BS;
    double a[DN_BS];
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) a[c] = A[c];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) {
        const double piv = lane_value(a[c], c);
        bad = bad || !(piv > 0.0) || !isfinite(piv);
        const double l = a[c] * fast_rsqrt(piv);       // lane c: sqrt(piv); lanes r < c hold nothing of the factor
        a[c] = l;
#pragma unroll
        for (int cc = c + 1; cc < DN_BS; ++cc) a[cc] -= l * lane_value(l, cc);
    }
    if (bad) { if (r == 0) st.step_failed = 1; return; }
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) A[c] = c <= r ? a[c] : 0.0;
}

// rows of the panel below the diagonal block: X = A L_jj^-T, one thread per row, right-looking substitution;
// one work-group per non-zero block row of this block column (rows[])
__global__ __launch_bounds__(64) void k_dn_trsm(Dev d, int j, const uint32_t *rows) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double Lt[DN_BS * DN_BS];       // read as broadcasts only: no padding needed
    __shared__ double rd[DN_BS];
    const size_t lda = (size_t)d.dn_pad;
    const int i = (int)rows[blockIdx.x], tid = threadIdx.x;
    const double *Lj = d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS;
    double *A = d.dn_S + ((size_t)i * DN_BS + tid) * lda + (size_t)j * DN_BS;     // this thread's row: 512 contiguous bytes
    for (int m = 0; m < DN_BS; ++m) Lt[m * DN_BS + tid] = Lj[(size_t)m * lda + tid];
    rd[tid] = 1.0 / Lj[(size_t)tid * lda + tid];
    double x[DN_BS];
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) x[c] = A[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) {
        x[c] *= rd[c];
#pragma unroll
        for (int cc = c + 1; cc < DN_BS; ++cc) x[cc] -= x[c] * Lt[cc * DN_BS + c];
    }
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) A[c] = x[c];
}

// trailing update A_ik -= L_ij L_kj^T over the non-zero block rows i >= k of block column j (tile list ti / tk):
// one 64x64 tile per work-group, 4x4 outputs per thread
__global__ __launch_bounds__(256) void k_dn_syrk(Dev d, int j, const uint32_t *ti, const uint32_t *tk) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    const int i = (int)ti[blockIdx.x], k = (int)tk[blockIdx.x];
    constexpr int KH = DN_BS / 2;
    __shared__ double sA[KH * TP], sB[KH * TP];      // transposed halves of the two panels: s[m][row]
    const size_t lda = (size_t)d.dn_pad;
    const double *Ai = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS;
    const double *Ak = d.dn_S + ((size_t)k * DN_BS) * lda + (size_t)j * DN_BS;
    const int tid = threadIdx.x, m0 = tid & (KH - 1), q0 = tid / KH;
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int kh = 0; kh < DN_BS; kh += KH) {
        if (kh) __syncthreads();
        for (int r = q0; r < DN_BS; r += 256 / KH) {
            sA[m0 * TP + r] = Ai[(size_t)r * lda + kh + m0];
            sB[m0 * TP + r] = Ak[(size_t)r * lda + kh + m0];
        }
        __syncthreads();
#pragma unroll 8
        for (int m = 0; m < KH; ++m) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = sA[m * TP + ty + 16 * a]; bv[a] = sB[m * TP + tx + 16 * a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
        }
    }
    double *C = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)k * DN_BS;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) C[(size_t)(ty + 16 * a) * lda + tx + 16 * b] -= acc[a][b];
}

// Back-substitution L^T x = y in the rhs row, right-looking: step i (block row i of L, x_i known) subtracts
// L_ij^T x_i from y_j for the non-zero blocks j < i (cols[], one work-group each); the last work-group owns
// j = i - 1, whose y is complete after its own update, and solves x_{i-1} = L_{i-1,i-1}^-T y_{i-1} right away.
// Step i = nbk (the rhs row itself) has no update and only solves the last block.
__global__ __launch_bounds__(256) void k_dn_bwd(Dev d, int i, const uint32_t *cols, int n_upd, int upd_last, int row) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double t[DN_BS * TP], part[4][DN_BS], rd[DN_BS];
    const size_t lda = (size_t)d.dn_pad;
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    double *xrow = d.dn_S + (size_t)(d.dn_pad + row) * lda;     // row 0: the solve's right-hand side; rows 1.. : extra ones (covariance)
    const bool solver = (int)blockIdx.x == n_upd;
    const int j = solver ? i - 1 : (int)cols[blockIdx.x];
    double s = 0.0;
    if (!solver || upd_last) {
        const double *Lij = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS;
        const double *xi = xrow + (size_t)i * DN_BS;
#pragma unroll 4
        for (int r = q; r < DN_BS; r += 4) s += Lij[(size_t)r * lda + c] * xi[r];
    }
    part[q][c] = s;
    if (solver) {
        const double *Lj = d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS;
        for (int m = q; m < DN_BS; m += 4) t[m * TP + c] = Lj[(size_t)m * lda + c];
        if (q == 0) rd[c] = 1.0 / Lj[(size_t)c * lda + c];
    }
    __syncthreads();
    if (q != 0) return;
    double v = xrow[j * DN_BS + c] - (part[0][c] + part[1][c] + part[2][c] + part[3][c]);
    if (solver) {
        // one wave: lane c owns x_c; columns of L^T are rows of L
        for (int m = DN_BS - 1; m >= 0; --m) {
            double xm = 0.0;
            if (c == m) xm = v * rd[m];
            xm = __shfl(xm, m, 64);
            if (c == m) v = xm;
            else if (c < m) v -= t[m * TP + c] * xm;
        }
        const int g = j * DN_BS + c;
        if (g < d.n_dn && row == 0) d.x0[g] = v;
    }
    xrow[j * DN_BS + c] = v;
}

// free shared blocks on the general layout: the columns of S_pb ride as rows 1..nb of the right-hand-side block row,
// so the factorisation forward-solves them and k_dn_bwd back-substitutes them like the solve's own right-hand side
__global__ __launch_bounds__(256) void k_dn_border_rows(Dev d, int store) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse || (store && st.step_failed)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_dn) return;
    const size_t lda = (size_t)d.dn_pad;
    for (int c = 0; c < d.nb; ++c) {
        double *row = d.dn_S + ((size_t)d.dn_pad + 1 + c) * lda;
        if (store) d.Zb[(size_t)i * NBP + c] = row[i];         // Z = S_pp^-1 S_pb
        else row[i] = d.Spb[(size_t)i * NBP + c];
    }
}

// ----------------------------------------------------------------- launchers ---
void launch_dense_schur(Launcher &L, const Dev &d) {
    if (d.dn_pad > 0) hipMemsetAsync(d.dn_S, 0, (size_t)(d.dn_pad + DN_BS) * d.dn_pad * sizeof(double), L.stream);
    if (d.phong) {       // 6-D landmark blocks: C^-1 first, W / Y are 6x6 (ssba_phong_solver.hip)
        launch_ph_dense_wy(L, d);
        LAUNCH(KC_SCHUR, k_dn_schur<6>, dim3(d.dn_nblk), dim3(256), 0, d);
        LAUNCH(KC_ASSEMBLE, k_dn_rhs<6>, dim3((d.nfree + 3) / 4), dim3(256), 0, d);
        if (d.nb) launch_ph_dense_border(L, d);
        return;
    }
    LAUNCH(KC_SCHUR, k_dn_wy, dim3((d.n_obs + 255) / 256), dim3(256), 0, d);
    LAUNCH(KC_SCHUR, k_dn_schur<3>, dim3(d.dn_nblk), dim3(256), 0, d);
    LAUNCH(KC_ASSEMBLE, k_dn_rhs<3>, dim3((d.nfree + 3) / 4), dim3(256), 0, d);
}

void launch_dense_finish(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_dn_finish, dim3((d.dn_pad + 255) / 256), dim3(256), 0, d);
}

void launch_dense_solve(Launcher &L, const Dev &d, int n_rhs_rows) {
    const DensePlan &pl = L.dense;
    const int nbk = pl.nbk;
    const bool border = d.nb > 0 && n_rhs_rows == 1;      // the solve of an iteration (not the covariance's unit rows)
    if (border) {
        LAUNCH(KC_BORDER, k_dn_border_rows, dim3((d.n_dn + 255) / 256), dim3(256), 0, d, 0);
        n_rhs_rows = 1 + d.nb;
    }
    for (int j = 0; j < nbk; ++j) {
        const uint32_t r0 = pl.row_start[j], nr = pl.row_start[j + 1] - r0, t0 = pl.tile_start[j], nt = pl.tile_start[j + 1] - t0;
        LAUNCH(KC_BCR_FACTOR, k_dn_potrf, dim3(1), dim3(64), 0, d, j);
        LAUNCH(KC_BCR_FACTOR, k_dn_trsm, dim3(nr), dim3(64), 0, d, j, d.dn_rows + r0);
        LAUNCH(KC_BCR_REDUCE, k_dn_syrk, dim3(nt), dim3(256), 0, d, j, d.dn_ti + t0, d.dn_tk + t0);
    }
    for (int row = 0; row < n_rhs_rows; ++row)
        for (int i = nbk; i >= 1; --i) {
            const uint32_t c0 = pl.col_start[i], nc = pl.col_start[i + 1] - c0;
            LAUNCH(KC_BCR_BACKSUB, k_dn_bwd, dim3(nc + 1), dim3(256), 0, d, i, d.dn_cols + c0, (int)nc, (int)pl.upd_last[i], row);
        }
    if (border) {
        LAUNCH(KC_BORDER, k_dn_border_rows, dim3((d.n_dn + 255) / 256), dim3(256), 0, d, 1);
        launch_border_finish(L, d);
    }
}

}  // namespace ssba

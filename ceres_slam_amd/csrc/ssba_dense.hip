// General-structure path of the stereo-BA solve for gfx950: problems whose landmark tracks are longer than
// the TW slots of the window layout, or whose pose co-visibility is not banded (loop closures), cannot use the
// block-tridiagonal reduced system of ssba_kernels.hip / ssba_bcr.hip.  They keep every other kernel
// (linearisation, back-substitution, trust-region control: templates on the observation layout) and swap the
// middle of the iteration for
//   k_dn_wy      per observation: W = J_p^T J_l and Y = W C^-1 (C = H_ll + damping), stored once (HBM stream)
//   k_dn_schur   one wave per 6x6 block (a <= b) of S = H_pp - sum_l Y_a W_b^T, summing that block's observation
//                pairs in a fixed order (no float atomics); writes the lower triangle of the dense matrix
//   k_dn_rhs     one wave per free pose: reduced gradient g_p - sum Y g_l, stored as an extra row of the matrix
//   k_dn_finish  Jacobi scale at iteration 0, LM damping on the diagonal
//   k_dn_potrf / k_dn_trsm / k_dn_syrk   right-looking blocked Cholesky (DN_BS = 64), fp64 FMA bound; the extra
//                row makes the forward solve part of the factorisation
//   k_dn_bwd     block back-substitution with L^T -> pose step x0
// What Ceres does here (SPARSE_SCHUR / DENSE_SCHUR on the reduced camera matrix, schur_complement_solver.cc)
// is the same elimination order: landmarks first, then one Cholesky of S.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "ssba_types.h"
#include "ssba_launch.h"
#include "ssba_device.h"

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
    obs_linearize(d, T, d.pts[l], d.pts[(size_t)d.Lpad + l], d.pts[2 * (size_t)d.Lpad + l], d.dn_u[e], d.dn_v[e], d.dn_d[e], o);
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

__global__ __launch_bounds__(256) void k_dn_schur(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= d.dn_nblk || lane >= 36) return;
    const uint32_t a = d.dn_blk_a[blk], b = d.dn_blk_b[blk];
    const int r = lane / 6, c = lane - r * 6;
    double v = 0.0;
    for (uint32_t i = d.dn_blk_start[blk]; i < d.dn_blk_start[blk + 1]; ++i) {
        const double *Y = d.dn_Y + (size_t)d.dn_pair_a[i] * 18 + 3 * r, *W = d.dn_W + (size_t)d.dn_pair_b[i] * 18 + 3 * c;
        v += Y[0] * W[0] + Y[1] * W[1] + Y[2] * W[2];
    }
    v = -v;
    const size_t lda = (size_t)d.dn_pad;
    if (a == b) {
        if (c < r) return;    // (r, c) with r <= c stands for the symmetric pair; stored at (row 6a+c, col 6a+r)
        v += d.hpp[(size_t)d.free_pose[a] * 21 + dn_tri21(r, c)];
    }
    d.dn_S[((size_t)b * 6 + c) * lda + (size_t)a * 6 + r] = v;     // block (b, a) of the lower triangle = block (a, b)^T
}

__global__ __launch_bounds__(256) void k_dn_rhs(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= d.nfree) return;
    const int k = d.free_pose[f];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = d.dn_pose_start[k] + lane; i < d.dn_pose_start[k + 1]; i += 64) {
        const uint32_t e = d.dn_pose_obs[i];
        const int l = (int)d.dn_obs_lm[e];
        const double g0 = d.gl[l], g1 = d.gl[(size_t)d.Lpad + l], g2 = d.gl[2 * (size_t)d.Lpad + l];
        const double *Y = d.dn_Y + (size_t)e * 18;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] += Y[3 * c] * g0 + Y[3 * c + 1] * g1 + Y[3 * c + 2] * g2;
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
__global__ __launch_bounds__(256) void k_dn_potrf(Dev d, int j) {
    State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double t[DN_BS * TP];
    __shared__ int bad;
    const size_t lda = (size_t)d.dn_pad;
    double *A = d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS;
    const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;
    for (int m = q; m < DN_BS; m += 4) t[m * TP + r] = A[(size_t)m * lda + r];      // t[row m][col r]
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int c = 0; c < DN_BS; ++c) {
        const double piv = t[c * TP + c];
        if (!(piv > 0.0) || !isfinite(piv)) { if (tid == 0) bad = 1; break; }      // uniform: every thread reads the same pivot
        const double inv = 1.0 / sqrt(piv);
        __syncthreads();
        if (tid == c) t[c * TP + c] = sqrt(piv);
        else if (q == 0 && r > c) t[r * TP + c] *= inv;
        __syncthreads();
        // trailing update of the lower triangle: rows r > c, columns c < cc <= r
        if (r > c)
            for (int cc = c + 1 + q; cc <= r; cc += 4) t[r * TP + cc] -= t[r * TP + c] * t[cc * TP + c];
        __syncthreads();
    }
    __syncthreads();
    if (bad) { if (tid == 0) st.step_failed = 1; return; }
    for (int m = q; m < DN_BS; m += 4) A[(size_t)m * lda + r] = r <= m ? t[m * TP + r] : 0.0;
}

// rows of the panel below the diagonal block: X = A L_jj^-T, one thread per row, right-looking substitution
__global__ __launch_bounds__(64) void k_dn_trsm(Dev d, int j) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double Lt[DN_BS * DN_BS];       // read as broadcasts only: no padding needed
    const size_t lda = (size_t)d.dn_pad;
    const int i = j + 1 + blockIdx.x, tid = threadIdx.x;
    const double *Lj = d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS;
    double *A = d.dn_S + ((size_t)i * DN_BS + tid) * lda + (size_t)j * DN_BS;     // this thread's row: 512 contiguous bytes
    for (int m = 0; m < DN_BS; ++m) Lt[m * DN_BS + tid] = Lj[(size_t)m * lda + tid];
    double x[DN_BS];
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) x[c] = A[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) {
        x[c] = x[c] / Lt[c * DN_BS + c];
#pragma unroll
        for (int cc = c + 1; cc < DN_BS; ++cc) x[cc] -= x[c] * Lt[cc * DN_BS + c];
    }
#pragma unroll
    for (int c = 0; c < DN_BS; ++c) A[c] = x[c];
}

// trailing update A_ik -= L_ij L_kj^T for j < k <= i: one 64x64 tile per work-group, 4x4 outputs per thread
__global__ __launch_bounds__(256) void k_dn_syrk(Dev d, int j) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    const int k = j + 1 + blockIdx.x, i = j + 1 + blockIdx.y;
    if (i < k) return;
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

// x_i = L_ii^-T (y_i - sum_{j > i} L_ji^T x_j); y and x live in the rhs row
__global__ __launch_bounds__(256) void k_dn_bwd(Dev d, int i) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double t[DN_BS * TP], part[4][DN_BS];
    const size_t lda = (size_t)d.dn_pad;
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    double *xrow = d.dn_S + (size_t)d.dn_pad * lda;
    const double *Lc = d.dn_S + (size_t)i * DN_BS;       // column panel of block column i
    double s = 0.0;
    for (int r = (i + 1) * DN_BS + q; r < d.dn_pad; r += 4) s += Lc[(size_t)r * lda + c] * xrow[r];
    part[q][c] = s;
    const double *Li = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)i * DN_BS;
    for (int m = q; m < DN_BS; m += 4) t[m * TP + c] = Li[(size_t)m * lda + c];
    __syncthreads();
    if (q != 0) return;
    double v = xrow[i * DN_BS + c] - (part[0][c] + part[1][c] + part[2][c] + part[3][c]);
    // one wave: lane c owns x_c; columns of L^T are rows of L
    for (int m = DN_BS - 1; m >= 0; --m) {
        double xm = 0.0;
        if (c == m) xm = v / t[m * TP + m];
        xm = __shfl(xm, m, 64);
        if (c == m) v = xm;
        else if (c < m) v -= t[m * TP + c] * xm;
    }
    const int g = i * DN_BS + c;
    xrow[g] = v;
    if (g < d.n_dn) d.x0[g] = v;
}

// ----------------------------------------------------------------- launchers ---
void launch_dense_schur(Launcher &L, const Dev &d) {
    if (d.dn_pad > 0) hipMemsetAsync(d.dn_S, 0, (size_t)(d.dn_pad + DN_BS) * d.dn_pad * sizeof(double), L.stream);
    LAUNCH(KC_SCHUR, k_dn_wy, dim3((d.n_obs + 255) / 256), dim3(256), 0, d);
    LAUNCH(KC_SCHUR, k_dn_schur, dim3((d.dn_nblk + 3) / 4), dim3(256), 0, d);
    LAUNCH(KC_ASSEMBLE, k_dn_rhs, dim3((d.nfree + 3) / 4), dim3(256), 0, d);
}

void launch_dense_finish(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_dn_finish, dim3((d.dn_pad + 255) / 256), dim3(256), 0, d);
}

void launch_dense_solve(Launcher &L, const Dev &d) {
    const int nbk = d.dn_pad / DN_BS;
    for (int j = 0; j < nbk; ++j) {
        LAUNCH(KC_BCR_FACTOR, k_dn_potrf, dim3(1), dim3(256), 0, d, j);
        LAUNCH(KC_BCR_FACTOR, k_dn_trsm, dim3(nbk - j), dim3(64), 0, d, j);        // block rows j+1 .. nbk (the rhs row)
        LAUNCH(KC_BCR_REDUCE, k_dn_syrk, dim3(nbk - 1 - j, nbk - j), dim3(256), 0, d, j);
    }
    for (int i = nbk - 1; i >= 0; --i) LAUNCH(KC_BCR_BACKSUB, k_dn_bwd, dim3(1), dim3(256), 0, d, i);
}

}  // namespace ssba

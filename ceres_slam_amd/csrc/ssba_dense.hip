// General-structure path of the solve for gfx950: problems whose landmark tracks are longer than the TW slots of the
// window layout, whose pose co-visibility is not banded (loop closures), or that carry what has no place in that
// layout (a second residual block on one (pose, landmark) pair, per-block stiffness, relative-pose blocks) cannot use
// the block-tridiagonal reduced system of ssba_kernels.hip / ssba_bcr.hip.  They keep every other kernel
// (linearisation, back-substitution, trust-region control: templates on the observation layout) and swap the
// middle of the iteration for
//   k_dn_wy      per observation: Z = W M^T with W = J_p^T J_l and C^-1 = M^T M (C = H_ll + damping), stored once (HBM stream:
//                W C^-1 W^T = Z Z^T, one factor for both sides of a pair), records in POSE-major order (dn_zpos);
//                with lighting terms k_ph_dn_wy (ssba_phong_solver.hip) stores the 6x6 version
//   k_dn_schur   one wave per 6x6 block (a <= b) of S = H_pp - sum_l Z_a Z_b^T: the block's observation pairs
//                (host-built list of record positions) spread over the lanes, fixed-order reduction (no float atomics),
//                plus the J_a^T J_b of relative-pose blocks; writes the lower triangle of the dense matrix
//   k_dn_rhs     one wave per free pose: reduced gradient g_p - sum Z (M g_l), stored as an extra row of the matrix
//   k_dn_finish  Jacobi scale at iteration 0, LM damping on the diagonal
//   k_dn_potrf / k_dn_trsm_mf / k_dn_syrk_mf   right-looking blocked Cholesky (DN_BS = 64): the diagonal block by one wave
//                (rows in registers, v_readlane), the panel solve and the trailing update on the fp64 matrix cores
//                (v_mfma_f64_16x16x4_f64; the panel solve blocked by 16 with explicit inverses of the diagonal 16 x 16
//                blocks).  The extra rows (the right-hand side, the columns of S_pb for free shared blocks, unit vectors
//                for a covariance block) are forward-solved as part of the factorisation.  k_dn_trsm / k_dn_syrk: the
//                same two steps on the fp64 VALU (r01; SSBA_DENSE_VALU=1, the A/B partner)
//   k_dn_bwd_all block back-substitution with L^T, left-looking, the whole sweep of one extra row in ONE work-group (all
//                rows in one launch) -> pose step x0, S_pp^-1 S_pb.  k_dn_bwd: one launch per block column (r01)
// The factorisation skips zero 64x64 blocks: ssba_finalize runs a symbolic Cholesky at block granularity, so a
// banded problem costs O(n b^2) and a loop closure only fills the block rows between its two ends.
// What Ceres does here (SPARSE_SCHUR / DENSE_SCHUR on the reduced camera matrix, schur_complement_solver.cc)
// is the same elimination order: landmarks first, then one Cholesky of S.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

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
    // C^-1 = M^T M (M = L^-1 of the damped block): ONE factor Z = W M^T per observation serves both sides of every pair
    // product, W_a C^-1 W_b^T = Z_a Z_b^T -- half the bytes of the (W, Y = W C^-1) pair the r01 kernel stored, and with
    // them the gathers of k_dn_schur fit the 256 MB of MALL at the sizes this path is meant for
    double h[6], dmp[3], m[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
    landmark_damping(d, st, l, h, dmp);
    if (!chol3_inv_fast(h, dmp, m)) {
        d.st->step_failed = 1;
#pragma unroll
        for (int c = 0; c < 6; ++c) m[c] = 0.0;
    }
    {       // u = M g_l for k_dn_rhs (every observation of the landmark writes the same three values)
        const double g0 = d.gl[l], g1 = d.gl[(size_t)d.Lpad + l], g2 = d.gl[2 * (size_t)d.Lpad + l];
        d.dn_Mg[l] = m[0] * g0;
        d.dn_Mg[(size_t)d.Lpad + l] = m[1] * g0 + m[2] * g1;
        d.dn_Mg[2 * (size_t)d.Lpad + l] = m[3] * g0 + m[4] * g1 + m[5] * g2;
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
    double *Z = d.dn_Y + (size_t)d.dn_zpos[e] * 18;       // pose-major record (d.dn_W is the same buffer: ssba_finalize)
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) w[c] = Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c] + Jp[12 + a] * Jl[6 + c];
        Z[3 * a + 0] = w[0] * m[0];
        Z[3 * a + 1] = w[0] * m[1] + w[1] * m[2];
        Z[3 * a + 2] = w[0] * m[3] + w[1] * m[4] + w[2] * m[5];
    }
}

// One work-group per block.  Every thread takes every NT-th observation pair of the block and accumulates the whole
// 6x6 product Y_a W_b^T in registers; the 36 partial sums are then reduced over the wave by shuffles and over the waves
// through LDS, both in a fixed order.  The kernel is bound by the two gathers per pair: ONE wave per block with three
// pairs in flight per lane beats four waves with one (P = 600, tracks of 24: 1.41 against 1.91 ms; 64 x 1: 1.54,
// 128 x 2: 1.49, 64 x 4: 1.41) -- the cross-wave reduction and the workgroup's tail cost more than the parallelism brings.
template <int LD, int NT, int CH> __global__ __launch_bounds__(NT) void k_dn_schur(Dev d) {      // LD = 3 (position) or 6 (position + normal)
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double part[NT / 64][36];
    const int blk = blockIdx.x;
    const uint32_t a = d.dn_blk_a[blk], b = d.dn_blk_b[blk];
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    // CH pairs per round: their two references, then their 2 x 6 LD operands are in flight together
    const uint32_t ie = d.dn_blk_start[blk + 1];
    for (uint32_t i0 = d.dn_blk_start[blk] + threadIdx.x; i0 < ie; i0 += NT * CH) {
        uint32_t pa[CH], pb[CH];
        double y[CH][6 * LD], w[CH][6 * LD];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool on = i0 + NT * u < ie;
            pa[u] = on ? d.dn_pair_a[i0 + NT * u] : 0xFFFFFFFFu;
            pb[u] = on ? d.dn_pair_b[i0 + NT * u] : 0u;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (pa[u] == 0xFFFFFFFFu) continue;
            const double2 *Y2 = reinterpret_cast<const double2 *>(d.dn_Y + (size_t)pa[u] * (6 * LD));
            const double2 *W2 = reinterpret_cast<const double2 *>(d.dn_W + (size_t)pb[u] * (6 * LD));
#pragma unroll
            for (int q = 0; q < 3 * LD; ++q) {
                const double2 yv = Y2[q], wv = W2[q];
                y[u][2 * q] = yv.x; y[u][2 * q + 1] = yv.y; w[u][2 * q] = wv.x; w[u][2 * q + 1] = wv.y;
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (pa[u] == 0xFFFFFFFFu) continue;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = acc[6 * r + c];
#pragma unroll
                    for (int m = 0; m < LD; ++m) v += y[u][LD * r + m] * w[u][LD * c + m];
                    acc[6 * r + c] = v;
                }
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
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v -= part[w][el];
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
        for (int m = 0; m < LD; ++m) g[m] = d.dn_Mg[(size_t)m * d.Lpad + l];      // Y holds Z = W M^T, so M g_l goes with it
        const double *Y = d.dn_Y + (size_t)i * (6 * LD);       // pose-major: entry i of the pose's list
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

// ---- the same two steps on the fp64 matrix cores ---------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15].
// A 64 x 64 block is a 4 x 4 grid of 16 x 16 tiles; wave w of a 256-thread work-group owns tile row w.  Operands sit
// row-major in LDS with a row stride of 68 doubles: a lane reads element (16 t + (lane & 15), 4 ks + (lane >> 4)), i.e.
// bank (8 (lane & 15) + 2 (lane >> 4)) mod 64 -- every even bank exactly twice per 64-lane read, the minimum for 8-byte reads.
typedef double dn_d4 __attribute__((ext_vector_type(4)));
constexpr int DN_LS = 68;
constexpr int DN_MF_THREADS = 256;

// 64 x 64 doubles, row-major with stride lda in global memory -> LDS (stride DN_LS), coalesced 16-byte loads
static __device__ __forceinline__ void dn_stage(double *__restrict__ dst, const double *__restrict__ src, size_t lda) {
#pragma unroll
    for (int q = 0; q < (DN_BS * DN_BS / 2) / DN_MF_THREADS; ++q) {
        const int idx = q * DN_MF_THREADS + (int)threadIdx.x, row = idx / (DN_BS / 2), c2 = idx - row * (DN_BS / 2);
        const double2 v = *reinterpret_cast<const double2 *>(src + (size_t)row * lda + 2 * c2);
        *reinterpret_cast<double2 *>(dst + row * DN_LS + 2 * c2) = v;
    }
}

// trailing update A_ik -= L_ij L_kj^T on the matrix cores: one 64 x 64 tile per work-group, both panels staged in LDS,
// wave w accumulates tile row w (four accumulators, 16 steps of k = 4)
__global__ __launch_bounds__(DN_MF_THREADS) void k_dn_syrk_mf(Dev d, int j, const uint32_t *ti, const uint32_t *tk) {
    const State &st = *d.st;
    const int i = (int)ti[blockIdx.x], k = (int)tk[blockIdx.x];
    extern __shared__ __align__(16) double dn_lds[];
    double *sA = dn_lds, *sB = dn_lds + DN_BS * DN_LS;
    const size_t lda = (size_t)d.dn_pad;
    // (a dead launch reads panels that hold no factor -- harmless -- and leaves before it writes)
    dn_stage(sA, d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS, lda);
    if (k != i) dn_stage(sB, d.dn_S + ((size_t)k * DN_BS) * lda + (size_t)j * DN_BS, lda);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    // the tile to be updated is read while the panels are still on their way (all three come from HBM: the panels were
    // written on other XCDs a launch ago; read after the product, this round trip alone was 6 of the launch's 15.7 us)
    double *C = d.dn_S + ((size_t)i * DN_BS + 16 * w + kq) * lda + (size_t)k * DN_BS + li;
    dn_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -C[(size_t)(4 * r) * lda + 16 * t];
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __syncthreads();
    const double *pB = k != i ? sB : sA;
    const double *pa = sA + (16 * w + li) * DN_LS + kq, *pb = pB + li * DN_LS + kq;
#pragma unroll 4
    for (int ks = 0; ks < DN_BS / 4; ++ks) {
        const double a = pa[4 * ks];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[16 * t * DN_LS + 4 * ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(4 * r) * lda + 16 * t] = -acc[t][r];       // -( -C + L L^T ) = C - L L^T
}

// rows of the panel below the diagonal block, X = A L_jj^-T, blocked by 16 on the matrix cores:
//   X_b = (A_b - sum_{c < b} X_c L_bc^T) inv(L_bb)^T,   b = 0 .. 3,
// with the four 16 x 16 diagonal inverses formed first (wave w inverts block w: lane c < 16 solves L_ww x = e_c by forward
// substitution, broadcast reads of L from LDS).  Wave w owns the 16 rows of tile row w; a result tile goes through LDS to
// change from the accumulator layout to the A-operand layout (wave-private rows, but the barrier keeps the code simple).
__global__ __launch_bounds__(DN_MF_THREADS) void k_dn_trsm_mf(Dev d, int j, const uint32_t *rows) {
    const State &st = *d.st;
    extern __shared__ __align__(16) double dn_lds[];
    double *sX = dn_lds, *sL = dn_lds + DN_BS * DN_LS, *sI = sL + DN_BS * DN_LS;       // A -> X in place | L_jj | the four inverses [b][r][c], stride DN_LS/4... see below
    const size_t lda = (size_t)d.dn_pad;
    const int i = (int)rows[blockIdx.x];
    dn_stage(sX, d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS, lda);
    dn_stage(sL, d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS, lda);
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    // inverse of diagonal block w: column li (lanes 0..15), stored row-major at sI[(16 w + r) * 20 + c]
    constexpr int IS = 20;      // row stride of an inverse block (16 + 4: keeps the operand reads below conflict-free in pairs)
    if (lane < 16) {
        const double *Lw = sL + (16 * w) * DN_LS + 16 * w;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double sum = r == li ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < r; ++c) sum -= Lw[r * DN_LS + c] * x[c];      // x[c] = 0 for c < li: the products vanish
            x[r] = r >= li ? sum / Lw[r * DN_LS + r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sI[(16 * w + r) * IS + li] = x[r];
    }
    __syncthreads();
    for (int b = 0; b < 4; ++b) {
        // T = A_b - sum_{c<b} X_c L_bc^T   (rows of wave w)
        dn_d4 t4;
        {
            const double *src = sX + (16 * w + kq) * DN_LS + 16 * b + li;
            t4 = dn_d4{src[0], src[4 * DN_LS], src[8 * DN_LS], src[12 * DN_LS]};
        }
        for (int c = 0; c < b; ++c) {
            const double *pa = sX + (16 * w + li) * DN_LS + 16 * c + kq;             // X_c as the A operand
            const double *pb = sL + (16 * b + li) * DN_LS + 16 * c + kq;             // B[k][jj] = L[16 b + jj][16 c + k]
            dn_d4 neg = dn_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) neg = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], neg, 0, 0, 0);
            t4 -= neg;
        }
        __syncthreads();       // every wave has read column block b of its rows (and the finished blocks) before it is overwritten
        {
            double *dst = sX + (16 * w + kq) * DN_LS + 16 * b + li;
            dst[0] = t4[0]; dst[4 * DN_LS] = t4[1]; dst[8 * DN_LS] = t4[2]; dst[12 * DN_LS] = t4[3];
        }
        __syncthreads();
        // X_b = T inv(L_bb)^T:  A = T (from LDS), B[k][jj] = inv[jj][k]
        const double *pa = sX + (16 * w + li) * DN_LS + 16 * b + kq;
        const double *pb = sI + (16 * b + li) * IS + kq;
        dn_d4 x4 = dn_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x4 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], x4, 0, 0, 0);
        __syncthreads();
        {
            double *dst = sX + (16 * w + kq) * DN_LS + 16 * b + li;
            dst[0] = x4[0]; dst[4 * DN_LS] = x4[1]; dst[8 * DN_LS] = x4[2]; dst[12 * DN_LS] = x4[3];
        }
        __syncthreads();
    }
    // X -> global, coalesced rows
    double *A = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS;
#pragma unroll
    for (int q = 0; q < (DN_BS * DN_BS / 2) / DN_MF_THREADS; ++q) {
        const int idx = q * DN_MF_THREADS + (int)threadIdx.x, row = idx / (DN_BS / 2), c2 = idx - row * (DN_BS / 2);
        *reinterpret_cast<double2 *>(A + (size_t)row * lda + 2 * c2) = *reinterpret_cast<const double2 *>(sX + row * DN_LS + 2 * c2);
    }
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

// The whole back-substitution of one right-hand-side row in ONE work-group (one per row, all rows in one launch),
// left-looking: for block column j = nbk - 1 .. 0,  y_j -= sum_{i > j} L_ij^T x_i  over the non-zero blocks of column j
// (rows[] of the factorisation plan; the x_i are final), then x_j = L_jj^-T y_j by wave 0.  A banded problem has ~3
// blocks per column: ~5 us per step against a launch (~18 us) per step of k_dn_bwd -- the chain of nbk dependent launches
// per row was a sixth of a general-structure iteration.
__global__ __launch_bounds__(256) void k_dn_bwd_all(Dev d, int nbk) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed || st.dl_reuse) return;
    __shared__ double part[4][DN_BS];
    const size_t lda = (size_t)d.dn_pad;
    const int row = blockIdx.x, tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    double *xrow = d.dn_S + (size_t)(d.dn_pad + row) * lda;
    for (int j = nbk - 1; j >= 0; --j) {
        double s = 0.0;
        const uint32_t r0 = d.dn_row_start[j], r1 = d.dn_row_start[j + 1] - 1;       // the last entry is the right-hand-side block row
        // the factor is read once, from HBM, by this one work-group: all loads of a block are issued before the first is
        // used (a rolled loop pays a memory round trip per row).  Wave 0 keeps column c of L_jj in registers for the solve.
        const double *Lj = d.dn_S + ((size_t)j * DN_BS) * lda + (size_t)j * DN_BS;
        double tc[DN_BS];
        if (q == 0) {
#pragma unroll
            for (int m = 0; m < DN_BS; ++m) tc[m] = Lj[(size_t)m * lda + c];
        }
        for (uint32_t e = r0; e < r1; ++e) {
            const int i = (int)d.dn_rows[e];
            const double *Lij = d.dn_S + ((size_t)i * DN_BS) * lda + (size_t)j * DN_BS;
            const double *xi = xrow + (size_t)i * DN_BS;
            double lv[DN_BS / 4], xv[DN_BS / 4];
#pragma unroll
            for (int m = 0; m < DN_BS / 4; ++m) { lv[m] = Lij[(size_t)(q + 4 * m) * lda + c]; xv[m] = xi[q + 4 * m]; }
#pragma unroll
            for (int m = 0; m < DN_BS / 4; ++m) s += lv[m] * xv[m];
        }
        part[q][c] = s;
        __syncthreads();
        if (q == 0) {
            double v = xrow[j * DN_BS + c] - (part[0][c] + part[1][c] + part[2][c] + part[3][c]);
            // one wave: lane c owns x_c and column c of L_jj (= row c of L_jj^T); the finished x_m travels by v_readlane
            double rdiag = 1.0;
#pragma unroll
            for (int m = 0; m < DN_BS; ++m) rdiag = c == m ? 1.0 / tc[m] : rdiag;
#pragma unroll
            for (int m = DN_BS - 1; m >= 0; --m) {
                const double xm = lane_value(v * rdiag, m);
                v = c == m ? xm : (c < m ? v - tc[m] * xm : v);
            }
            const int g = j * DN_BS + c;
            if (g < d.n_dn && row == 0) d.x0[g] = v;
            xrow[j * DN_BS + c] = v;
            __threadfence_block();
        }
        __syncthreads();
    }
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

// zero-fill of the structurally non-zero tiles (the factorisation works in place, so they hold the last factor); every
// other tile of the dense array is never written and keeps the zeros it was allocated with
__global__ __launch_bounds__(256) void k_dn_zero_tiles(Dev d) {
    const uint32_t t = d.dn_ztile[blockIdx.x];
    const size_t lda = (size_t)d.dn_pad;
    double *T = d.dn_S + ((size_t)(t >> 16) * DN_BS) * lda + (size_t)(t & 0xFFFFu) * DN_BS;
    for (int i = threadIdx.x; i < DN_BS * DN_BS / 2; i += 256) {
        const int row = i / (DN_BS / 2), c2 = i - row * (DN_BS / 2);
        *reinterpret_cast<double2 *>(T + (size_t)row * lda + 2 * c2) = make_double2(0.0, 0.0);
    }
}

// ----------------------------------------------------------------- launchers ---
constexpr size_t DN_SYRK_LDS = (size_t)2 * DN_BS * DN_LS * sizeof(double);                       // 69 632 B
constexpr size_t DN_TRSM_LDS = (size_t)(2 * DN_BS * DN_LS + DN_BS * 20) * sizeof(double);       // 79 872 B
int configure_dense() {
    if (hipFuncSetAttribute((const void *)k_dn_syrk_mf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_SYRK_LDS) != hipSuccess) return -1;
    return hipFuncSetAttribute((const void *)k_dn_trsm_mf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DN_TRSM_LDS) == hipSuccess ? 0 : -1;
}
void launch_dense_schur(Launcher &L, const Dev &d, bool fuse_finish) {
    if (d.wide) { launch_wide_schur(L, d, fuse_finish); return; }       // banded, tracks of <= 24 observations: 144-row super-blocks (ssba_wide.hip)
    LAUNCH(KC_SMALL, k_dn_zero_tiles, dim3(d.dn_nztile), dim3(256), 0, d);
    if (d.phong) {       // 6-D landmark blocks: C^-1 first, W / Y are 6x6 (ssba_phong_solver.hip)
        launch_ph_dense_wy(L, d);
        LAUNCH(KC_SCHUR, (k_dn_schur<6, 64, 2>), dim3(d.dn_nblk), dim3(64), 0, d);
        LAUNCH(KC_ASSEMBLE, k_dn_rhs<6>, dim3((d.nfree + 3) / 4), dim3(256), 0, d);
        if (d.nb) launch_ph_dense_border(L, d);
        return;
    }
    LAUNCH(KC_SCHUR, k_dn_wy, dim3((d.n_obs + 255) / 256), dim3(256), 0, d);
    LAUNCH(KC_SCHUR, (k_dn_schur<3, 64, 3>), dim3(d.dn_nblk), dim3(64), 0, d);
    LAUNCH(KC_ASSEMBLE, k_dn_rhs<3>, dim3((d.nfree + 3) / 4), dim3(256), 0, d);
}

void launch_dense_finish(Launcher &L, const Dev &d, bool fused) {
    if (d.wide) { launch_wide_finish(L, d, fused); return; }
    LAUNCH(KC_SMALL, k_dn_finish, dim3((d.dn_pad + 255) / 256), dim3(256), 0, d);
}

void launch_dense_solve(Launcher &L, const Dev &d, int n_rhs_rows) {
    if (d.wide) { launch_wide_solve(L, d); return; }       // (a wide handle never gets here with extra right-hand-side rows: ssba_api.hip)
    const DensePlan &pl = L.dense;
    const int nbk = pl.nbk;
    // SSBA_DENSE_VALU=1: the panel solve and the trailing update on the fp64 VALU (the r01 kernels; A/B partner, tests)
    const char *lv = getenv("SSBA_DENSE_VALU");
    const bool legacy = lv && lv[0] == '1';
    const bool border = d.nb > 0 && n_rhs_rows == 1;      // the solve of an iteration (not the covariance's unit rows)
    if (border) {
        LAUNCH(KC_BORDER, k_dn_border_rows, dim3((d.n_dn + 255) / 256), dim3(256), 0, d, 0);
        n_rhs_rows = 1 + d.nb;
    }
    for (int j = 0; j < nbk; ++j) {
        const uint32_t r0 = pl.row_start[j], nr = pl.row_start[j + 1] - r0, t0 = pl.tile_start[j], nt = pl.tile_start[j + 1] - t0;
        LAUNCH(KC_BCR_FACTOR, k_dn_potrf, dim3(1), dim3(64), 0, d, j);
        if (legacy) {
            LAUNCH(KC_BCR_FACTOR, k_dn_trsm, dim3(nr), dim3(64), 0, d, j, d.dn_rows + r0);
            LAUNCH(KC_BCR_REDUCE, k_dn_syrk, dim3(nt), dim3(256), 0, d, j, d.dn_ti + t0, d.dn_tk + t0);
        } else {
            LAUNCH(KC_BCR_FACTOR, k_dn_trsm_mf, dim3(nr), dim3(DN_MF_THREADS), DN_TRSM_LDS, d, j, d.dn_rows + r0);
            LAUNCH(KC_BCR_REDUCE, k_dn_syrk_mf, dim3(nt), dim3(DN_MF_THREADS), DN_SYRK_LDS, d, j, d.dn_ti + t0, d.dn_tk + t0);
        }
    }
    if (!legacy) LAUNCH(KC_BCR_BACKSUB, k_dn_bwd_all, dim3(n_rhs_rows), dim3(256), 0, d, nbk);
    else
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

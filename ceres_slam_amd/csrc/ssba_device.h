// Device functions shared by the solver kernels (ssba_kernels.hip, ssba_phong_solver.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "ssba_types.h"

namespace ssba {

// ------------------------------------------------------------------ helpers ---
// The words of the solver state every kernel tests first ("terminated", "step failed", "reuse the dogleg data") were
// written by the previous launch, usually on another XCD: a cold read of a microsecond or two.  Through the scalar path
// that latency sits in front of everything, because scalar loads return out of order and the first `s_waitcnt lgkmcnt(0)`
// -- the one for the kernel arguments the addresses are computed from -- waits for the state words too.  Through the
// VECTOR path the read has its own counter (vmcnt, in order): issued first, it is waited for where its value is used, with
// the operand loads of the kernel already in flight behind it.  mbcnt(0, 0) is zero in every lane but opaque to the
// compiler, which therefore emits a per-lane global_load.
static __device__ __forceinline__ int state_word_vmem(const int *p) { return p[__builtin_amdgcn_mbcnt_lo(0u, 0u)]; }
struct StateFlags {
    int terminated, step_failed, dl_reuse;
    __device__ __forceinline__ int dead() const { return terminated | step_failed | dl_reuse; }
};
static __device__ __forceinline__ StateFlags state_flags_vmem(const State *st) {
    StateFlags f;
    f.terminated = state_word_vmem(&st->terminated);
    f.step_failed = state_word_vmem(&st->step_failed);
    f.dl_reuse = state_word_vmem(&st->dl_reuse);
    return f;
}

// Wave reductions on the DPP path (row shifts inside the rows of 16 lanes, then row_bcast:15 / :31): a fixed tree like the
// __shfl_down one they replace (r04), but a shuffle of a double is two ds_bpermute_b32 through the LDS crossbar per step --
// tools/proto/dpp_reduce_test.hip: 2.1 x the throughput, and no LDS round trip on the dependent chain of the single-work-group
// control kernels.  The total arrives in lane 63 and is broadcast: every lane returns it.
template <int CTRL, int ROW_MASK> static __device__ __forceinline__ double wave_dpp(double v, double identity) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
static __device__ __forceinline__ double wave_sum(double v) {
    v += wave_dpp<0x111, 0xf>(v, 0.0);      // row_shr:1
    v += wave_dpp<0x112, 0xf>(v, 0.0);      // row_shr:2
    v += wave_dpp<0x114, 0xf>(v, 0.0);      // row_shr:4
    v += wave_dpp<0x118, 0xf>(v, 0.0);      // row_shr:8: lane 15 of every row holds the row's sum
    v += wave_dpp<0x142, 0xa>(v, 0.0);      // row_bcast:15 into rows 1 and 3
    v += wave_dpp<0x143, 0xc>(v, 0.0);      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return wave_lane63(v);
}
static __device__ __forceinline__ double wave_max(double v) {       // (of non-negative values or with -inf as the floor: identity = v itself)
    v = fmax(v, wave_dpp<0x111, 0xf>(v, v));
    v = fmax(v, wave_dpp<0x112, 0xf>(v, v));
    v = fmax(v, wave_dpp<0x114, 0xf>(v, v));
    v = fmax(v, wave_dpp<0x118, 0xf>(v, v));
    v = fmax(v, wave_dpp<0x142, 0xa>(v, v));
    v = fmax(v, wave_dpp<0x143, 0xc>(v, v));
    return wave_lane63(v);
}
// Deterministic block sum; result valid on thread 0.  sm needs blockDim/64 doubles.
static __device__ __forceinline__ double block_sum(double v, double *sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) t += sm[i];
    return t;
}
// N sums at once: one pair of barriers instead of N (the single-work-group control kernels are chains of these).  sm needs
// N * blockDim/64 doubles; results valid on thread 0.
template <int N> static __device__ __forceinline__ void block_sums(double (&v)[N], double *sm) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) sm[q * nw + w] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            double t = 0.0;
            for (int i = 0; i < nw; ++i) t += sm[q * nw + i];
            v[q] = t;
        }
    }
}
static __device__ __forceinline__ double block_max(double v, double *sm) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) t = fmax(t, sm[i]);
    return t;
}

// Work-group -> item for launches whose neighbouring items share operands (poses that see the same landmarks): work-groups are
// dealt to the eight XCDs round-robin (work-group id mod 8, verified with HW_REG_XCC_ID stamps: profiles/r04_bcr_bench_stamps_n84.txt)
// and every XCD has its own L2, so consecutive items on consecutive work-groups pull every shared line into all eight.  Items go
// to the XCDs in GROUPS of 16 consecutive ones instead (group g on XCD g mod 8).  (One contiguous eighth per XCD shares more, but a
// landmark shard of a multi-GPU run has observations on a contiguous seventh of the poses only: all its work landed on one XCD,
// 0.537 -> 0.609 ms per iteration of rank 4 of 8.)  Launch xcd_grouped_grid(n) work-groups; returns -1 for the padding.
constexpr int XCD_GROUP = 16;
static __device__ __forceinline__ int xcd_contiguous_item(int wg, int n) {
    const int q = wg >> 3;                                   // position among this XCD's work-groups
    const int item = ((q / XCD_GROUP) * 8 + (wg & 7)) * XCD_GROUP + (q % XCD_GROUP);
    return item < n ? item : -1;
}
static __host__ __device__ __forceinline__ int xcd_contiguous_grid(int n) {
    const int groups = (n + XCD_GROUP - 1) / XCD_GROUP;
    return ((groups + 7) / 8) * 8 * XCD_GROUP;
}

struct ObsLin {
    double r[3];     // loss-corrected residual
    double A[9];     // sqrt(rho') * S * J_pi(q)
    double q[3];     // point in the camera frame
    double half_rho; // 1/2 rho(|r|^2)
};

// stereo_reprojection_error.hpp:38-50, stereo_camera.hpp:77-104, Huber corrector
// [Ceres corrector.cc with rho'' <= 0].  T is the 12-double pose block.
static __device__ __forceinline__ void obs_linearize_S(const Dev &d, const double *__restrict__ S, const double *__restrict__ T,
                                                double px, double py, double pz, double u, double v,
                                                double dd, ObsLin &o) {
    const double q0 = T[3] * px + T[4] * py + T[5] * pz + T[0];
    const double q1 = T[6] * px + T[7] * py + T[8] * pz + T[1];
    const double q2 = T[9] * px + T[10] * py + T[11] * pz + T[2];
    const double iz = 1.0 / q2;
    const double e0 = d.fu * q0 * iz + d.cu - u;
    const double e1 = d.fv * q1 * iz + d.cv - v;
    const double e2 = d.fu * d.b * iz - dd;
    const double j00 = d.fu * iz, j11 = d.fv * iz;
    const double iz2 = iz * iz;
    const double j02 = -d.fu * q0 * iz2, j12 = -d.fv * q1 * iz2, j22 = -d.fu * d.b * iz2;
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double s0 = S[3 * i], s1 = S[3 * i + 1], s2 = S[3 * i + 2];
        o.r[i] = s0 * e0 + s1 * e1 + s2 * e2;
        o.A[3 * i] = s0 * j00;
        o.A[3 * i + 1] = s1 * j11;
        o.A[3 * i + 2] = s0 * j02 + s1 * j12 + s2 * j22;
        sq += o.r[i] * o.r[i];
    }
    o.q[0] = q0; o.q[1] = q1; o.q[2] = q2;
    o.half_rho = 0.5 * sq;
    if (d.huber_a > 0.0 && sq > d.huber_a * d.huber_a) {
        const double rs = sqrt(sq);
        const double rho1 = fmax(DBL_MIN, d.huber_a / rs);
        const double sc = sqrt(rho1);
        o.half_rho = 0.5 * (2.0 * d.huber_a * rs - d.huber_a * d.huber_a);
#pragma unroll
        for (int i = 0; i < 3; ++i) o.r[i] *= sc;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.A[i] *= sc;
    }
}

// the stiffness shared by all stereo residual blocks (every reference driver except dataset_vo_sun)
static __device__ __forceinline__ void obs_linearize(const Dev &d, const double *__restrict__ T,
                                              double px, double py, double pz, double u, double v,
                                              double dd, ObsLin &o) {
    obs_linearize_S(d, d.S, T, px, py, pz, u, v, dd, o);
}

// 1/2 rho(|r|^2) only (candidate evaluation)
static __device__ __forceinline__ double obs_cost_S(const Dev &d, const double *__restrict__ S, const double *__restrict__ T, double px,
                                             double py, double pz, double u, double v, double dd) {
    const double q0 = T[3] * px + T[4] * py + T[5] * pz + T[0];
    const double q1 = T[6] * px + T[7] * py + T[8] * pz + T[1];
    const double q2 = T[9] * px + T[10] * py + T[11] * pz + T[2];
    const double iz = 1.0 / q2;
    const double e0 = d.fu * q0 * iz + d.cu - u;
    const double e1 = d.fv * q1 * iz + d.cv - v;
    const double e2 = d.fu * d.b * iz - dd;
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double r = S[3 * i] * e0 + S[3 * i + 1] * e1 + S[3 * i + 2] * e2;
        sq += r * r;
    }
    if (d.huber_a > 0.0 && sq > d.huber_a * d.huber_a)
        return 0.5 * (2.0 * d.huber_a * sqrt(sq) - d.huber_a * d.huber_a);
    return 0.5 * sq;
}
static __device__ __forceinline__ double obs_cost(const Dev &d, const double *__restrict__ T, double px,
                                           double py, double pz, double u, double v, double dd) {
    return obs_cost_S(d, d.S, T, px, py, pz, u, v, dd);
}

// J_l = A R (3x3)   [dq/dp = R]
static __device__ __forceinline__ void jac_point(const ObsLin &o, const double *__restrict__ T, double Jl[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Jl[3 * i + j] = o.A[3 * i] * T[3 + j] + o.A[3 * i + 1] * T[6 + j] + o.A[3 * i + 2] * T[9 + j];
}
// J_p = A [I | -q^] (3x6)   [dq/deps at eps = 0: perturbations.hpp:61-62, so3group.hpp:277-280]
static __device__ __forceinline__ void jac_pose(const ObsLin &o, double Jp[18]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a0 = o.A[3 * i], a1 = o.A[3 * i + 1], a2 = o.A[3 * i + 2];
        Jp[6 * i + 0] = a0;
        Jp[6 * i + 1] = a1;
        Jp[6 * i + 2] = a2;
        Jp[6 * i + 3] = -a1 * o.q[2] + a2 * o.q[1];
        Jp[6 * i + 4] = a0 * o.q[2] - a2 * o.q[0];
        Jp[6 * i + 5] = -a0 * o.q[1] + a1 * o.q[0];
    }
}

// J_p dp without forming J_p = A [I | -q^]:  A (dp_t + dp_r x q)
static __device__ __forceinline__ void pose_step_rows(const ObsLin &o, const double *__restrict__ dp, double jd[3]) {
    const double w0 = dp[0] + (dp[4] * o.q[2] - dp[5] * o.q[1]);
    const double w1 = dp[1] + (dp[5] * o.q[0] - dp[3] * o.q[2]);
    const double w2 = dp[2] + (dp[3] * o.q[1] - dp[4] * o.q[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) jd[i] = o.A[3 * i] * w0 + o.A[3 * i + 1] * w1 + o.A[3 * i + 2] * w2;
}
// acc[0..20] += upper triangle of J_p^T J_p (row-major: (0,0) (0,1) .. (0,5) (1,1) ..), acc[21..26] += J_p^T r.
// (r04 tried E^T G E with G = A^T A, E = [I | -q^]: 57 multiply-adds instead of 81, -1.5 us at C2.  Its rotation block is a
// difference of products of SUMS where this form adds squares of per-row differences; the undamped covariance test, whose Schur
// complement cancels eight digits, moved from 1.x e-2 to 3.3e-2 -- which later turned out to be within what a change of the
// summation order alone does to that figure.  The Jacobian form stayed: sums of squares on the diagonal.)
static __device__ __forceinline__ void pose_normal_terms(const ObsLin &o, double acc[27]) {
    double Jp[18];
    jac_pose(o, Jp);
    int n = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) {
            acc[n] += Jp[a] * Jp[c] + Jp[6 + a] * Jp[6 + c] + Jp[12 + a] * Jp[12 + c];
            ++n;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += Jp[a] * o.r[0] + Jp[6 + a] * o.r[1] + Jp[12 + a] * o.r[2];
}

// The Schur factor of one stereo observation,  Z = W M^T  (W = J_p^T J_l, 6 x 3; C^-1 = M^T M, M lower triangular
// m00 m10 m11 m20 m21 m22), without forming the Jacobians:  J_p = A [I | -q^],  J_l = A R  give
//     W = [X ; q x X]  with  X = G R,  G = A^T A    =>    Z = [B ; q x B],  B = G R M^T
// -- 81 multiply-adds against the 120 of jac_pose + jac_point + W + Z (r04: the producer phase of the Schur kernels is fp64
// VALU work on the datapath the matrix instructions need).  A = sqrt(rho') S J_pi as obs_linearize_S forms it; the residual
// is only evaluated when a loss is set (it gives rho').  z[6 c + a] = Z[a][c].
static __device__ __forceinline__ double fast_rcp(double a);
static __device__ __forceinline__ void obs_schur_factor(const Dev &d, const double *__restrict__ S, const double *__restrict__ T,
                                                        double px, double py, double pz, double u, double v, double dd,
                                                        const double m[6], double z[18]) {
    const double q0 = T[3] * px + T[4] * py + T[5] * pz + T[0];
    const double q1 = T[6] * px + T[7] * py + T[8] * pz + T[1];
    const double q2 = T[9] * px + T[10] * py + T[11] * pz + T[2];
    const double iz = fast_rcp(q2);         // (the factor only shapes the step, like M's reciprocal square roots; cost and gradient keep the IEEE divide)
    const double j00 = d.fu * iz, j11 = d.fv * iz;
    const double iz2 = iz * iz;
    const double j02 = -d.fu * q0 * iz2, j12 = -d.fv * q1 * iz2, j22 = -d.fu * d.b * iz2;
    double A[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double s0 = S[3 * i], s1 = S[3 * i + 1], s2 = S[3 * i + 2];
        A[3 * i] = s0 * j00;
        A[3 * i + 1] = s1 * j11;
        A[3 * i + 2] = s0 * j02 + s1 * j12 + s2 * j22;
    }
    double w2 = 1.0;        // rho' (the corrector scales A by its square root: G by rho')
    if (d.huber_a > 0.0) {
        const double e0 = d.fu * q0 * iz + d.cu - u, e1 = d.fv * q1 * iz + d.cv - v, e2 = d.fu * d.b * iz - dd;
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r = S[3 * i] * e0 + S[3 * i + 1] * e1 + S[3 * i + 2] * e2;
            sq += r * r;
        }
        if (sq > d.huber_a * d.huber_a) {
            const double sc = sqrt(fmax(DBL_MIN, d.huber_a / sqrt(sq)));     // as obs_linearize_S: A *= sc
#pragma unroll
            for (int i = 0; i < 9; ++i) A[i] *= sc;
        }
    }
    (void)w2;
    const double g00 = A[0] * A[0] + A[3] * A[3] + A[6] * A[6], g01 = A[0] * A[1] + A[3] * A[4] + A[6] * A[7];
    const double g02 = A[0] * A[2] + A[3] * A[5] + A[6] * A[8], g11 = A[1] * A[1] + A[4] * A[4] + A[7] * A[7];
    const double g12 = A[1] * A[2] + A[4] * A[5] + A[7] * A[8], g22 = A[2] * A[2] + A[5] * A[5] + A[8] * A[8];
    double B[9];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double ga0 = a == 0 ? g00 : a == 1 ? g01 : g02, ga1 = a == 0 ? g01 : a == 1 ? g11 : g12, ga2 = a == 0 ? g02 : a == 1 ? g12 : g22;
        const double x0 = ga0 * T[3] + ga1 * T[6] + ga2 * T[9];          // X = G R
        const double x1 = ga0 * T[4] + ga1 * T[7] + ga2 * T[10];
        const double x2 = ga0 * T[5] + ga1 * T[8] + ga2 * T[11];
        B[3 * a] = x0 * m[0];                                             // B = X M^T
        B[3 * a + 1] = x0 * m[1] + x1 * m[2];
        B[3 * a + 2] = x0 * m[3] + x1 * m[4] + x2 * m[5];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double b0 = B[c], b1 = B[3 + c], b2 = B[6 + c];
        z[6 * c] = b0; z[6 * c + 1] = b1; z[6 * c + 2] = b2;
        z[6 * c + 3] = q1 * b2 - q2 * b1;
        z[6 * c + 4] = q2 * b0 - q0 * b2;
        z[6 * c + 5] = q0 * b1 - q1 * b0;
    }
}

// so3group.hpp:273-291 ; perturbations.hpp:61-62 with se3group.hpp:176-183,323-325
static __device__ __forceinline__ void se3_plus(const double *__restrict__ T, const double *__restrict__ eps,
                                         double *__restrict__ out) {
    const double p0 = eps[3], p1 = eps[4], p2 = eps[5];
    const double angle = sqrt(p0 * p0 + p1 * p1 + p2 * p2);
    double E[9];
    if (angle <= DBL_EPSILON) {
        E[0] = 1.0; E[1] = -p2; E[2] = p1;
        E[3] = p2;  E[4] = 1.0; E[5] = -p0;
        E[6] = -p1; E[7] = p0;  E[8] = 1.0;
    } else {
        const double a0 = p0 / angle, a1 = p1 / angle, a2 = p2 / angle;
        const double cp = cos(angle), sn = sin(angle), omc = 1.0 - cp;
        E[0] = cp + omc * a0 * a0;      E[1] = omc * a0 * a1 - sn * a2; E[2] = omc * a0 * a2 + sn * a1;
        E[3] = omc * a1 * a0 + sn * a2; E[4] = cp + omc * a1 * a1;      E[5] = omc * a1 * a2 - sn * a0;
        E[6] = omc * a2 * a0 - sn * a1; E[7] = omc * a2 * a1 + sn * a0; E[8] = cp + omc * a2 * a2;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[i] = E[3 * i] * T[0] + E[3 * i + 1] * T[1] + E[3 * i + 2] * T[2] + eps[i];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            out[3 + 3 * i + j] = E[3 * i] * T[3 + j] + E[3 * i + 1] * T[6 + j] + E[3 * i + 2] * T[9 + j];
    }
}

// inverse of the damped 3x3 landmark block through its Cholesky factor.
// h = (h00,h01,h02,h11,h12,h22), dmp = LM diagonal.  Returns false on breakdown.
static __device__ __forceinline__ bool inv3_spd(const double h[6], const double dmp[3], double Ci[6]) {
    const double c00 = h[0] + dmp[0], c11 = h[3] + dmp[1], c22 = h[5] + dmp[2];
    if (!(c00 > 0.0)) return false;
    const double l00 = sqrt(c00);
    const double l10 = h[1] / l00, l20 = h[2] / l00;
    const double d1 = c11 - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double l11 = sqrt(d1);
    const double l21 = (h[4] - l20 * l10) / l11;
    const double d2 = c22 - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double l22 = sqrt(d2);
    const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
    const double m10 = -l10 * m00 * m11;
    const double m21 = -l21 * m11 * m22;
    const double m20 = -(l20 * m00 + l21 * m10) * m22;
    Ci[0] = m00 * m00 + m10 * m10 + m20 * m20;
    Ci[1] = m10 * m11 + m20 * m21;
    Ci[2] = m20 * m22;
    Ci[3] = m11 * m11 + m21 * m21;
    Ci[4] = m21 * m22;
    Ci[5] = m22 * m22;
    return true;
}

// radius that scales the LM-type diagonal: the trust-region radius for Levenberg-Marquardt, 1/mu for
// the regularised Gauss-Newton solve of the dogleg strategy [dogleg_strategy.cc ComputeGaussNewtonStep]
static __device__ __forceinline__ double damp_radius(const State &st) { return st.opt.strategy ? 1.0 / st.mu : st.radius; }

// LM diagonal of a landmark in unscaled coordinates:
//   D^2 = clamp(s^2 h, min, max) / (radius s^2)   [levenberg_marquardt_strategy.cc on the
//   Jacobi-scaled Jacobian, mapped back through delta = s .* step]
static __device__ __forceinline__ void landmark_damping(const Dev &d, const State &st, int l, const double h[6],
                                                 double dmp[3]) {
    const double hd[3] = {h[0], h[3], h[5]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double s = d.sl[(size_t)c * d.Lpad + l];
        const double s2 = s * s;
        dmp[c] = fmin(fmax(hd[c] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    }
}


// 1/x and 1/sqrt(x) from the hardware estimates + Newton steps (IEEE divide / sqrt cost ~110 / ~150
// dependent cycles on gfx950, tools/fp64_calib.hip); relative error ~1e-16
static __device__ __forceinline__ double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}
static __device__ __forceinline__ double fast_rsqrt(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}
// M = L^-1 (lower triangular: m00, m10, m11, m20, m21, m22) of the damped 3x3 landmark block C = L L^T, so that
// C^-1 = M^T M: W C^-1 W^T = (W M^T)(W M^T)^T is a symmetric product of ONE factor (k_schur_windows)
static __device__ __forceinline__ bool chol3_inv_fast(const double h[6], const double dmp[3], double m[6]) {
    const double c00 = h[0] + dmp[0], c11 = h[3] + dmp[1], c22 = h[5] + dmp[2];
    if (!(c00 > 0.0)) return false;
    const double m00 = fast_rsqrt(c00);
    const double l10 = h[1] * m00, l20 = h[2] * m00;
    const double d1 = c11 - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double m11 = fast_rsqrt(d1);
    const double l21 = (h[4] - l20 * l10) * m11;
    const double d2 = c22 - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double m22 = fast_rsqrt(d2);
    const double m10 = -l10 * m00 * m11;
    m[0] = m00; m[1] = m10; m[2] = m11;
    m[4] = -l21 * m11 * m22;
    m[3] = -(l20 * m00 + l21 * m10) * m22;
    m[5] = m22;
    return true;
}
// inverse of the damped 3x3 landmark block, Cholesky based, reciprocal square roots only
static __device__ __forceinline__ bool inv3_spd_fast(const double h[6], const double dmp[3], double Ci[6]) {
    const double c00 = h[0] + dmp[0], c11 = h[3] + dmp[1], c22 = h[5] + dmp[2];
    if (!(c00 > 0.0)) return false;
    const double m00 = fast_rsqrt(c00);
    const double l10 = h[1] * m00, l20 = h[2] * m00;
    const double d1 = c11 - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double m11 = fast_rsqrt(d1);
    const double l21 = (h[4] - l20 * l10) * m11;
    const double d2 = c22 - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double m22 = fast_rsqrt(d2);
    const double m10 = -l10 * m00 * m11;
    const double m21 = -l21 * m11 * m22;
    const double m20 = -(l20 * m00 + l21 * m10) * m22;
    Ci[0] = m00 * m00 + m10 * m10 + m20 * m20;
    Ci[1] = m10 * m11 + m20 * m21;
    Ci[2] = m20 * m22;
    Ci[3] = m11 * m11 + m21 * m21;
    Ci[4] = m21 * m22;
    Ci[5] = m22 * m22;
    return true;
}


// ------------------------------------------------------------------ dogleg scalars ---
// Real parts of all roots of a polynomial of degree <= 4 (coefficients highest degree first), as
// Ceres's FindPolynomialRoots(p, &real, NULL) hands them to DoglegStrategy [polynomial.cc]: leading
// zeros removed, closed forms for degree 1 and 2, otherwise Aberth-Ehrlich iteration (Ceres takes the
// eigenvalues of the companion matrix: the same roots).  Returns the number of roots, -1 on failure.
static __device__ int poly_roots_real(const double *coef_in, int ncoef, double *re) {
    int lead = 0;
    while (lead < ncoef - 1 && coef_in[lead] == 0.0) ++lead;
    const double *c = coef_in + lead;
    const int deg = ncoef - lead - 1;
    if (deg < 0) return -1;
    if (deg == 0) return 0;
    if (deg == 1) { re[0] = -c[1] / c[0]; return 1; }
    if (deg == 2) {
        const double a = c[0], b = c[1], cc = c[2];
        const double D = b * b - 4 * a * cc, sq = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { re[0] = (-b - sq) / (2.0 * a); re[1] = (2.0 * cc) / (-b - sq); }
            else { re[0] = (2.0 * cc) / (-b + sq); re[1] = (-b + sq) / (2.0 * a); }
        } else {
            re[0] = re[1] = -b / (2.0 * a);
        }
        return 2;
    }
    if (deg > 4) return -1;
    // (ONE lane of a work-group runs this; the arrays are indexed at run time: LDS, not scratch memory)
    __shared__ double m[5], zr[4], zi[4];
    double bound = 0.0, fuji = 0.0;
    for (int i = 0; i <= deg; ++i) {
        m[i] = c[i] / c[0];
        if (!isfinite(m[i])) return -1;
        if (i && fabs(m[i]) > bound) bound = fabs(m[i]);
        // Fujiwara: every root has |z| <= 2 max_k |m_k|^(1/k); the k-th roots bounded from above with square roots alone
        // (as ssba_linesearch.h: ls_root_radius).  Cauchy's 1 + max |m_k| is ~lambda^4 for roots ~lambda here: the start
        // circle was 1e10 x too wide and the iteration spent ~40 sweeps (60-90 us on this one lane) contracting it
        if (i) {
            double x = fabs(m[i]);
            const int nsq = x >= 1.0 ? (i >= 4 ? 2 : i >= 2 ? 1 : 0) : (i >= 3 ? 2 : i >= 2 ? 1 : 0);
            for (int j = 0; j < nsq; ++j) x = sqrt(x);
            fuji = fmax(fuji, x);
        }
    }
    const double radius = fuji > 0.0 ? fmin(1.0 + bound, 2.0 * fuji) : 1.0;
    for (int i = 0; i < deg; ++i) {
        const double ang = 2.0 * 3.14159265358979323846 * i / deg + 0.4;
        zr[i] = radius * cos(ang); zi[i] = radius * sin(ang);
    }
    int polished = 0;
    for (int it = 0; it < 500; ++it) {
        double worst = 0.0;
        for (int i = 0; i < deg; ++i) {
            double pr = m[0], pi = 0.0, dr = 0.0, di = 0.0;     // Horner: value and derivative
            for (int k = 1; k <= deg; ++k) {
                const double ndr = dr * zr[i] - di * zi[i] + pr, ndi = dr * zi[i] + di * zr[i] + pi;
                dr = ndr; di = ndi;
                const double npr = pr * zr[i] - pi * zi[i] + m[k], npi = pr * zi[i] + pi * zr[i];
                pr = npr; pi = npi;
            }
            if (pr == 0.0 && pi == 0.0) continue;
            double rr, ri;                                        // ratio = p / p'
            const double dn = dr * dr + di * di;
            if (dn == 0.0) { rr = 1e-3 * (1.0 + sqrt(zr[i] * zr[i] + zi[i] * zi[i])); ri = 0.0; }
            else { rr = (pr * dr + pi * di) / dn; ri = (pi * dr - pr * di) / dn; }
            double sr = 0.0, si = 0.0;                            // sum 1 / (z_i - z_k)
            for (int k = 0; k < deg; ++k)
                if (k != i) {
                    const double er = zr[i] - zr[k], ei = zi[i] - zi[k], en = er * er + ei * ei;
                    sr += er / en; si -= ei / en;
                }
            const double qr = 1.0 - (rr * sr - ri * si), qi = -(rr * si + ri * sr), qn = qr * qr + qi * qi;
            const double stepr = (rr * qr + ri * qi) / qn, stepi = (ri * qr - rr * qi) / qn;
            zr[i] -= stepr; zi[i] -= stepi;
            const double rel = sqrt(stepr * stepr + stepi * stepi) / (1.0 + sqrt(zr[i] * zr[i] + zi[i] * zi[i]));
            if (rel > worst) worst = rel;
        }
        if (polished) break;                  /* cubic convergence: one sweep after the 1e-13 sweep reaches rounding level */
        if (worst < 1e-13) polished = 1;
    }
    for (int i = 0; i < deg; ++i) {
        if (!isfinite(zr[i])) return -1;
        re[i] = zr[i];
    }
    return deg;
}

// UnitVectorPerturbation::operator() (perturbations.hpp:98-102)
static __device__ __forceinline__ void unit_plus(const double x[3], const double dl[3], double out[3]) {
    const double s = (dl[0] * x[0] + dl[1] * x[1] + dl[2] * x[2]) / (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const double y0 = x[0] + dl[0] - s * x[0], y1 = x[1] + dl[1] - s * x[1], y2 = x[2] + dl[2] - s * x[2];
    const double nrm = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
    out[0] = y0 / nrm; out[1] = y1 / nrm; out[2] = y2 / nrm;
}

// Candidate shared blocks (light, materials, texture) = projected Plus(x, alpha * delta_b) and their part of |dx|^2 / the
// non-finite flag, by the first wave of a 256-lane work-group.  Up to r04 a launch of its own (k_ph_border_update, 5.8 us:
// three per iteration with the device-side line search); now the work-group behind the pose blocks of k_pose_update -- its
// inputs are complete when that launch starts.  ls_round: a round of the device-side search, a no-op unless one is under way.
static __device__ __forceinline__ void ph_border_update_block(const Dev &d, int ls_round, double *sdf /* 64 doubles of LDS */) {
    const State &st = *d.st;
    if (st.terminated || (ls_round && !st.ls_active)) return;
    const int i = threadIdx.x, M3 = 3 * d.M;
    const bool in = i < d.nsh;
    const double old = in ? d.sh[i] : 0.0;
    double nw = old, bad = 0.0;
    const bool moved = !st.step_failed && d.nb;
    if (moved) {
        int col = -1;       // border column of this entry (-1: its block is constant)
        if (i < 3) col = d.b_light >= 0 ? d.b_light + i : -1;
        else if (i < 3 + M3) col = d.b_phong >= 0 ? d.b_phong + (i - 3) : -1;
        else if (in) col = d.b_tex >= 0 ? d.b_tex + (i - 3 - M3) : -1;
        double db = 0.0;    // LM: beta = 1, gamma = 0; dogleg: beta * delta_gn + gamma * v
        if (col >= 0) {
            db = st.ls_alpha * (st.beta * d.bsys[BS_DB + col] + st.gamma * d.bsys[BS_VB + col]);
            if (!isfinite(db)) bad = 1.0;
        }
        const double x3[3] = {__shfl(old, 0, 64), __shfl(old, 1, 64), __shfl(old, 2, 64)};
        const double d3[3] = {__shfl(db, 0, 64), __shfl(db, 1, 64), __shfl(db, 2, 64)};
        if (col >= 0) {
            if (i < 3 && d.light_type == 1) {
                double o3[3];
                unit_plus(x3, d3, o3);
                nw = o3[i];
            } else {
                nw = old + db;
            }
            if (d.constrained && i >= 3) {   // ParameterBlock::Plus projects onto the box constraints
                const int bi = i < 3 + M3 ? (i - 3) % 3 : 3;
                nw = fmin(fmax(nw, d.blo[bi]), d.bhi[bi]);
            }
        }
    }
    if (in) d.cand_sh[i] = nw;
    if (i < 64) sdf[i] = (nw - old) * (nw - old);
    const bool any_bad = __ballot(bad != 0.0) != 0ull;      // (the entries all sit in wave 0, and so does lane 0)
    __syncthreads();
    if (i != 0) return;
    double dn = 0.0;
    if (moved)
        for (int k = 0; k < d.nsh; ++k) dn += sdf[k];
    d.part_pose[d.n_pose_blocks * NPP] = dn;
    d.part_pose[d.n_pose_blocks * NPP + 1] = any_bad ? 1.0 : 0.0;
}

// border part of the dogleg vectors: v_b = s^2 g / D^2 and its share of |gradient_|^2, |gn|^2, gradient_.gn (one lane;
// the work-group behind the pose blocks of k_dogleg_vec, a launch of its own -- k_ph_dogleg_border, 8.1 us -- up to r04)
static __device__ __forceinline__ void ph_dogleg_border_lane(const Dev &d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    double gsq = 0.0, nsq = 0.0, dot = 0.0;
    for (int c = 0; c < d.nb; ++c) {
        const double s = d.bsys[BS_S + c], s2 = s * s, g = d.bsys[BS_G + c], gn = d.bsys[BS_DB + c];
        const double D2 = fmin(fmax(d.bsys[BS_H + c] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag);
        d.bsys[BS_VB + c] = s2 * g / D2;
        gsq += s2 * g * g / D2;
        nsq += D2 * gn * gn / s2;
        dot += g * gn;
    }
    double *o = d.part_dl + (size_t)(d.n_lm_blocks + d.n_pose_blocks) * NDL;
    o[0] = gsq; o[1] = nsq; o[2] = dot; o[3] = 0.0; o[4] = 0.0; o[5] = 0.0;
}


}  // namespace ssba

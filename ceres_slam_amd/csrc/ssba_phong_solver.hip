// Config-3 solver kernels: stereo + Phong intensity + normal residual blocks per observation
// (/root/reference tests/dataset_ba_phong.cpp:53-69,102-195), landmark block = [position | normal]
// (6-D, the normal through UnitVectorPerturbation, perturbations.hpp:87-113); the shared light /
// material / texture blocks are constant in this build, so the problem stays bipartite and the
// reduced camera system keeps its block-tridiagonal shape: assembly, BCR, pose update, control
// kernels are the stereo ones (ssba_kernels.hip, ssba_bcr.hip).  Only the per-observation kernels
// differ:
//   k_ph_linearize_landmarks   H_ll (21 unique of 6x6), g_l (6), cost, Jacobi scale
//   k_ph_linearize_poses       H_pp (21), g_p (6) over 7 residual rows
//   k_ph_schur_windows         W = J_p^T J_l (6x6), Y = W C^-1, pair products with K = 6
//   k_ph_backsub_eval          delta = -C^-1 (g_l + sum W^T delta_p), Plus, model change, candidate cost
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "ssba_device.h"
#include "ssba_launch.h"
#include "ssba_linesearch.h"
#include "ssba_phong_device.h"
#include "ssba_types.h"
#include "ssba_check.h"

namespace ssba {

__constant__ uint8_t c_ph_pair_a[NPAIR];
__constant__ uint8_t c_ph_pair_b[NPAIR];

int upload_phong_tables(hipStream_t s) {
    uint8_t a[NPAIR], b[NPAIR];
    int n = 0;
    for (int i = 0; i < TW; ++i)
        for (int j = i; j < TW; ++j) { a[n] = (uint8_t)i; b[n] = (uint8_t)j; ++n; }
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ph_pair_a), a, NPAIR, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ph_pair_b), b, NPAIR, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}

// The 7 residual rows of one observation and their local Jacobians:
//   rows 0-2 stereo    : pose Jp3 (3x6)   landmark [Jl3 (3x3) | 0]
//   row  3   intensity : pose jp (6)      landmark [jpos (3) | jn (3)]
//   rows 4-6 normal    : pose Jnp (3x6)   landmark [0 | Jnn (3x3)]
struct ObsPh {
    double r[7];
    double Jp[42];   // 7 x 6
    double Jl[42];   // 7 x 6
    double jb[NBQ];  // intensity row w.r.t. the shared blocks: [phong 3 | kd | light 3]
    double half_sq;  // 1/2 |r|^2
};

// shared blocks of one observation out of the packed state [light 3 | phong 3M | texture M]
struct Shared { double light[3], ph3[3], kd; };
static __device__ __forceinline__ void load_shared(const Dev &d, const double *__restrict__ sh, uint32_t mat, Shared &x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { x.light[c] = sh[c]; x.ph3[c] = sh[3 + 3 * mat + c]; }
    x.kd = sh[3 + 3 * d.M + mat];
}
// border column of entry q of jb for material `mat`, -1 when that block is constant
static __device__ __forceinline__ int bcol(const Dev &d, uint32_t mat, int q) {
    if (q < 3) return d.b_phong < 0 ? -1 : d.b_phong + 3 * (int)mat + q;
    if (q == 3) return d.b_tex < 0 ? -1 : d.b_tex + (int)mat;
    return d.b_light < 0 ? -1 : d.b_light + (q - 4);
}

struct LmIn { double p[3], n[3]; uint32_t mat; };
static __device__ __forceinline__ void load_lm(const Dev &d, int l, LmIn &x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { x.p[c] = d.pts[(size_t)c * d.Lpad + l]; x.n[c] = d.nrm[(size_t)c * d.Lpad + l]; }
    x.mat = d.lm_mat[l];
}

// Observation slots of one landmark and where their data sit (lighting kernels).  Windowed layout: the TW slots of
// the landmark's window in the transposed ELL arrays.  General layout (tracks > TW, loop closures): the same arrays
// hold the observations landmark-major, slot s of landmark l at dn_lm_start[l] + s.
template <bool DN> struct PhSlots;
template <> struct PhSlots<false> {
    uint32_t mask, win;
    size_t obase;
    __device__ __forceinline__ PhSlots(const Dev &d, int l, uint32_t m)
        : mask(m), win(d.lm_win[l]), obase((size_t)(l >> 6) * (TW * LMG) + (l & 63)) {}
    __device__ __forceinline__ int count() const { return TW; }
    __device__ __forceinline__ bool has(int s) const { return (mask >> s) & 1u; }
    __device__ __forceinline__ uint32_t pose(const Dev &d, int s) const { return d.win_pose[win * TW + s]; }
    __device__ __forceinline__ size_t at(int s) const { return obase + (size_t)s * LMG; }
};
template <> struct PhSlots<true> {
    uint32_t b, n;
    __device__ __forceinline__ PhSlots(const Dev &d, int l, uint32_t) : b(d.dn_lm_start[l]), n(d.dn_lm_start[l + 1] - d.dn_lm_start[l]) {}
    __device__ __forceinline__ int count() const { return (int)n; }
    __device__ __forceinline__ bool has(int) const { return true; }
    __device__ __forceinline__ uint32_t pose(const Dev &d, int s) const { return d.dn_obs_pose[b + s]; }
    __device__ __forceinline__ size_t at(int s) const { return (size_t)b + s; }
};
// observation behind entry i of the pose-major list: landmark and data index
template <bool DN> static __device__ __forceinline__ void pose_list_entry(const Dev &d, uint32_t i, int &l, size_t &oi) {
    if (DN) {
        oi = d.dn_pose_obs[i];
        l = (int)d.dn_obs_lm[oi];
    } else {
        const uint32_t ref = d.pose_obs_ref[i];
        l = (int)(ref >> 4);
        const int s = (int)(ref & 15u);
        oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)s * LMG + (l & 63);
    }
}

static __device__ __forceinline__ void obs_ph_linearize(const Dev &d, const double *__restrict__ sh, const double *__restrict__ T, const double p[3],
                                                        const double n[3], uint32_t mat, double u, double v, double dd,
                                                        double inten, const double nobs[3], bool want_pose, ObsPh &o) {
    ObsLin s;
    obs_linearize(d, T, p[0], p[1], p[2], u, v, dd, s);
    double Jl3[9];
    jac_point(s, T, Jl3);
#pragma unroll
    for (int i = 0; i < 42; ++i) o.Jl[i] = 0.0;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        o.r[m] = s.r[m];
#pragma unroll
        for (int c = 0; c < 3; ++c) o.Jl[6 * m + c] = Jl3[3 * m + c];
    }
    if (want_pose) {
        double Jp3[18];
        jac_pose(s, Jp3);
#pragma unroll
        for (int i = 0; i < 18; ++i) o.Jp[i] = Jp3[i];
    }
    Shared sx;
    load_shared(d, sh, mat, sx);
    double ri, J19[19], rn[3], Jnp[18], Jnn[9];
    intensity_residual(d.light_type, T, p, n, sx.ph3, sx.kd, sx.light, inten, d.int_stiff, &ri, J19);
#pragma unroll
    for (int q = 0; q < NBQ; ++q) o.jb[q] = J19[12 + q];
    normal_residual(T, n, nobs, d.Sn, rn, Jnp, Jnn);
    o.r[3] = ri;
#pragma unroll
    for (int c = 0; c < 6; ++c) o.Jl[18 + c] = J19[6 + c];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        o.r[4 + m] = rn[m];
#pragma unroll
        for (int c = 0; c < 3; ++c) o.Jl[6 * (4 + m) + 3 + c] = Jnn[3 * m + c];
    }
    if (want_pose) {
#pragma unroll
        for (int c = 0; c < 6; ++c) o.Jp[18 + c] = J19[c];
#pragma unroll
        for (int i = 0; i < 18; ++i) o.Jp[24 + i] = Jnp[i];
    }
    if (d.pos_const) {      // constant position blocks (dataset_ba_phong.cpp:213-220): their Jacobian columns leave the problem
#pragma unroll
        for (int m = 0; m < 7; ++m) o.Jl[6 * m] = o.Jl[6 * m + 1] = o.Jl[6 * m + 2] = 0.0;
    }
    o.half_sq = s.half_rho + 0.5 * (ri * ri + rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
}

// The same seven residual rows handed out part by part -- stereo rows 0..2, intensity row 3, normal rows 4..6 -- so that a
// kernel that only folds rows into sums never holds the whole 7 x 6 Jacobians (obs_ph_linearize's ObsPh is 100 doubles;
// the landmark kernels built on it sat at 330-370 registers, one wave per SIMD).  fn(part, m, r, jp, jl, jb):
//   part   std::integral_constant<int, 0 / 1 / 2>; the landmark columns of the part's rows that can be non-zero are
//          [ph_jl_lo(part), ph_jl_hi(part))  (positions for the stereo rows, all six for the intensity row, the normal for
//          the normal rows)
//   m, r   row index and residual;  jp: row m of J_p (6 entries, only with want_pose);  jl: row m of J_l (6 entries,
//          structural zeros filled in, position columns zeroed for constant position blocks);  jb: the seven border
//          entries of the intensity row, nullptr for the other rows.
// Returns 1/2 |r|^2 (with the stereo loss applied).
__host__ __device__ constexpr int ph_jl_lo(int part) { return part == 2 ? 3 : 0; }
__host__ __device__ constexpr int ph_jl_hi(int part) { return part == 0 ? 3 : 6; }
template <class F>
static __device__ __forceinline__ double ph_rows(const Dev &d, const double *__restrict__ sh, const double *__restrict__ T, const LmIn &x,
                                                 double u, double v, double dd, double inten, const double nobs[3], bool want_pose, F &&fn) {
    double half_sq;
    {
        ObsLin s;
        obs_linearize(d, T, x.p[0], x.p[1], x.p[2], u, v, dd, s);
        double Jl3[9], Jp3[18];
        jac_point(s, T, Jl3);
        if (want_pose) jac_pose(s, Jp3);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const double jl[6] = {d.pos_const ? 0.0 : Jl3[3 * m], d.pos_const ? 0.0 : Jl3[3 * m + 1], d.pos_const ? 0.0 : Jl3[3 * m + 2], 0.0, 0.0, 0.0};
            fn(std::integral_constant<int, 0>(), m, s.r[m], Jp3 + 6 * m, jl, (const double *)nullptr);
        }
        half_sq = s.half_rho;
    }
    {
        Shared sx;
        load_shared(d, sh, x.mat, sx);
        double ri, J19[19];
        intensity_residual(d.light_type, T, x.p, x.n, sx.ph3, sx.kd, sx.light, inten, d.int_stiff, &ri, J19);
        const double jl[6] = {d.pos_const ? 0.0 : J19[6], d.pos_const ? 0.0 : J19[7], d.pos_const ? 0.0 : J19[8], J19[9], J19[10], J19[11]};
        fn(std::integral_constant<int, 1>(), 3, ri, J19, jl, J19 + 12);
        half_sq += 0.5 * ri * ri;
    }
    {
        double rn[3], Jnp[18], Jnn[9];
        normal_residual(T, x.n, nobs, d.Sn, rn, Jnp, Jnn);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const double jl[6] = {0.0, 0.0, 0.0, Jnn[3 * m], Jnn[3 * m + 1], Jnn[3 * m + 2]};
            fn(std::integral_constant<int, 2>(), 4 + m, rn[m], Jnp + 6 * m, jl, (const double *)nullptr);
        }
        half_sq += 0.5 * (rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
    }
    return half_sq;
}

// residuals only (candidate evaluation)
static __device__ __forceinline__ double obs_ph_cost(const Dev &d, const double *__restrict__ sh, const double *__restrict__ T,
                                                     const double p[3], const double n[3], uint32_t mat, double u, double v,
                                                     double dd, double inten, const double nobs[3]) {
    Shared sx;
    load_shared(d, sh, mat, sx);
    double ri, rn[3];
    intensity_residual(d.light_type, T, p, n, sx.ph3, sx.kd, sx.light, inten, d.int_stiff, &ri, nullptr);
    normal_residual(T, n, nobs, d.Sn, rn, nullptr, nullptr);
    return obs_cost(d, T, p[0], p[1], p[2], u, v, dd) + 0.5 * (ri * ri + rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
}

__device__ __forceinline__ int tri6(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }   // r <= c
// Structure of the 7 x 6 landmark Jacobian [position | normal]: the stereo rows 0..2 touch the position only, the normal
// rows 4..6 the normal only, the intensity row 3 both.  In fully unrolled loops the test folds away and with it 18 of
// the 42 entries (their registers and 43 % of the FMAs of W = J_p^T J_l).
__device__ __forceinline__ constexpr bool jl_nz(int m, int c) { return m == 3 || (m < 3 ? c < 3 : c >= 3); }

// inverse of the damped 6x6 landmark block (packed upper h[21], diagonal damping dmp[6]) through its
// Cholesky factor; result packed upper Ci[21].  false on breakdown.
static __device__ __forceinline__ bool inv6_spd(const double h[21], const double dmp[6], double Ci[21], double *Mo = nullptr) {
    double L[6][6], M[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = h[tri6(j, j)] + dmp[j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k];
        if (!(dj > 0.0) || !isfinite(dj)) return false;
        const double r = fast_rsqrt(dj);
        L[j][j] = dj * r;
        M[j][j] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = h[tri6(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * r;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s -= L[i][k] * M[k][j];
            M[i][j] = s * M[i][i];
        }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = b; k < 6; ++k) s += M[k][a] * M[k][b];
            Ci[tri6(a, b)] = s;
        }
    if (Mo) {       // M = L^-1, packed lower, row-major: C^-1 = M^T M
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Mo[i * (i + 1) / 2 + j] = M[i][j];
    }
    return true;
}

static __device__ __forceinline__ void ph_damping(const Dev &d, const State &st, int l, const double h[21], double dmp[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double s = d.sl[(size_t)c * d.Lpad + l], s2 = s * s;
        dmp[c] = fmin(fmax(h[tri6(c, c)] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    }
}


// ------------------------------------------------------------------ kernels ---
// One lane per landmark: C^-1 = (H_ll + D^2)^-1 for the current radius; read by the Schur producers
// and the back-substitution (a radius change alone re-runs this, not the linearisation).
// Also clears the block-tridiagonal reduced system (D and L of every super-block; of this rank's chain in a partitioned
// solve) that k_assemble_reduced fills after the Schur launch -- no memset launch.
// blk / nblk: this work-group's index among the n_lm_blocks that share the pass (k_ph_invert, or the tail of k_ph_linpose_invert)
static __device__ __forceinline__ void ph_invert_body(const Dev &d, int blk, int nblk) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int l = blk * 256 + threadIdx.x;
    {
        const size_t b0 = d.part ? (size_t)d.chain0 * BD * BD : 0;
        const size_t n2 = (d.part ? (size_t)(d.chain1 - d.chain0 + 1) : (size_t)d.Nsb) * (BD * BD / 2);       // double2 per range
        double2 *zD = reinterpret_cast<double2 *>(d.xv + d.off_D + b0), *zL = reinterpret_cast<double2 *>(d.xv + d.off_L + b0);
        for (size_t i = (size_t)l; i < n2; i += (size_t)nblk * 256) {
            zD[i] = make_double2(0.0, 0.0);
            zL[i] = make_double2(0.0, 0.0);
        }
    }
    double Ci[21], Mf[21];
#pragma unroll
    for (int c = 0; c < 21; ++c) { Ci[c] = 0.0; Mf[c] = 0.0; }
    if (d.lm_mask[l]) {
        double h[21], dmp[6];
#pragma unroll
        for (int c = 0; c < 21; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
        ph_damping(d, st, l, h, dmp);
        if (!inv6_spd(h, dmp, Ci, Mf)) {
            d.st->step_failed = 1;
#pragma unroll
            for (int c = 0; c < 21; ++c) { Ci[c] = 0.0; Mf[c] = 0.0; }
        }
    }
#pragma unroll
    for (int c = 0; c < 21; ++c) { d.cinv[(size_t)c * d.Lpad + l] = Ci[c]; d.cfac[(size_t)c * d.Lpad + l] = Mf[c]; }
    if (d.lmMV) {       // M V_j (6 x 7): the border columns of this landmark's rows in the Schur product (k_ph_schur_windows<true>)
        double V[42];
#pragma unroll
        for (int i = 0; i < 42; ++i) V[i] = d.lmV[(size_t)i * d.Lpad + l];
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b <= c; ++b) v += Mf[c * (c + 1) / 2 + b] * V[b * NBQ + q];
                d.lmMV[(size_t)(c * NBQ + q) * d.Lpad + l] = v;
            }
    }
}
__global__ __launch_bounds__(256) void k_ph_invert(Dev d) { ph_invert_body(d, (int)blockIdx.x, (int)gridDim.x); }

// blk: the group of 256 landmarks (the work-group index of k_ph_linearize_landmarks; an offset one inside k_ph_linearize_all)
template <bool DN> static __device__ __forceinline__ void ph_lin_landmarks_body(const Dev &d, int blk) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    __shared__ double sm[4];
    const int l = blk * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double cost = 0.0, xn = 0.0, gm = 0.0;
    if (mask) {
        const PhSlots<DN> sl(d, l, mask);
        LmIn x;
        load_lm(d, l, x);
        double h[21], g[6];
#pragma unroll
        for (int i = 0; i < 21; ++i) h[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = 0.0;
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            cost += ph_rows(d, d.sh, d.poses + (size_t)k * 12, x, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, false,
                            [&](auto part, int, double r, const double *, const double *jl, const double *) {
                                constexpr int P = decltype(part)::value;
#pragma unroll
                                for (int a = ph_jl_lo(P); a < ph_jl_hi(P); ++a) {
                                    g[a] += jl[a] * r;
#pragma unroll
                                    for (int b = a; b < ph_jl_hi(P); ++b) h[tri6(a, b)] += jl[a] * jl[b];
                                }
                            });
        }
#pragma unroll
        for (int c = 0; c < 21; ++c) d.hll[(size_t)c * d.Lpad + l] = h[c];
#pragma unroll
        for (int c = 0; c < 6; ++c) d.gl[(size_t)c * d.Lpad + l] = g[c];
        if (st.iteration == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                d.sl[(size_t)c * d.Lpad + l] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h[tri6(c, c)])) : 1.0;
        }
        xn = x.n[0] * x.n[0] + x.n[1] * x.n[1] + x.n[2] * x.n[2];
        if (!d.pos_const) xn += x.p[0] * x.p[0] + x.p[1] * x.p[1] + x.p[2] * x.p[2];
        // projected gradient |x - Plus(x, -g)|_inf: Euclidean for the position, unit-vector Plus for the normal
        const double ng[3] = {-g[3], -g[4], -g[5]};
        double nn[3];
        unit_plus(x.n, ng, nn);
        gm = fmax(fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))),
                  fmax(fabs(nn[0] - x.n[0]), fmax(fabs(nn[1] - x.n[1]), fabs(nn[2] - x.n[2]))));
    }
    const double c0 = block_sum(cost, sm);
    const double c1 = block_sum(xn, sm);
    const double c2 = block_max(gm, sm);
    if (threadIdx.x == 0) {
        d.part_lin[blk * 4 + 0] = c0;
        d.part_lin[blk * 4 + 1] = c1;
        d.part_lin[blk * 4 + 2] = c2;
    }
}
template <bool DN> __global__ __launch_bounds__(256, 2) void k_ph_linearize_landmarks(Dev d) { ph_lin_landmarks_body<DN>(d, (int)blockIdx.x); }

template <bool DN, int NT, int CH> static __device__ __forceinline__ void ph_lin_pose_body(const Dev &d, int k) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    if (d.pose_free[k] < 0) return;
    __shared__ double sm[NT / 64][27];
    const double *T = d.poses + (size_t)k * 12;
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    const uint32_t b = DN ? d.dn_pose_start[k] : d.pose_obs_start[k], e = DN ? d.dn_pose_start[k + 1] : d.pose_obs_start[k + 1];
    // CH observations per round: their list entries, then their 13 operands each are in flight together (a rolled loop pays
    // two dependent memory round trips per observation)
    for (uint32_t i0 = b + threadIdx.x; i0 < e; i0 += NT * CH) {
        int l[CH];
        size_t oi[CH];
        LmIn x[CH];
        double in[CH][7];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            l[q] = -1;
            if (i0 + NT * q < e) pose_list_entry<DN>(d, i0 + NT * q, l[q], oi[q]);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (l[q] < 0) continue;
            load_lm(d, l[q], x[q]);
            in[q][0] = d.onx[oi[q]]; in[q][1] = d.ony[oi[q]]; in[q][2] = d.onz[oi[q]];
            in[q][3] = d.ou[oi[q]]; in[q][4] = d.ov[oi[q]]; in[q][5] = d.od[oi[q]]; in[q][6] = d.oi[oi[q]];
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (l[q] < 0) continue;
            const double nobs[3] = {in[q][0], in[q][1], in[q][2]};
            ph_rows(d, d.sh, T, x[q], in[q][3], in[q][4], in[q][5], in[q][6], nobs, true,
                    [&](auto, int, double r, const double *jp, const double *, const double *) {
                        int n = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
                            acc[21 + a] += jp[a] * r;
#pragma unroll
                            for (int c = a; c < 6; ++c) acc[n++] += jp[a] * jp[c];
                        }
                    });
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        const double v = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += sm[w][threadIdx.x];
        if (threadIdx.x < 21) d.hpp[(size_t)k * 21 + threadIdx.x] = v;
        else d.gp[(size_t)k * 6 + (threadIdx.x - 21)] = v;
    }
}
template <bool DN, int NT, int CH> __global__ __launch_bounds__(NT) void k_ph_linearize_poses(Dev d) {
    const int k = xcd_contiguous_item((int)blockIdx.x, d.P);      // neighbouring poses share their landmarks: one XCD's L2 (ssba_device.h)
    if (k >= 0) ph_lin_pose_body<DN, NT, CH>(d, k);
}
// Pose linearisation and the inversion of the damped landmark blocks in ONE launch (windowed layout, constant shared
// blocks): both only need the landmark linearisation, so the n_lm_blocks inversion work-groups run beside the P pose
// work-groups instead of as a launch of their own between them and the Schur kernel (-20 us on the chain at C3).
__global__ __launch_bounds__(256, 2) void k_ph_linpose_invert(Dev d) {
    const int np = xcd_contiguous_grid(d.P);
    if ((int)blockIdx.x < np) {
        const int k = xcd_contiguous_item((int)blockIdx.x, d.P);
        if (k >= 0) ph_lin_pose_body<false, 256, 2>(d, k);
    } else ph_invert_body(d, (int)blockIdx.x - np, d.n_lm_blocks);
}

// Output-stationary Schur complement for 6-D landmark blocks on the fp64 matrix cores: the structure of
// k_schur_windows (ssba_kernels.hip) with six factor rows per landmark.  C^-1 = M^T M comes from k_ph_invert, so
// W_a C^-1 W_b^T = Z_a Z_b^T with Z = W M^T (6 x 6 per (landmark, slot)); Zm[k][col], k = 6*landmark + c, is staged
// k-major in LDS (96 rows per batch of 16 landmarks, 80 columns: 72 + the gradient column u = M g_l + padding) and
// S[72 x 73] += Zm^T Zm runs on v_mfma_f64_16x16x4_f64, the 15 upper tiles 4 / 4 / 4 / 3 per wave.
// W = J_p^T J_l of one observation (6 x 6, [a][c]; the 7 residual rows: stereo 3, intensity 1, normal 3), built part by
// part so that only one part's Jacobians are alive at a time (the full ObsPh -- 7 x 6 twice -- plus W and Z is more
// than a wave's registers at two waves per SIMD).  Same products in the same row order as the generic loop over
// obs_ph_linearize's output.
static __device__ __forceinline__ void ph_W(const Dev &d, const double *__restrict__ sh, const double *__restrict__ T, const LmIn &x,
                                            double u, double v, double dd, double inten, const double nobs[3], double W[36]) {
    {
        ObsLin s;
        obs_linearize(d, T, x.p[0], x.p[1], x.p[2], u, v, dd, s);
        double Jl3[9], Jp3[18];
        jac_point(s, T, Jl3);
        jac_pose(s, Jp3);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                W[6 * a + c] = d.pos_const ? 0.0 : Jp3[a] * Jl3[c] + Jp3[6 + a] * Jl3[3 + c] + Jp3[12 + a] * Jl3[6 + c];
    }
    {
        Shared sx;
        load_shared(d, sh, x.mat, sx);
        double ri, J19[19];
        intensity_residual(d.light_type, T, x.p, x.n, sx.ph3, sx.kd, sx.light, inten, d.int_stiff, &ri, J19);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = 0; c < 3; ++c) W[6 * a + c] += d.pos_const ? 0.0 : J19[a] * J19[6 + c];
#pragma unroll
            for (int c = 3; c < 6; ++c) W[6 * a + c] = J19[a] * J19[6 + c];
        }
    }
    {
        double rn[3], Jnp[18], Jnn[9];
        normal_residual(T, x.n, nobs, d.Sn, rn, Jnp, Jnn);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int m = 0; m < 3; ++m) W[6 * a + 3 + c] += Jnp[6 * m + a] * Jnn[3 * m + c];
    }
}

constexpr int PH_THREADS = 256;
// BORDER (free shared blocks, windowed layout): the pose rows of the border, S_pb = H_pb - sum_j W_j C_j^-1 V_j, ride in
// the same product.  With C^-1 = M^T M the sum is  sum_j Z_j (M V_j) , i.e. Zm^T [MV]: every k-row (landmark, c) gets 32
// more columns holding row c of M V_j at the border columns of the landmark's material (zero elsewhere), and the waves
// accumulate ten more 16 x 16 tiles (72 x 32) per item, stored to slabB and gathered by k_ph_spb_assemble; H_pb comes
// from k_ph_hpb on linearisation.  This replaces k_ph_border_poses, which re-linearised every observation a fifth time per
// iteration from one wave per pose (400 us at C3).  The wider k-major matrix costs LDS: batches of 12 landmarks
// (72 x 112 doubles) keep two workgroups on a CU.
template <bool BORDER> struct PhSchurCfg {
    static constexpr int BATCH = BORDER ? 13 : 16;          // 156 / 192 producer lanes
    static constexpr int KB = (6 * BATCH + 3) / 4 * 4;      // 80 / 96 factor rows (78 used) = 20 / 24 MFMA steps
    // row stride: 72 + gradient column + padding; BORDER: the border columns start at 74 -- over the padding, whose
    // content only reaches output entries that are never stored
    static constexpr int BC0 = 74;
    static constexpr int RS = BORDER ? BC0 + NBP : 80;
    static constexpr int LMW = BORDER ? 72 : 28;            // per-landmark staging: M (21) | g_l (6) | pad | M V (42) | material
    static constexpr int LDS_DOUBLES = KB * RS + BATCH * LMW;
};
typedef double ph_d4 __attribute__((ext_vector_type(4)));

// check_parts > 0: one more work-group at the head of the grid does k_check's work (see k_schur_windows, ssba_kernels.hip)
template <bool BORDER> __global__ __launch_bounds__(PH_THREADS, 2) void k_ph_schur_windows(Dev d, int check_parts) {
    typedef PhSchurCfg<BORDER> C;
    constexpr int PH_BATCH = C::BATCH, PH_KB = C::KB, PH_RS = C::RS, LMW = C::LMW, BC0 = C::BC0;
    if (check_parts > 0 && blockIdx.x == 0) { check_body(d, check_parts, true); return; }      // (the first work-group: it must not be the launch's tail)
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    extern __shared__ __align__(16) double ph_lds[];
    double *sZ = ph_lds;                         // [k][col]
    double *sLM = ph_lds + PH_KB * PH_RS;        // [landmark][M(21) | g_l(6) | pad | M V (42) | material]
    const int item = (int)blockIdx.x - (check_parts > 0 ? 1 : 0);
    const uint32_t win = d.slab_win[item];
    const int lb = (int)d.slab_lm_begin[item], le = (int)d.slab_lm_end[item];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool producer = t < PH_BATCH * TW;
    const int li = t / TW, s = t - li * TW;
    const int ti0 = wv == 0 ? 0 : wv == 1 ? 0 : wv == 2 ? 1 : 3, tj0 = wv == 0 ? 0 : wv == 1 ? 4 : wv == 2 ? 4 : 3;
    const int ti1 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 3, tj1 = wv == 0 ? 1 : wv == 1 ? 1 : wv == 2 ? 2 : 4;
    const int ti2 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 4, tj2 = wv == 0 ? 2 : wv == 1 ? 2 : wv == 2 ? 3 : 4;
    const int ti3 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 4, tj3 = wv == 0 ? 3 : wv == 1 ? 3 : wv == 2 ? 4 : 4;
    const bool has3 = wv != 3;
    // border tiles n = wv, wv + 4, wv + 8 (< 10): rows of tile n % 5, columns 80 + 16 (n / 5) ..
    const int bi0 = wv, bi1 = (wv + 4) % 5, bi2 = (wv + 8) % 5;
    const int bj0 = 5, bj1 = 5 + (wv + 4) / 5, bj2 = 6;
    const bool hasb2 = wv + 8 < 10;
    ph_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0, bac0 = acc0, bac1 = acc0, bac2 = acc0;
    // padding columns 73..79 stay zero for the whole item (column 72 is rewritten every batch); BORDER: the padding is
    // column 73, the k-rows beyond 6 x BATCH stay zero
    if (!BORDER) {
        for (int e = t; e < PH_KB * 8; e += PH_THREADS) sZ[(e >> 3) * PH_RS + 72 + (e & 7)] = 0.0;
    } else {
        for (int e = t; e < PH_KB; e += PH_THREADS) sZ[e * PH_RS + 73] = 0.0;
        for (int e = t; e < (PH_KB - 6 * PH_BATCH) * PH_RS; e += PH_THREADS) sZ[6 * PH_BATCH * PH_RS + e] = 0.0;
    }
    bool pose_ok = false;
    uint32_t k = 0xFFFFFFFFu;
    if (producer) {
        k = d.win_pose[win * TW + s];
        pose_ok = (k != 0xFFFFFFFFu) && d.pose_free[k] >= 0;
    }
    for (int l0 = lb; l0 < le; l0 += PH_BATCH) {
        // phase 0: stage M (21) and g_l (6) of the batch (+ M V and the material)
        for (int i = t; i < PH_BATCH * 27; i += PH_THREADS) {
            const int c = i / PH_BATCH, j = i - c * PH_BATCH, l = l0 + j;
            double v = 0.0;
            if (l < le) v = c < 21 ? d.cfac[(size_t)c * d.Lpad + l] : d.gl[(size_t)(c - 21) * d.Lpad + l];
            sLM[j * LMW + c] = v;
        }
        if (BORDER) {
            for (int i = t; i < PH_BATCH * 43; i += PH_THREADS) {
                const int c = i / PH_BATCH, j = i - c * PH_BATCH, l = l0 + j;
                double v = 0.0;
                if (l < le) v = c < 42 ? d.lmMV[(size_t)c * d.Lpad + l] : (double)d.lm_mat[l];
                sLM[j * LMW + 28 + c] = v;
            }
        }
        __syncthreads();
        // phase 1: W = J_p^T J_l and Z = W M^T per (landmark, slot); u = M g_l per landmark
        if (producer) {
            const int l = l0 + li;
            const double *Mf = sLM + li * LMW;
            if (s == 0) {
                const double *g = Mf + 21;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q <= c; ++q) v += Mf[c * (c + 1) / 2 + q] * g[q];
                    sZ[(li * 6 + c) * PH_RS + 72] = v;
                }
            }
            if (BORDER) {       // lane (landmark, s): half s / 6 of the border columns of k-row (landmark, s % 6)
                const int c = s % 6, h = s / 6;
                double *row = sZ + (li * 6 + c) * PH_RS + BC0 + 16 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) reinterpret_cast<double2 *>(row)[q] = make_double2(0.0, 0.0);
                const uint32_t mat = (uint32_t)Mf[28 + 42];
#pragma unroll
                for (int q = 0; q < NBQ; ++q) {
                    const int col = bcol(d, mat, q);
                    if (col >= 0 && (col >> 4) == h) row[col & 15] = Mf[28 + c * NBQ + q];
                }
            }
            double z[36];      // [c][a]
            bool live = false;
            if (l < le && pose_ok && ((d.lm_mask[l] >> s) & 1u)) {
                live = true;
                LmIn x;
                load_lm(d, l, x);
                const size_t oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)s * LMG + (l & 63);
                const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
                double W[36];
                ph_W(d, d.sh, d.poses + (size_t)k * 12, x, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, W);
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q <= c; ++q) v += W[6 * a + q] * Mf[c * (c + 1) / 2 + q];
                        z[6 * c + a] = v;
                    }
            }
            if (!live) {
#pragma unroll
                for (int i = 0; i < 36; ++i) z[i] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double2 *dz = reinterpret_cast<double2 *>(sZ + (li * 6 + c) * PH_RS + s * 6);
#pragma unroll
                for (int q = 0; q < 3; ++q) dz[q] = make_double2(z[6 * c + 2 * q], z[6 * c + 2 * q + 1]);
            }
        }
        __syncthreads();
        // phase 2: S += Zm^T Zm  (BORDER: + Zm^T [M V])
        {
            // Branch-free and software-pipelined like k_schur_windows (ssba_kernels.hip): operands of k-step ks + 1 requested before the
            // products of ks are issued; a wave without a fourth main tile / a third border tile repeats its last one (not stored).
            const int kq = lane >> 4, i = lane & 15;
            const double *z0 = sZ + kq * PH_RS + i;
            const bool a1x = wv == 0 || wv == 3;        // A operand of the second main tile: row ti0 or ti3 (see k_schur_windows)
            const int bi2e = hasb2 ? bi2 : bi1, bc2e = BC0 + 16 * ((hasb2 ? bj2 : bj1) - 5);
            double ax = z0[16 * ti0], ay = z0[16 * ti3], b0 = z0[16 * tj0], b1 = z0[16 * tj1], b2 = z0[16 * tj2], b3 = z0[16 * tj3];
            double ba0 = 0.0, ba1 = 0.0, ba2 = 0.0, bb0 = 0.0, bb1 = 0.0, bb2 = 0.0;
            if (BORDER) {
                ba0 = z0[16 * bi0]; ba1 = z0[16 * bi1]; ba2 = z0[16 * bi2e];
                bb0 = z0[BC0 + 16 * (bj0 - 5)]; bb1 = z0[BC0 + 16 * (bj1 - 5)]; bb2 = z0[bc2e];
            }
#pragma unroll
            for (int ks = 0; ks < PH_KB / 4; ++ks) {
                double nax = 0.0, nay = 0.0, nb0 = 0.0, nb1 = 0.0, nb2 = 0.0, nb3 = 0.0;
                double nba0 = 0.0, nba1 = 0.0, nba2 = 0.0, nbb0 = 0.0, nbb1 = 0.0, nbb2 = 0.0;
                if (ks + 1 < PH_KB / 4) {
                    const double *zr = z0 + 4 * (ks + 1) * PH_RS;
                    nax = zr[16 * ti0]; nay = zr[16 * ti3]; nb0 = zr[16 * tj0]; nb1 = zr[16 * tj1]; nb2 = zr[16 * tj2]; nb3 = zr[16 * tj3];
                    if (BORDER) {
                        nba0 = zr[16 * bi0]; nba1 = zr[16 * bi1]; nba2 = zr[16 * bi2e];
                        nbb0 = zr[BC0 + 16 * (bj0 - 5)]; nbb1 = zr[BC0 + 16 * (bj1 - 5)]; nbb2 = zr[bc2e];
                    }
                }
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1x ? ax : ay, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b3, acc3, 0, 0, 0);
                if (BORDER) {
                    bac0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ba0, bb0, bac0, 0, 0, 0);
                    bac1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ba1, bb1, bac1, 0, 0, 0);
                    bac2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ba2, bb2, bac2, 0, 0, 0);
                }
                ax = nax; ay = nay; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
                ba0 = nba0; ba1 = nba1; ba2 = nba2; bb0 = nbb0; bb1 = nbb1; bb2 = nbb2;
            }
        }
        __syncthreads();
    }
    auto store_tile = [&](const ph_d4 &acc, int ti, int tj) {
        const int col = 16 * tj + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + (lane >> 4) + 4 * r;
            if (row >= 72) continue;
            const int a = row / 6, ra = row - 6 * a;
            if (col == 72) { d.slab[(size_t)item * SLAB_DOUBLES + NPAIR * 36 + row] = acc[r]; continue; }
            if (col > 72 || col < row) continue;
            const int b = col / 6, cb = col - 6 * b;
            double *out = d.slab + (size_t)item * SLAB_DOUBLES + (size_t)(a * TW - (a * (a - 1)) / 2 + (b - a)) * 36;
            out[ra * 6 + cb] = acc[r];
            if (a == b) out[cb * 6 + ra] = acc[r];
        }
    };
    store_tile(acc0, ti0, tj0);
    store_tile(acc1, ti1, tj1);
    store_tile(acc2, ti2, tj2);
    if (has3) store_tile(acc3, ti3, tj3);
    if (BORDER) {
        auto store_border = [&](const ph_d4 &acc, int ti, int tj) {
            const int col = 16 * (tj - 5) + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + (lane >> 4) + 4 * r;
                if (row < 72) d.slabB[((size_t)item * 72 + row) * NBP + col] = acc[r];
            }
        };
        store_border(bac0, bi0, bj0);
        store_border(bac1, bi1, bj1);
        if (hasb2) store_border(bac2, bi2, bj2);
    }
}

// On linearisation, windowed layout: H_pb of every free pose = sum over its observations of
// (intensity row of J_p)^T (x) j_b -- the part of S_pb that does not depend on the trust-region radius.  One wave per
// (pose, material): the pose's references are sorted by material, so the wave owns four columns of H_pb outright and
// leaves its share of the three light columns as a partial that k_ph_spb_assemble sums over the materials in order.
// (One workgroup per pose walking the materials one after the other: 82 us -- a pass has ~170 observations.)
__global__ __launch_bounds__(64) void k_ph_hpb(Dev d) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const int k = blockIdx.x / d.M, m = blockIdx.x - k * d.M, f = d.pose_free[k];
    if (f < 0) return;
    const int t = threadIdx.x;
    const double *T = d.poses + (size_t)k * 12;
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; ++i) acc[i] = 0.0;
    const uint32_t b = d.pose_mat_start[(size_t)k * (d.M + 1) + m], e = d.pose_mat_start[(size_t)k * (d.M + 1) + m + 1];
    Shared sx;
    load_shared(d, d.sh, (uint32_t)m, sx);
    for (uint32_t i = b + t; i < e; i += 64) {
        int l;
        size_t oi;
        pose_list_entry<false>(d, i, l, oi);
        LmIn x;
        load_lm(d, l, x);
        double ri, J19[19];
        intensity_residual(d.light_type, T, x.p, x.n, sx.ph3, sx.kd, sx.light, d.oi[oi], d.int_stiff, &ri, J19);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int q = 0; q < NBQ; ++q) acc[q < 4 ? a * 4 + q : 24 + a * 3 + (q - 4)] += J19[a] * J19[12 + q];
    }
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < 42; ++i) {
        const double v = wave_sum(acc[i]);
        const double v0 = __shfl(v, 0, 64);
        if (t == i) mine = v0;
    }
    if (t < 24) {
        const int a = t / 4, col = bcol(d, (uint32_t)m, t - a * 4);
        if (col >= 0) d.Hpb[((size_t)f * 6 + a) * NBP + col] = mine;
    } else if (t < 42) {
        d.HpbL[((size_t)k * d.M + m) * 18 + (t - 24)] = mine;
    }
}

// S_pb = H_pb - sum over the Schur items of their border tiles (k_ph_schur_windows<true>): one thread per entry, the
// items of a pose in list order (the prow lists of k_assemble_reduced)
// The work-groups behind the first n_asm sum the columns of the border partials (k_ph_border_colsum's launch up to r04: both only
// need k_ph_border_schur's and the Schur launch's output).  ride_dst: where the border columns ride through the reduced solve
// (launch_bcr copied Spb there, one more 5 us node per iteration) -- ride_count entries, the padding rows included.
__global__ __launch_bounds__(256) void k_ph_spb_assemble(Dev d, int n_asm, double *ride_dst, size_t ride_count) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    if ((int)blockIdx.x >= n_asm) {
        if (threadIdx.x >= 64) return;
        const int idx = (int)blockIdx.x - n_asm, lane = threadIdx.x;
        double a = 0.0;
        for (int b = lane; b < d.n_lm_blocks; b += 64) a += d.part_b[(size_t)b * d.M * NBV + idx];
        a = wave_sum(a);
        if (lane == 0) d.part_b[(size_t)d.n_lm_blocks * d.M * NBV + idx] = a;
        return;
    }
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)d.nfree * 6 * NBP) {
        if (ride_dst && gid < ride_count) ride_dst[gid] = d.Spb[gid];
        return;
    }
    const int col = (int)(gid % NBP), row = (int)(gid / NBP), f = row / 6, a = row - 6 * f;
    double v = d.Hpb[gid];
    if (d.b_light >= 0 && col >= d.b_light && col < d.b_light + 3) {        // light columns: the materials' partials, in order
        const size_t base = (size_t)d.free_pose[f] * d.M * 18 + a * 3 + (col - d.b_light);
        v = 0.0;
        for (int m = 0; m < d.M; ++m) v += d.HpbL[base + (size_t)m * 18];
    }
    if (col < d.nb)
        for (uint32_t i = d.prow_start[f]; i < d.prow_start[f + 1]; ++i) {
            const uint32_t cw = d.prow_contrib[i];
            v -= d.slabB[((size_t)(cw / TW) * 72 + (cw % TW) * 6 + a) * NBP + col];
        }
    d.Spb[gid] = v;
    if (ride_dst && gid < ride_count) ride_dst[gid] = v;
}

// fuse_best (constant shared blocks): also does k_best's share for the landmarks (x -> best when the k_check of this
// iteration saw the cost improve; before the termination test, like k_backsub_eval_w)
// bounds: the landmark part of the projected line search's alpha-independent terms -- max|delta_l| (its min_step_size test)
// and g_l . delta_l (phi'(0)) -- one partial pair per work-group for k_ph_ls_fast / k_ph_ls_reduce.  Formed by the evaluation
// kernel that has delta_l in registers (a launch of its own, k_ph_ls_dir, up to r04: 5.4 us per iteration).
static __device__ __forceinline__ void ph_ls_dir_terms(const Dev &d, int l, bool moved, const double dl[6], double *sm) {
    double dmax = 0.0, gd = 0.0;
    if (moved) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            dmax = fmax(dmax, fabs(dl[c]));
            gd += d.gl[(size_t)c * d.Lpad + l] * dl[c];
        }
    }
    const double f2 = block_max(dmax, sm), g2 = block_sum(gd, sm);
    if (threadIdx.x == 0) {
        double *o = d.part_ls + (size_t)blockIdx.x * NLS;
        o[4] = f2; o[5] = g2;
    }
}

template <bool DN> __global__ __launch_bounds__(256, 2) void k_ph_backsub_eval(Dev d, int fuse_best) {
    const State &st = *d.st;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    LmIn x;
    load_lm(d, l, x);
    if (fuse_best && st.copy_best == st.check_count) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { d.best_pts[(size_t)c * d.Lpad + l] = x.p[c]; d.best_nrm[(size_t)c * d.Lpad + l] = x.n[c]; }
    }
    if (st.terminated) return;
    double ccost = 0.0, mcc = 0.0, dn = 0.0, nonfinite = 0.0;
    double np_[3] = {x.p[0], x.p[1], x.p[2]}, nn[3] = {x.n[0], x.n[1], x.n[2]};
    double dl[6] = {0, 0, 0, 0, 0, 0};
    if (mask && !st.step_failed) {
        const PhSlots<DN> sl(d, l, mask);
        double gl[6], tt[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) { gl[c] = d.gl[(size_t)c * d.Lpad + l]; tt[c] = gl[c]; }
        double dbq[NBQ];   // border step seen by this landmark's material
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int c = d.nb ? bcol(d, x.mat, q) : -1;
            dbq[q] = c >= 0 ? d.bsys[BS_DB + c] : 0.0;
        }
        // e = J_p delta_p + J_b delta_b of an observation row; tt = g_l + sum J_l^T e, and the model cost change
        // -(J d)^T (r + J d / 2) of this landmark's rows is
        //   -(sum e.r + dl.g_l) - (sum e.e + 2 dl.(tt - g_l) + dl^T H_ll dl) / 2 :  one linearisation pass
        double er = 0.0, ee = 0.0;
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const int f = d.pose_free[k];
            if (f < 0 && !d.nb) continue;
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            double dp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (f >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) dp[c] = d.x0[(size_t)f * 6 + c];
            }
            ph_rows(d, d.sh, d.poses + (size_t)k * 12, x, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, f >= 0,
                    [&](auto part, int, double r, const double *jp, const double *jl, const double *jb) {
                        constexpr int P = decltype(part)::value;
                        double e = 0.0;
                        if (f >= 0) {
#pragma unroll
                            for (int c = 0; c < 6; ++c) e += jp[c] * dp[c];
                        }
                        if (P == 1) {
#pragma unroll
                            for (int q = 0; q < NBQ; ++q) e += jb[q] * dbq[q];
                        }
                        er += e * r;
                        ee += e * e;
#pragma unroll
                        for (int c = ph_jl_lo(P); c < ph_jl_hi(P); ++c) tt[c] += jl[c] * e;
                    });
        }
        double Ci[21];
#pragma unroll
        for (int c = 0; c < 21; ++c) Ci[c] = d.cinv[(size_t)c * d.Lpad + l];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) v += Ci[q <= a ? tri6(q, a) : tri6(a, q)] * tt[q];
            dl[a] = -v;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (!isfinite(dl[a])) nonfinite = 1.0;
        {
            double dg = 0.0, dt = 0.0, dhd = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                dg += dl[a] * gl[a];
                dt += dl[a] * (tt[a] - gl[a]);
                double hd = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) hd += d.hll[(size_t)(q <= a ? tri6(q, a) : tri6(a, q)) * d.Lpad + l] * dl[q];
                dhd += dl[a] * hd;
            }
            mcc = -(er + dg) - 0.5 * (ee + 2.0 * dt + dhd);
        }
        np_[0] = x.p[0] + dl[0]; np_[1] = x.p[1] + dl[1]; np_[2] = x.p[2] + dl[2];
        unit_plus(x.n, dl + 3, nn);
        dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2] + (nn[0] - x.n[0]) * (nn[0] - x.n[0]) +
             (nn[1] - x.n[1]) * (nn[1] - x.n[1]) + (nn[2] - x.n[2]) * (nn[2] - x.n[2]);
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            ccost += obs_ph_cost(d, d.cand_sh, d.cand_poses + (size_t)k * 12, np_, nn, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d.cand_pts[(size_t)c * d.Lpad + l] = np_[c];
        d.cand_nrm[(size_t)c * d.Lpad + l] = nn[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) d.dlm[(size_t)c * d.Lpad + l] = dl[c];
    if (d.constrained) ph_ls_dir_terms(d, l, mask && !st.step_failed, dl, sm);
    const double a = block_sum(ccost, sm), b = block_sum(mcc, sm), c = block_sum(dn, sm), e = block_sum(nonfinite, sm);
    if (threadIdx.x == 0) {
        d.part_eval[blockIdx.x * 4 + 0] = a;
        d.part_eval[blockIdx.x * 4 + 1] = b;
        d.part_eval[blockIdx.x * 4 + 2] = c;
        d.part_eval[blockIdx.x * 4 + 3] = e;
    }
}

// ------------------------------------------------------------------ border ---
// Free shared blocks (light, Phong parameters and texture of each material: the blocks every
// intensity residual of dataset_ba_phong.cpp:108-139 touches) are a dense border of the system:
//   [S_pp S_pb; S_pb^T S_bb] [dp; db] = -[g_p^; g_b^]
// with S_pb = H_pb - sum_j W_j C_j^-1 V_j,  S_bb = H_bb + D_b^2 - sum_j V_j^T C_j^-1 V_j,
// g_b^ = g_b - sum_j V_j^T C_j^-1 g_l,j  and V_j = H_lb of landmark j (6 x 7: only the columns of its
// material and of the light are non-zero).  The kernels below build V_j / H_bb / g_b (on
// linearisation), the Schur-complemented border blocks (every iteration) and the pose rows of S_pb;
// ssba_border.hip solves the arrowhead system on top of the block-cyclic-reduction factors.

__device__ __forceinline__ int tri7(int r, int c) { return r * 7 - (r * (r - 1)) / 2 + (c - r); }   // r <= c

// one lane per landmark, on linearisation: V_j (42), H_bb,j (28 unique), g_b,j (7)
template <bool DN> static __device__ __forceinline__ void ph_border_landmarks_body(const Dev &d, int blk) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const int l = blk * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double V[42], H[28], G[7];
#pragma unroll
    for (int i = 0; i < 42; ++i) V[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 28; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) G[i] = 0.0;
    if (mask) {
        const PhSlots<DN> sl(d, l, mask);
        LmIn x;
        load_lm(d, l, x);
        Shared sx;
        load_shared(d, d.sh, x.mat, sx);
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            double ri, J19[19];
            intensity_residual(d.light_type, d.poses + (size_t)k * 12, x.p, x.n, sx.ph3, sx.kd, sx.light,
                               d.oi[sl.at(s)], d.int_stiff, &ri, J19);
            int c = 0;
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                G[q] += J19[12 + q] * ri;
#pragma unroll
                for (int a = 0; a < 6; ++a) V[a * NBQ + q] += (a < 3 && d.pos_const ? 0.0 : J19[6 + a]) * J19[12 + q];
#pragma unroll
                for (int q2 = q; q2 < NBQ; ++q2) H[c++] += J19[12 + q] * J19[12 + q2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 42; ++i) d.lmV[(size_t)i * d.Lpad + l] = V[i];
#pragma unroll
    for (int i = 0; i < 28; ++i) d.lmH[(size_t)i * d.Lpad + l] = H[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) d.lmG[(size_t)i * d.Lpad + l] = G[i];
}
template <bool DN> __global__ __launch_bounds__(256) void k_ph_border_landmarks(Dev d) { ph_border_landmarks_body<DN>(d, (int)blockIdx.x); }
// Free shared blocks on the window layout: the landmark pass and the pose pass of a linearisation read the same inputs and neither
// reads what the other writes (with constant shared blocks the pose pass shares its launch with the inversion instead:
// k_ph_linpose_invert).  The landmark pass puts 1 564 waves on a machine that holds 2 048 at its register count -- one under-filled
// round; in ONE launch, landmark groups first, the pose work-groups fill what it leaves: 53 + 54 us -> 96 at C3 (-11 us per
// iteration, same bits).  The border pass of the landmarks (k_ph_border_landmarks) in the same launch as well: slower by 19 us -- the
// three bodies in one kernel need 292 bytes of scratch per lane at two waves per SIMD.  SSBA_PH_LIN_LAUNCHES=1 keeps the launches apart.
__global__ __launch_bounds__(256, 2) void k_ph_linearize_all(Dev d, int n_lm) {
    const int b = (int)blockIdx.x;
    if (b < n_lm) { ph_lin_landmarks_body<false>(d, b); return; }
    const int k = xcd_contiguous_item(b - n_lm, d.P);
    if (k >= 0) ph_lin_pose_body<false, 256, 2>(d, k);
}

// one lane per landmark, every iteration (after k_ph_invert): the landmark's contribution to
// S_bb, g_b^, diag(H_bb) and g_b, summed per material over the block of 256 landmarks (fixed order).
constexpr int BSP = 17;   // components per LDS pass (3 passes cover NBV = 49)
constexpr int SSBA_MAX_MATERIALS_DEV = 15;   // = SSBA_MAX_MATERIALS (ssba.h): BSP x M lanes of a 256-lane block sum the per-material partials
__global__ __launch_bounds__(256) void k_ph_border_schur(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double sv[256 * BSP];
    __shared__ uint8_t smat[256];
    const int t = threadIdx.x, l = blockIdx.x * 256 + t;
    const uint32_t mask = d.lm_mask[l];
    double val[NBV + 2];
#pragma unroll
    for (int i = 0; i < NBV + 2; ++i) val[i] = 0.0;
    if (mask) {
        double V[42], Ci[21], CV[42], Cg[6], g[6];
#pragma unroll
        for (int i = 0; i < 42; ++i) V[i] = d.lmV[(size_t)i * d.Lpad + l];
#pragma unroll
        for (int i = 0; i < 21; ++i) Ci[i] = d.cinv[(size_t)i * d.Lpad + l];
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = d.gl[(size_t)i * d.Lpad + l];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double cg = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) cg += Ci[a <= b ? tri6(a, b) : tri6(b, a)] * g[b];
            Cg[a] = cg;
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b) v += Ci[a <= b ? tri6(a, b) : tri6(b, a)] * V[b * NBQ + q];
                CV[a * NBQ + q] = v;
            }
        }
        int c = 0;
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
#pragma unroll
            for (int q2 = q; q2 < NBQ; ++q2) {
                double v = d.lmH[(size_t)c * d.Lpad + l];
                if (q2 == q) val[35 + q] = v;
#pragma unroll
                for (int a = 0; a < 6; ++a) v -= V[a * NBQ + q] * CV[a * NBQ + q2];
                val[c++] = v;
            }
            const double gq = d.lmG[(size_t)q * d.Lpad + l];
            double v = gq;
#pragma unroll
            for (int a = 0; a < 6; ++a) v -= V[a * NBQ + q] * Cg[a];
            val[28 + q] = v;
            val[42 + q] = gq;
        }
    }
    smat[t] = mask ? (uint8_t)d.lm_mat[l] : (uint8_t)0xFF;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < BSP; ++c) sv[t * BSP + c] = val[p * BSP + c];
        __syncthreads();
        const int cnt = (NBV - p * BSP) < BSP ? (NBV - p * BSP) : BSP;
        if (t < cnt * d.M) {
            const int comp = t / d.M, m = t - comp * d.M;
            double acc = 0.0;
            for (int i = 0; i < 256; ++i)
                if (smat[i] == m) acc += sv[i * BSP + comp];
            d.part_b[((size_t)blockIdx.x * d.M + m) * NBV + p * BSP + comp] = acc;
        }
    }
}

// column sums of the per-block partials: one wave per (material, component), fixed summation order
__global__ __launch_bounds__(64) void k_ph_border_colsum(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int idx = blockIdx.x, lane = threadIdx.x;
    double a = 0.0;
    for (int b = lane; b < d.n_lm_blocks; b += 64) a += d.part_b[(size_t)b * d.M * NBV + idx];
    a = wave_sum(a);
    if (lane == 0) d.part_b[(size_t)d.n_lm_blocks * d.M * NBV + idx] = a;
}

// scatters the summed partials into the border system (one block): thread (c, c2) owns entry (c, c2) of S_bb and
// collects its contributions -- one per material for a material's own columns, all materials for the light columns --
// in material order (the first version did the whole scatter on one lane: 35 us)
__global__ __launch_bounds__(1024) void k_ph_border_reduce(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double tot[SSBA_MAX_MATERIALS_DEV * NBV];
    const int t = threadIdx.x;
    for (int idx = t; idx < d.M * NBV; idx += 1024) tot[idx] = d.part_b[(size_t)d.n_lm_blocks * d.M * NBV + idx];
    __syncthreads();
    const int c = t / NBP, c2 = t - c * NBP;
    double sv = 0.0, rhs = 0.0, hh = 0.0, gg = 0.0;
    for (int m = 0; m < d.M; ++m) {
        const double *tm = tot + m * NBV;
        int q = -1, q2 = -1;
#pragma unroll
        for (int x = 0; x < NBQ; ++x) {
            const int cc = bcol(d, (uint32_t)m, x);
            if (cc == c) q = x;
            if (cc == c2) q2 = x;
        }
        if (q >= 0 && q2 >= 0) sv += tm[q <= q2 ? tri7(q, q2) : tri7(q2, q)];
        if (q >= 0 && c2 == 0) { rhs += tm[28 + q]; hh += tm[35 + q]; gg += tm[42 + q]; }
    }
    d.bsys[BS_SBB + c * NBP + c2] = (c < d.nb && c2 < d.nb) ? sv : 0.0;
    if (c2 == 0) {
        d.bsys[BS_RHS + c] = c < d.nb ? rhs : 0.0;
        d.bsys[BS_H + c] = c < d.nb ? hh : 0.0;
        d.bsys[BS_G + c] = c < d.nb ? gg : 0.0;
        if (st.iteration == 0 && c < d.nb) d.bsys[BS_S + c] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(hh)) : 1.0;
    }
}

// one block per free pose, every iteration: its six rows of S_pb = H_pb - sum_j Y_j V_j.  The pose's
// observation references are sorted by material, so a thread accumulates one material's four columns
// at a time (plus the three light columns throughout) and the block reduces them in a fixed order.
// One wave per pose: a pass covers one material's ~N_pose/M observations (C3: ~300), so a narrow block wastes few
// lanes on the ragged last iteration of every pass (256 lanes: 531 us, 128: 440 us, 64: 400 us; splitting a pose over
// several waves brought nothing: the kernel is bound by the dependent gathers of one observation, not by occupancy)
constexpr int BP_THREADS = 64;
template <bool DN> __global__ __launch_bounds__(BP_THREADS) void k_ph_border_poses(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int k = blockIdx.x, f = d.pose_free[k];
    if (f < 0) return;
    __shared__ double sm[BP_THREADS / 64][24];
    const int t = threadIdx.x;
    const double *T = d.poses + (size_t)k * 12;
    double accL[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) accL[i] = 0.0;
    for (int m = 0; m <= d.M; ++m) {
        double accM[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) accM[i] = 0.0;
        if (m < d.M) {
            const uint32_t b = d.pose_mat_start[(size_t)k * (d.M + 1) + m], e = d.pose_mat_start[(size_t)k * (d.M + 1) + m + 1];
            for (uint32_t i = b + t; i < e; i += BP_THREADS) {
                int l;
                size_t oi;
                pose_list_entry<DN>(d, i, l, oi);
                LmIn x;
                load_lm(d, l, x);
                const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
                ObsPh o;
                obs_ph_linearize(d, d.sh, T, x.p, x.n, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, true, o);
                double Ci[21];
#pragma unroll
                for (int c = 0; c < 21; ++c) Ci[c] = d.cinv[(size_t)c * d.Lpad + l];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double w[6], y[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int r = 0; r < 7; ++r) if (jl_nz(r, c)) v += o.Jp[6 * r + a] * o.Jl[6 * r + c];
                        w[c] = v;
                    }
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < 6; ++q) v += w[q] * Ci[q <= c ? tri6(q, c) : tri6(c, q)];
                        y[c] = v;
                    }
#pragma unroll
                    for (int q = 0; q < NBQ; ++q) {
                        double v = o.Jp[18 + a] * o.jb[q];
#pragma unroll
                        for (int c = 0; c < 6; ++c) v -= y[c] * d.lmV[(size_t)(c * NBQ + q) * d.Lpad + l];
                        if (q < 4) accM[a * 4 + q] += v;
                        else accL[a * 3 + (q - 4)] += v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) accM[i] = accL[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const double v = wave_sum(accM[i]);
            if ((t & 63) == 0) sm[t >> 6][i] = v;
        }
        __syncthreads();
        if (t < 24) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < BP_THREADS / 64; ++w) v += sm[w][t];
            int a, col;
            if (m < d.M) { a = t / 4; col = bcol(d, (uint32_t)m, t - a * 4); }
            else { a = t / 3; col = (t < 18) ? bcol(d, 0u, 4 + (t - a * 3)) : -1; }
            if (col >= 0) d.Spb[((size_t)f * 6 + a) * NBP + col] = v;
        }
    }
}

// candidate shared blocks = Plus(x_b, delta_b): Euclidean, UnitVectorPerturbation on a directional
// light (dataset_ba_phong.cpp:201-204); its |dx|^2 and non-finite flag join the pose partials
// (one lane per entry of the shared blocks [light 3 | phong 3 M | texture M], nsh <= 63: the serial version spent 18 us in
// chains of dependent loads; |dx|^2 is still summed in index order)
// (the body lives in ssba_device.h: it usually runs as the last work-group of k_pose_update; this launch is left for the
// iterations whose poses the reduced solve has already moved)
__global__ __launch_bounds__(64) void k_ph_border_update(Dev d, int ls_round) {
    __shared__ double sdf[64];
    if (blockIdx.x == 0) ph_border_update_block(d, ls_round, sdf);
}

// ------------------------------------------------------------------ dogleg (config 3) ---
// per landmark: Gauss-Newton back-substitution, v_l, the landmark parts of the norms and the products
// |J v|^2, |J delta_gn|^2, (J v).(J delta_gn) over the landmark's observations
template <bool DN> __global__ __launch_bounds__(256, 2) void k_ph_dogleg_gn(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double sums[NDL];
#pragma unroll
    for (int q = 0; q < NDL; ++q) sums[q] = 0.0;
    double dl[6] = {0, 0, 0, 0, 0, 0}, vl[6] = {0, 0, 0, 0, 0, 0};
    if (mask && !st.step_failed) {
        const PhSlots<DN> sl(d, l, mask);
        LmIn x;
        load_lm(d, l, x);
        double gl[6], tt[6], tv[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) { gl[c] = d.gl[(size_t)c * d.Lpad + l]; tt[c] = gl[c]; tv[c] = 0.0; }
        double gbq[NBQ], vbq[NBQ];   // border Gauss-Newton step and v seen by this landmark's material
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int c = d.nb ? bcol(d, x.mat, q) : -1;
            gbq[q] = c >= 0 ? d.bsys[BS_DB + c] : 0.0;
            vbq[q] = c >= 0 ? d.bsys[BS_VB + c] : 0.0;
        }
        // ONE pass over the landmark's rows (r04: a second one formed J v and J gn row by row).  With e_g = J_p gn_p + J_b gn_b and
        // e_v = J_p v_p + J_b v_b of a row,  J gn = J_l dl + e_g  and  J v = J_l vl + e_v,  so over the landmark's rows
        //   |J gn|^2 = dl^T H_ll dl + 2 dl . sum J_l^T e_g + sum e_g^2     (H_ll = sum J_l^T J_l: the linearisation's block,
        //   |J v|^2  = vl^T H_ll vl + 2 vl . sum J_l^T e_v + sum e_v^2      rows of constant poses included)
        //   Jv . Jgn = vl^T H_ll dl + vl . sum J_l^T e_g + dl . sum J_l^T e_v + sum e_v e_g
        // and  sum J_l^T e_g = tt - g_l  is what the back-substitution forms anyway (the scheme of k_ph_backsub_eval's model cost).
        double see = 0.0, svv = 0.0, sev = 0.0;
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const int f = d.pose_free[k];
            if (f < 0 && !d.nb) continue;       // (a constant pose's intensity row still has e = J_b . border step with free shared blocks)
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            double dp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, vp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (f >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) { dp[c] = d.x0[(size_t)f * 6 + c]; vp[c] = d.vp[(size_t)k * 6 + c]; }
            }
            ph_rows(d, d.sh, d.poses + (size_t)k * 12, x, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, f >= 0,
                    [&](auto part, int, double, const double *jp, const double *jl, const double *jb) {
                        constexpr int P = decltype(part)::value;
                        double jd = 0.0, jv = 0.0;
                        if (f >= 0) {
#pragma unroll
                            for (int c = 0; c < 6; ++c) { jd += jp[c] * dp[c]; jv += jp[c] * vp[c]; }
#pragma unroll
                            for (int c = ph_jl_lo(P); c < ph_jl_hi(P); ++c) { tt[c] += jl[c] * jd; tv[c] += jl[c] * jv; }
                        }
                        double eg = jd, ev = jv;      // (the border part of sum J_l^T e comes from V_j below, as before)
                        if (P == 1) {
#pragma unroll
                            for (int q = 0; q < NBQ; ++q) { eg += jb[q] * gbq[q]; ev += jb[q] * vbq[q]; }
                        }
                        see += eg * eg; svv += ev * ev; sev += ev * eg;
                    });
        }
        if (d.nb) {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int q = 0; q < NBQ; ++q) {
                    const double V = d.lmV[(size_t)(a * NBQ + q) * d.Lpad + l];
                    tt[a] += V * gbq[q];
                    tv[a] += V * vbq[q];
                }
        }
        double Ci[21];
#pragma unroll
        for (int c = 0; c < 21; ++c) Ci[c] = d.cinv[(size_t)c * d.Lpad + l];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) v += Ci[q <= a ? tri6(q, a) : tri6(a, q)] * tt[q];
            dl[a] = -v;
        }
        double H[21];
#pragma unroll
        for (int c = 0; c < 21; ++c) H[c] = d.hll[(size_t)c * d.Lpad + l];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double s = d.sl[(size_t)c * d.Lpad + l], s2 = s * s;
            const double D2 = fmin(fmax(H[tri6(c, c)] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag);
            vl[c] = s2 * gl[c] / D2;
            sums[0] += s2 * gl[c] * gl[c] / D2;
            sums[1] += D2 * dl[c] * dl[c] / s2;
            sums[2] += gl[c] * dl[c];
        }
        double Hd[6], Hv[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double hd = 0.0, hv = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double h = H[q <= a ? tri6(q, a) : tri6(a, q)];
                hd += h * dl[q];
                hv += h * vl[q];
            }
            Hd[a] = hd; Hv[a] = hv;
        }
        double vHv = 0.0, dHd = 0.0, vHd = 0.0, vtv = 0.0, dtg = 0.0, vtg = 0.0, dtv = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double tg = tt[c] - gl[c];
            vHv += vl[c] * Hv[c]; dHd += dl[c] * Hd[c]; vHd += vl[c] * Hd[c];
            vtv += vl[c] * tv[c]; dtg += dl[c] * tg; vtg += vl[c] * tg; dtv += dl[c] * tv[c];
        }
        sums[3] = vHv + 2.0 * vtv + svv;
        sums[4] = dHd + 2.0 * dtg + see;
        sums[5] = vHd + vtg + dtv + sev;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        d.dl_gn[(size_t)c * d.Lpad + l] = dl[c];
        d.vl[(size_t)c * d.Lpad + l] = vl[c];
    }
#pragma unroll
    for (int q = 0; q < NDL; ++q) {
        const double v = block_sum(sums[q], sm);
        if (threadIdx.x == 0) d.part_dl[(size_t)blockIdx.x * NDL + q] = v;
    }
}

// per landmark: delta_l = beta * gn + gamma * v, candidate point / normal, model cost change, candidate cost
template <bool DN> __global__ __launch_bounds__(256, 2) void k_ph_dogleg_eval(Dev d) {
    const State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double ccost = 0.0, mcc = 0.0, dn = 0.0, nonfinite = 0.0;
    LmIn x;
    load_lm(d, l, x);
    double np_[3] = {x.p[0], x.p[1], x.p[2]}, nn[3] = {x.n[0], x.n[1], x.n[2]};
    double dl[6] = {0, 0, 0, 0, 0, 0};
    if (mask && !st.step_failed) {
        const PhSlots<DN> sl(d, l, mask);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            dl[c] = st.beta * d.dl_gn[(size_t)c * d.Lpad + l] + st.gamma * d.vl[(size_t)c * d.Lpad + l];
            if (!isfinite(dl[c])) nonfinite = 1.0;
        }
        np_[0] = x.p[0] + dl[0]; np_[1] = x.p[1] + dl[1]; np_[2] = x.p[2] + dl[2];
        unit_plus(x.n, dl + 3, nn);
        dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2] + (nn[0] - x.n[0]) * (nn[0] - x.n[0]) +
             (nn[1] - x.n[1]) * (nn[1] - x.n[1]) + (nn[2] - x.n[2]) * (nn[2] - x.n[2]);
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            const double u = d.ou[oi], v = d.ov[oi], dd = d.od[oi], inten = d.oi[oi];
            // (the model cost change comes from the six sums of the dogleg model: k_dogleg_interp, State::dl_mcc -- r04: this
            // kernel made a full Jacobian pass for it)
            ccost += obs_ph_cost(d, d.cand_sh, d.cand_poses + (size_t)k * 12, np_, nn, x.mat, u, v, dd, inten, nobs);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d.cand_pts[(size_t)c * d.Lpad + l] = np_[c];
        d.cand_nrm[(size_t)c * d.Lpad + l] = nn[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) d.dlm[(size_t)c * d.Lpad + l] = dl[c];
    if (d.constrained) ph_ls_dir_terms(d, l, mask && !st.step_failed, dl, sm);
    const double a = block_sum(ccost, sm), b = block_sum(mcc, sm), c = block_sum(dn, sm), e = block_sum(nonfinite, sm);
    if (threadIdx.x == 0) {
        d.part_eval[blockIdx.x * 4 + 0] = a;
        d.part_eval[blockIdx.x * 4 + 1] = b;
        d.part_eval[blockIdx.x * 4 + 2] = c;
        d.part_eval[blockIdx.x * 4 + 3] = e;
    }
}

// ------------------------------------------------------------------ projected line search ---
// [Ceres 1.x TrustRegionMinimizer::DoLineSearch] phi(a) = cost(Plus(x, a * delta)) and
// phi'(a) = delta . gradient(Plus(x, a * delta)) = sum_obs r . (J delta), r and J evaluated at the
// trial point; the scalar Armijo logic (ssba_linesearch.h) runs in k_ph_ls_reduce for the rounds enqueued with the
// iteration, and on the host for what those cannot finish.
__global__ void k_ls_set_alpha(Dev d, double alpha) {
    if (threadIdx.x == 0 && blockIdx.x == 0) d.st->ls_alpha = alpha;
}

// per landmark: trial point for st.ls_alpha (candidate poses / shared blocks are already in place),
// its cost, phi', |dx_l|^2, and the alpha-independent max|delta_l| and g_l . delta_l
template <bool DN> __global__ __launch_bounds__(256, 2) void k_ph_ls_probe(Dev d, int ls_round) {
    const State &st = *d.st;
    if (st.terminated || (ls_round && !st.ls_active)) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double cost = 0.0, dphi = 0.0, dn = 0.0, nonfinite = 0.0, dmax = 0.0, gd = 0.0;
    LmIn x;
    load_lm(d, l, x);
    double np_[3] = {x.p[0], x.p[1], x.p[2]}, nn[3] = {x.n[0], x.n[1], x.n[2]};
    if (mask && !st.step_failed) {
        const PhSlots<DN> sl(d, l, mask);
        double dl[6], sdl[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            dl[c] = d.dlm[(size_t)c * d.Lpad + l];
            sdl[c] = st.ls_alpha * dl[c];
            if (!isfinite(dl[c])) nonfinite = 1.0;
            dmax = fmax(dmax, fabs(dl[c]));
            gd += d.gl[(size_t)c * d.Lpad + l] * dl[c];
        }
        double dbq[NBQ];
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int c = d.nb ? bcol(d, x.mat, q) : -1;
            dbq[q] = c >= 0 ? st.beta * d.bsys[BS_DB + c] + st.gamma * d.bsys[BS_VB + c] : 0.0;
        }
        np_[0] = x.p[0] + sdl[0]; np_[1] = x.p[1] + sdl[1]; np_[2] = x.p[2] + sdl[2];
        unit_plus(x.n, sdl + 3, nn);
        dn = sdl[0] * sdl[0] + sdl[1] * sdl[1] + sdl[2] * sdl[2] + (nn[0] - x.n[0]) * (nn[0] - x.n[0]) +
             (nn[1] - x.n[1]) * (nn[1] - x.n[1]) + (nn[2] - x.n[2]) * (nn[2] - x.n[2]);
        for (int s = 0; s < sl.count(); ++s) {
            if (!sl.has(s)) continue;
            const uint32_t k = sl.pose(d, s);
            const int f = d.pose_free[k];
            const size_t oi = sl.at(s);
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            double dp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (f >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    dp[c] = st.opt.strategy ? st.beta * d.x0[(size_t)f * 6 + c] + st.gamma * d.vp[(size_t)k * 6 + c] : d.x0[(size_t)f * 6 + c];
            }
            LmIn xt;
#pragma unroll
            for (int c = 0; c < 3; ++c) { xt.p[c] = np_[c]; xt.n[c] = nn[c]; }
            xt.mat = x.mat;
            cost += ph_rows(d, d.cand_sh, d.cand_poses + (size_t)k * 12, xt, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, f >= 0,
                            [&](auto part, int, double r, const double *jp, const double *jl, const double *jb) {
                                constexpr int P = decltype(part)::value;
                                double jd = 0.0;
#pragma unroll
                                for (int c = ph_jl_lo(P); c < ph_jl_hi(P); ++c) jd += jl[c] * dl[c];
                                if (f >= 0) {
#pragma unroll
                                    for (int c = 0; c < 6; ++c) jd += jp[c] * dp[c];
                                }
                                if (P == 1) {
#pragma unroll
                                    for (int q = 0; q < NBQ; ++q) jd += jb[q] * dbq[q];
                                }
                                dphi += r * jd;
                            });
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d.cand_pts[(size_t)c * d.Lpad + l] = np_[c];
        d.cand_nrm[(size_t)c * d.Lpad + l] = nn[c];
    }
    const double a = block_sum(cost, sm), b = block_sum(dphi, sm), c = block_sum(dn, sm), e = block_sum(nonfinite, sm);
    const double f2 = block_max(dmax, sm), g2 = block_sum(gd, sm);
    if (threadIdx.x == 0) {
        double *o = d.part_ls + (size_t)blockIdx.x * NLS;
        o[0] = a; o[1] = b; o[2] = c; o[3] = e; o[4] = f2; o[5] = g2;
    }
}

// ---- the common case of the projected line search without the host ----------------------------------------------
// [line_search.cc ArmijoLineSearch::DoSearch] The first sample is the full step, and it is accepted iff
//     phi(1) <= phi(0) + 1e-4 phi'(0),    phi(1) = the candidate cost the evaluation kernel has just formed,
//     phi'(0) = g . delta (every block: landmarks, poses, shared blocks).
// phi'(1), which Ceres also evaluates there, is only USED by the interpolation of a rejected sample.  So the accepted full
// step needs no probe at all: the evaluation kernel forms the landmark part of g . delta and max|delta| (the search's
// min_step_size test; ph_ls_dir_terms), k_ph_ls_fast adds the pose / border parts and decides.  A rejected full step parks the solver
// (terminated with LS_PENDING) until the host has driven the search (ssba_api.hip: finish_pending_search) -- exactly the
// evaluations Ceres makes, starting with phi'(1).
__device__ inline void ls_park(State &st) {
    st.ls_active = 0;
    st.ls_pending = 1;
    st.terminated = 1;
    st.termination_type = LS_PENDING;
}
// The border entries of the step (delta_b = beta db + gamma vb) and of the gradient for lane 0's loop below: loaded by nb lanes
// at once into LDS (lane 0 walked them in global memory up to r04 -- 23 dependent round trips in front of the decision).  The
// caller's next barrier (block_sum) publishes them; the additions keep their order.
static __device__ __forceinline__ void ls_stage_border(const Dev &d, const State &st, double *sv, double *sg) {
    const int c = threadIdx.x;
    if (c < d.nb) {
        sv[c] = st.beta * d.bsys[BS_DB + c] + st.gamma * d.bsys[BS_VB + c];
        sg[c] = d.bsys[BS_G + c];
    }
}
// The pose part of max|delta| and g . delta for a 1024-lane work-group: entry i = 6 f + c of the pose step.  Eight trips' loads in
// flight at a time, the pose index first and what depends on it behind (the rolled loop paid two dependent round trips per
// trip: free_pose, then vp -- twelve in a row at C3, most of k_ph_ls_fast's 15 us); every lane still adds its entries in ascending order.
static __device__ __forceinline__ void ls_pose_terms(const Dev &d, const State &st, double &pmax, double &pgd) {
    constexpr int Q = 8;
    const int n = d.nfree * 6;
    const bool dog = st.opt.strategy != 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * Q) {
        int k[Q];
        double x[Q], g[Q], v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = i0 + 1024 * q;
            k[q] = (dog && i < n) ? d.free_pose[i / 6] : 0;
            x[q] = i < n ? d.x0[i] : 0.0;
            g[q] = i < n ? d.xv[d.off_gp + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = i0 + 1024 * q, c = i - (i / 6) * 6;
            v[q] = (dog && i < n) ? d.vp[(size_t)k[q] * 6 + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = i0 + 1024 * q;
            if (i >= n) continue;
            const double dpc = dog ? st.beta * x[q] + st.gamma * v[q] : x[q];
            pmax = fmax(pmax, fabs(dpc));
            pgd += g[q] * dpc;
        }
    }
}
// (1024 lanes: the loops below are chains of cold loads, six trips instead of 24 -- 29 us -> see profiles/README.md)
// n_eval_parts > 0: the launch forms the evaluation sums first (k_reduce_eval's launch up to r04: the same 256 lanes add the
// same partials in the same order, the other waves add zeros)
// x_world > 0 (landmark sharding): the landmark part of max|delta| and g . delta comes from the exchanged vector (k_ph_ls_pack)
__global__ __launch_bounds__(1024) void k_ph_ls_fast(Dev d, int n_eval_parts, int x_world) {
    State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[16];
    if (n_eval_parts > 0) {
        double a = 0.0, b = 0.0, c = 0.0, e = 0.0;
        if (threadIdx.x < 256)
            for (int i = threadIdx.x; i < n_eval_parts; i += 256) {
                a += d.part_eval[i * 4];
                b += d.part_eval[i * 4 + 1];
                c += d.part_eval[i * 4 + 2];
                e += d.part_eval[i * 4 + 3];
            }
        a = block_sum(a, sm);
        b = block_sum(b, sm);
        c = block_sum(c, sm);
        e = block_sum(e, sm);
        if (threadIdx.x == 0) { d.scal2[0] = a; d.scal2[1] = b; d.scal2[2] = c; d.scal2[3] = e; }      // (read again by this lane below)
    }
    double lmax = 0.0, lgd = 0.0;
    if (!x_world)
        for (int i = threadIdx.x; i < d.n_lm_blocks; i += 1024) {
            const double *o = d.part_ls + (size_t)i * NLS;
            lmax = fmax(lmax, o[4]); lgd += o[5];
        }
    double pmax = 0.0, pgd = 0.0, pbad = 0.0;
    ls_pose_terms(d, st, pmax, pgd);
    for (int i = threadIdx.x; i < d.n_pose_blocks + (d.nb ? 1 : 0); i += 1024) pbad += d.part_pose[i * NPP + 1];
    __shared__ double sbv[NBP], sbg[NBP];
    ls_stage_border(d, st, sbv, sbg);
    lmax = block_max(lmax, sm); lgd = block_sum(lgd, sm);
    pmax = block_max(pmax, sm); pgd = block_sum(pgd, sm); pbad = block_sum(pbad, sm);
    if (threadIdx.x != 0) return;
    if (x_world) {
        const double *x = d.ls_out + NLS_OUT + NLS_MACH;
        lgd = x[4];
        for (int r = 0; r < x_world; ++r) lmax = fmax(lmax, x[NLS_X + r]);
    }
    double bmax = 0.0, bgd = 0.0;
    for (int c = 0; c < d.nb; ++c) {
        const double v = sbv[c];
        bmax = fmax(bmax, fabs(v));
        bgd += sbg[c] * v;
    }
    // ComputeTrustRegionStep: the search only runs on a valid step (as k_ph_ls_reduce's ls_out[6])
    const bool valid = !st.step_failed && (d.scal2[3] + pbad) == 0.0 && (st.opt.strategy ? st.dl_mcc : d.scal2[1]) > 0.0;
    if (!valid) return;
    const double phi0 = st.x_cost, dphi0 = lgd + pgd + bgd, phi1 = d.scal2[0];
    const bool value_ok = isfinite(phi1);
    if (value_ok && !(phi1 > phi0 + 1e-4 * dphi0 * 1.0)) { st.ls_steps += 1; return; }       // the full step satisfies the Armijo condition: nothing to search (one evaluation in Ceres' count)
    if (d.ls_rounds > 0) { st.ls_active = 1; return; }                // the search rounds behind this launch wake up
    ls_park(st);
}
// Landmark sharding with bounds: this rank's landmark sums of the last evaluation (k_ph_ls_probe; or only the direction terms the
// evaluation kernel of the iteration left: slots 4 and 5 of the partials) packed for the exchange -- layout at NLS_X (ssba_types.h).
// The maximum travels through the SUM exchange in one slot per rank.  Always writes the vector (zeros while the solver is parked).
__global__ __launch_bounds__(1024) void k_ph_ls_pack(Dev d, int rank, int world) {
    const State &st = *d.st;
    __shared__ double sm[16];
    double acc[NLS] = {0, 0, 0, 0, 0, 0};
    if (!st.terminated)
        for (int i = threadIdx.x; i < d.n_lm_blocks; i += 1024) {
            const double *o = d.part_ls + (size_t)i * NLS;
            acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; acc[4] = fmax(acc[4], o[4]); acc[5] += o[5];
        }
    const double cost = block_sum(acc[0], sm), dphi = block_sum(acc[1], sm), dn = block_sum(acc[2], sm), bad = block_sum(acc[3], sm);
    const double lmax = block_max(acc[4], sm), lgd = block_sum(acc[5], sm);
    if (threadIdx.x != 0) return;
    double *x = d.ls_out + NLS_OUT + NLS_MACH;
    x[0] = cost; x[1] = dphi; x[2] = dn; x[3] = bad; x[4] = lgd; x[5] = 0.0; x[6] = 0.0; x[7] = 0.0;
    for (int r = 0; r < world; ++r) x[NLS_X + r] = r == rank ? lmax : 0.0;
}
__global__ void k_ls_resume(Dev d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    State &st = *d.st;
    if (!st.ls_pending) return;
    st.ls_pending = 0;
    st.terminated = 0;
    st.termination_type = 1;
}

// one block: totals of the probe plus the pose / border parts of max|delta| and g . delta, and the
// validity of the trust-region step (as k_decide will judge it)
// ls_round: a round of the device-side search -- lane 0 then feeds the evaluation to the Armijo state machine: the search
// ends (accepted sample: its sums replace the evaluation kernel's for k_decide, as k_ph_ls_accept does for the host), goes
// on (st.ls_alpha = the next step: the following round evaluates it) or is handed to the host (`last` round and still not
// done, or a failed search that would have to restore the full step: the host starts that search again from the top --
// the evaluations are deterministic, it takes the same path)
// x_world > 0 (landmark sharding): the landmark sums come from the exchanged vector (k_ph_ls_pack) instead of this rank's partials
__global__ __launch_bounds__(1024) void k_ph_ls_reduce(Dev d, int ls_round, int last, int x_world) {
    State &st = *d.st;
    if (st.terminated || (ls_round && !st.ls_active)) return;
    __shared__ double sm[16];
    double acc[NLS] = {0, 0, 0, 0, 0, 0};
    if (!x_world)
        for (int i = threadIdx.x; i < d.n_lm_blocks; i += 1024) {
            const double *o = d.part_ls + (size_t)i * NLS;
            acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; acc[4] = fmax(acc[4], o[4]); acc[5] += o[5];
        }
    double pmax = 0.0, pgd = 0.0, pbad = 0.0;
    ls_pose_terms(d, st, pmax, pgd);
    for (int i = threadIdx.x; i < d.n_pose_blocks + (d.nb ? 1 : 0); i += 1024) pbad += d.part_pose[i * NPP + 1];
    __shared__ double sbv[NBP], sbg[NBP];
    ls_stage_border(d, st, sbv, sbg);
    double cost = block_sum(acc[0], sm), dphi = block_sum(acc[1], sm), dn = block_sum(acc[2], sm), bad = block_sum(acc[3], sm);
    double lmax = block_max(acc[4], sm), lgd = block_sum(acc[5], sm);
    const double qmax = block_max(pmax, sm), qgd = block_sum(pgd, sm), qbad = block_sum(pbad, sm);
    if (threadIdx.x != 0) return;
    if (x_world) {
        const double *x = d.ls_out + NLS_OUT + NLS_MACH;
        cost = x[0]; dphi = x[1]; dn = x[2]; bad = x[3]; lgd = x[4];
        for (int r = 0; r < x_world; ++r) lmax = fmax(lmax, x[NLS_X + r]);
    }
    double bmax = 0.0, bgd = 0.0;
    for (int c = 0; c < d.nb; ++c) {
        const double v = sbv[c];
        bmax = fmax(bmax, fabs(v));
        bgd += sbg[c] * v;
    }
    d.ls_out[0] = cost; d.ls_out[1] = dphi; d.ls_out[2] = dn; d.ls_out[3] = bad;
    d.ls_out[4] = fmax(fmax(lmax, qmax), bmax);
    d.ls_out[5] = lgd + qgd + bgd;
    // ComputeTrustRegionStep: valid iff the solve succeeded, the step is finite and model_cost_change > 0
    d.ls_out[6] = (!st.step_failed && (d.scal2[3] + qbad) == 0.0 && (st.opt.strategy ? st.dl_mcc : d.scal2[1]) > 0.0) ? 1.0 : 0.0;
    d.ls_out[7] = st.x_cost;
    if (!ls_round) return;
    static_assert(sizeof(Armijo) <= NLS_MACH * sizeof(double), "the search state lives behind ls_out");
    Armijo &a = *reinterpret_cast<Armijo *>(d.ls_out + NLS_OUT);
    if (st.ls_active == 1) {
        a.begin(st.x_cost, d.ls_out[5], d.ls_out[4]);
        st.ls_active = 2;
    }
    a.feed(cost, dphi);
    if (a.done && (a.success || st.ls_alpha == 1.0)) {
        if (st.ls_alpha != 1.0) { d.scal2[0] = cost; d.scal2[2] = dn; d.scal2[3] = bad; }
        st.ls_active = 0;
        st.ls_steps += a.num_feeds;
        st.ls_searches += 1;
    } else if (a.done || last) {
        ls_park(st);
    } else {
        st.ls_alpha = a.current.x;
    }
}

// the accepted trial point becomes the candidate k_decide judges; the model cost change of the
// unscaled trust-region step stays [trust_region_minimizer.cc: DoLineSearch only rescales delta]
__global__ void k_ph_ls_accept(Dev d) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || d.st->terminated) return;
    d.scal2[0] = d.ls_out[0];
    d.scal2[2] = d.ls_out[2];
    d.scal2[3] = d.ls_out[3];
}

// (with free shared blocks the inversion also forms M V from k_ph_border_landmarks' output, which runs after the poses;
// SSBA_PH_INVERT_LAUNCH=1 keeps the launch of its own: A/B, tests)
static bool ph_invert_with_poses(const Dev &d) {
    const char *e = getenv("SSBA_PH_INVERT_LAUNCH");
    return !d.dense && !d.nb && !(e && e[0] == '1');
}
void launch_ph_linearize(Launcher &L, const Dev &d) {
    static const bool separate = [] { const char *e = getenv("SSBA_PH_LIN_LAUNCHES"); return e && e[0] == '1'; }();
    if (!separate && d.nb && !d.dense && !ph_invert_with_poses(d)) {
        LAUNCH(KC_LIN_LM, k_ph_linearize_all, dim3(d.n_lm_blocks + xcd_contiguous_grid(d.P)), dim3(256), 0, d, d.n_lm_blocks);
        LAUNCH(KC_BORDER, k_ph_border_landmarks<false>, dim3(d.n_lm_blocks), dim3(256), 0, d);
        if (d.lmMV) LAUNCH(KC_BORDER, k_ph_hpb, dim3(d.P * d.M), dim3(64), 0, d);
        return;
    }
    LAUNCH(KC_LIN_LM, (d.dense ? k_ph_linearize_landmarks<true> : k_ph_linearize_landmarks<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
    // two observations in flight per lane: 54 us against 60 for the rolled loop at C3; three need 274 registers and lose (75 us),
    // 128 lanes per pose lose too (64-76 us) -- unlike the stereo kernel, whose 175 registers leave room for five
    if (d.dense) LAUNCH(KC_LIN_POSE, (k_ph_linearize_poses<true, 256, 1>), dim3(xcd_contiguous_grid(d.P)), dim3(256), 0, d);
    else if (ph_invert_with_poses(d)) LAUNCH(KC_LIN_POSE, k_ph_linpose_invert, dim3(xcd_contiguous_grid(d.P) + d.n_lm_blocks), dim3(256), 0, d);
    else LAUNCH(KC_LIN_POSE, (k_ph_linearize_poses<false, 256, 2>), dim3(xcd_contiguous_grid(d.P)), dim3(256), 0, d);
    if (d.nb) LAUNCH(KC_BORDER, (d.dense ? k_ph_border_landmarks<true> : k_ph_border_landmarks<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
    if (d.lmMV) LAUNCH(KC_BORDER, k_ph_hpb, dim3(d.P * d.M), dim3(64), 0, d);
}
// general layout: per observation W = J_p^T J_l (6x6 over the 7 residual rows) and Y = W C^-1, stored for the
// pair-list Schur kernel of ssba_dense.hip
__global__ __launch_bounds__(256) void k_ph_dn_wy(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= d.n_obs) return;
    const uint32_t k = d.dn_obs_pose[e];
    if (d.pose_free[k] < 0) return;
    const int l = (int)d.dn_obs_lm[e];
    LmIn x;
    load_lm(d, l, x);
    const double nobs[3] = {d.onx[e], d.ony[e], d.onz[e]};
    ObsPh o;
    obs_ph_linearize(d, d.sh, d.poses + (size_t)k * 12, x.p, x.n, x.mat, d.ou[e], d.ov[e], d.od[e], d.oi[e], nobs, true, o);
    // one factor Z = W M^T per observation (C^-1 = M^T M, M from k_ph_invert) for both sides of every pair product, and
    // u = M g_l per landmark for the right-hand side: see k_dn_wy (ssba_dense.hip)
    double Mf[21];
#pragma unroll
    for (int c = 0; c < 21; ++c) Mf[c] = d.cfac[(size_t)c * d.Lpad + l];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b <= c; ++b) v += Mf[c * (c + 1) / 2 + b] * d.gl[(size_t)b * d.Lpad + l];
        d.dn_Mg[(size_t)c * d.Lpad + l] = v;
    }
    double *Z = d.dn_Y + (size_t)d.dn_zpos[e] * 36;       // pose-major record (d.dn_W is the same buffer)
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double w[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
#pragma unroll
            for (int m = 0; m < 7; ++m) if (jl_nz(m, c)) v += o.Jp[6 * m + a] * o.Jl[6 * m + c];
            w[c] = v;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
#pragma unroll
            for (int b = 0; b <= c; ++b) v += w[b] * Mf[c * (c + 1) / 2 + b];
            Z[6 * a + c] = v;
        }
    }
}
void launch_ph_dense_wy(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_ph_invert, dim3(d.n_lm_blocks), dim3(256), 0, d);
    LAUNCH(KC_SCHUR, k_ph_dn_wy, dim3((d.n_obs + 255) / 256), dim3(256), 0, d);
}
void launch_ph_dense_border(Launcher &L, const Dev &d) {
    LAUNCH(KC_BORDER, k_ph_border_schur, dim3(d.n_lm_blocks), dim3(256), 0, d);
    LAUNCH(KC_SMALL, k_ph_border_colsum, dim3(d.M * NBV), dim3(64), 0, d);
    LAUNCH(KC_SMALL, k_ph_border_reduce, dim3(1), dim3(1024), 0, d);
    LAUNCH(KC_BORDER, k_ph_border_poses<true>, dim3(d.P), dim3(BP_THREADS), 0, d);
}

void launch_ph_schur(Launcher &L, const Dev &d, bool check_in_schur) {
    if (!ph_invert_with_poses(d)) LAUNCH(KC_SMALL, k_ph_invert, dim3(d.n_lm_blocks), dim3(256), 0, d);
    const int cp = check_in_schur ? d.n_lm_blocks : 0, xg = check_in_schur ? 1 : 0;
    if (d.lmMV) LAUNCH(KC_SCHUR, k_ph_schur_windows<true>, dim3(d.n_slabs + xg), dim3(PH_THREADS), PhSchurCfg<true>::LDS_DOUBLES * sizeof(double), d, cp);
    else LAUNCH(KC_SCHUR, k_ph_schur_windows<false>, dim3(d.n_slabs + xg), dim3(PH_THREADS), PhSchurCfg<false>::LDS_DOUBLES * sizeof(double), d, cp);
    if (d.nb) {
        LAUNCH(KC_BORDER, k_ph_border_schur, dim3(d.n_lm_blocks), dim3(256), 0, d);
        // pose rows of the border: from the Schur product's border tiles (windowed layout), or pose by pose
        L.spb_rides = false;
        if (d.lmMV) {
            // single GPU: the columns go straight to where they ride through the reduced solve (launch_bcr)
            double *dst = nullptr;
            size_t cnt = 0;
            if (L.spb_in_place_ok && bcr_border_rides(d)) {
                dst = d.pcr.level >= 0 ? d.pcr.Bb : d.lev[0].B;
                cnt = (size_t)(d.pcr.level >= 0 ? d.pcr.n : d.Nsb) * BD * NBP;
                if (cnt > (size_t)d.nf_pad * 6 * NBP) { dst = nullptr; cnt = 0; }      // (Spb has nf_pad rows: never with the plans ssba_finalize makes)
            }
            L.spb_rides = dst != nullptr;
            const size_t entries = std::max((size_t)d.nfree * 6 * NBP, cnt);
            const int n_asm = (int)((entries + 255) / 256);
            LAUNCH(KC_BORDER, k_ph_spb_assemble, dim3(n_asm + d.M * NBV), dim3(256), 0, d, n_asm, dst, cnt);
        } else {
            LAUNCH(KC_SMALL, k_ph_border_colsum, dim3(d.M * NBV), dim3(64), 0, d);
            LAUNCH(KC_BORDER, (d.dense ? k_ph_border_poses<true> : k_ph_border_poses<false>), dim3(d.P), dim3(BP_THREADS), 0, d);
        }
        LAUNCH(KC_SMALL, k_ph_border_reduce, dim3(1), dim3(1024), 0, d);
    }
}
void launch_ph_backsub_eval(Launcher &L, const Dev &d, int fuse_best, bool border_moved) {
    if (d.nb && !border_moved) LAUNCH(KC_SMALL, k_ph_border_update, dim3(1), dim3(64), 0, d, 0);
    LAUNCH(KC_BACKSUB_EVAL, (d.dense ? k_ph_backsub_eval<true> : k_ph_backsub_eval<false>), dim3(d.n_lm_blocks), dim3(256), 0, d, fuse_best);
}
void launch_ph_dogleg_gn(Launcher &L, const Dev &d) {       // (the border part: last work-group of k_dogleg_vec)
    LAUNCH(KC_DOGLEG, (d.dense ? k_ph_dogleg_gn<true> : k_ph_dogleg_gn<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
}
void launch_ph_dogleg_eval(Launcher &L, const Dev &d) {     // (the candidate shared blocks: last work-group of k_pose_update)
    LAUNCH(KC_DOGLEG, (d.dense ? k_ph_dogleg_eval<true> : k_ph_dogleg_eval<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
}
// one evaluation of the line-search function at step `alpha` (alpha < 0: keep the current one)
// stage 0: one GPU.  Landmark sharding: stage 1 = up to this rank's packed landmark sums, stage 2 = from the summed vector on.
void launch_ph_ls_probe(Launcher &L, const Dev &d, double alpha, int moved, int stage, int rank, int world) {
    if (stage != 2) {
        if (alpha >= 0.0) hipLaunchKernelGGL(k_ls_set_alpha, dim3(1), dim3(64), 0, L.stream, d, alpha);
        if (moved) {
            launch_pose_update(L, d);       // (and the shared blocks)
        }
        LAUNCH(KC_BACKSUB_EVAL, (d.dense ? k_ph_ls_probe<true> : k_ph_ls_probe<false>), dim3(d.n_lm_blocks), dim3(256), 0, d, 0);
    }
    if (stage == 1) { launch_ph_ls_pack(L, d, rank, world); return; }
    LAUNCH(KC_SMALL, k_ph_ls_reduce, dim3(1), dim3(1024), 0, d, 0, 0, stage == 2 ? world : 0);
}
void launch_ph_ls_pack(Launcher &L, const Dev &d, int rank, int world) {
    LAUNCH(KC_SMALL, k_ph_ls_pack, dim3(1), dim3(1024), 0, d, rank, world);
}
// the device-side test of the full step (the common case) and the search rounds behind it
void launch_ph_ls_fast(Launcher &L, const Dev &d, bool reduce_eval, int x_world) {
    LAUNCH(KC_SMALL, k_ph_ls_fast, dim3(1), dim3(1024), 0, d, reduce_eval ? eval_parts(d) : 0, x_world);
    // the search itself, enqueued blindly like the trust-region loop: round 0 evaluates phi and phi' at the full step (the
    // candidate of the update kernels is that trial point), every further round moves the candidate to st.ls_alpha first
    for (int r = 0; r < d.ls_rounds; ++r) {
        if (r) {
            launch_pose_update(L, d, 1);        // (and the shared blocks)
        }
        LAUNCH(KC_BACKSUB_EVAL, (d.dense ? k_ph_ls_probe<true> : k_ph_ls_probe<false>), dim3(d.n_lm_blocks), dim3(256), 0, d, 1);
        LAUNCH(KC_SMALL, k_ph_ls_reduce, dim3(1), dim3(1024), 0, d, 1, r == d.ls_rounds - 1 ? 1 : 0, 0);
    }
}
void launch_ls_resume(Launcher &L, const Dev &d) {
    hipLaunchKernelGGL(k_ls_resume, dim3(1), dim3(64), 0, L.stream, d);
}
void launch_ph_ls_accept(Launcher &L, const Dev &d) {
    hipLaunchKernelGGL(k_ph_ls_accept, dim3(1), dim3(64), 0, L.stream, d);
}
int configure_phong() {
    if (hipFuncSetAttribute((const void *)k_ph_schur_windows<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(PhSchurCfg<false>::LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    return hipFuncSetAttribute((const void *)k_ph_schur_windows<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(PhSchurCfg<true>::LDS_DOUBLES * sizeof(double))) == hipSuccess ? 0 : -1;
}

}  // namespace ssba

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

#include "ssba_device.h"
#include "ssba_launch.h"
#include "ssba_phong_device.h"
#include "ssba_types.h"

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
    double half_sq;  // 1/2 |r|^2
};

static __device__ __forceinline__ void obs_ph_linearize(const Dev &d, const double *__restrict__ T, const double p[3],
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
    const double ph3[3] = {d.mat[4 * mat], d.mat[4 * mat + 1], d.mat[4 * mat + 2]};
    double ri, J19[19], rn[3], Jnp[18], Jnn[9];
    intensity_residual(d.light_type, T, p, n, ph3, d.mat[4 * mat + 3], d.light, inten, d.int_stiff, &ri, J19);
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
    o.half_sq = s.half_rho + 0.5 * (ri * ri + rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
}

// residuals only (candidate evaluation)
static __device__ __forceinline__ double obs_ph_cost(const Dev &d, const double *__restrict__ T, const double p[3],
                                                     const double n[3], uint32_t mat, double u, double v, double dd,
                                                     double inten, const double nobs[3]) {
    const double ph3[3] = {d.mat[4 * mat], d.mat[4 * mat + 1], d.mat[4 * mat + 2]};
    double ri, rn[3];
    intensity_residual(d.light_type, T, p, n, ph3, d.mat[4 * mat + 3], d.light, inten, d.int_stiff, &ri, nullptr);
    normal_residual(T, n, nobs, d.Sn, rn, nullptr, nullptr);
    return obs_cost(d, T, p[0], p[1], p[2], u, v, dd) + 0.5 * (ri * ri + rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
}

// UnitVectorPerturbation::operator() (perturbations.hpp:98-102)
static __device__ __forceinline__ void unit_plus(const double x[3], const double dl[3], double out[3]) {
    const double s = (dl[0] * x[0] + dl[1] * x[1] + dl[2] * x[2]) / (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const double y0 = x[0] + dl[0] - s * x[0], y1 = x[1] + dl[1] - s * x[1], y2 = x[2] + dl[2] - s * x[2];
    const double nrm = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
    out[0] = y0 / nrm; out[1] = y1 / nrm; out[2] = y2 / nrm;
}

__device__ __forceinline__ int tri6(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }   // r <= c

// inverse of the damped 6x6 landmark block (packed upper h[21], diagonal damping dmp[6]) through its
// Cholesky factor; result packed upper Ci[21].  false on breakdown.
static __device__ __forceinline__ bool inv6_spd(const double h[21], const double dmp[6], double Ci[21]) {
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
    return true;
}

static __device__ __forceinline__ void ph_damping(const Dev &d, const State &st, int l, const double h[21], double dmp[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double s = d.sl[(size_t)c * d.Lpad + l], s2 = s * s;
        dmp[c] = fmin(fmax(h[tri6(c, c)] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    }
}

struct LmIn { double p[3], n[3]; uint32_t mat; };
static __device__ __forceinline__ void load_lm(const Dev &d, int l, LmIn &x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { x.p[c] = d.pts[(size_t)c * d.Lpad + l]; x.n[c] = d.nrm[(size_t)c * d.Lpad + l]; }
    x.mat = d.lm_mat[l];
}

// ------------------------------------------------------------------ kernels ---
// One lane per landmark: C^-1 = (H_ll + D^2)^-1 for the current radius; read by the Schur producers
// and the back-substitution (a radius change alone re-runs this, not the linearisation).
__global__ __launch_bounds__(256) void k_ph_invert(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int l = blockIdx.x * 256 + threadIdx.x;
    double Ci[21];
#pragma unroll
    for (int c = 0; c < 21; ++c) Ci[c] = 0.0;
    if (d.lm_mask[l]) {
        double h[21], dmp[6];
#pragma unroll
        for (int c = 0; c < 21; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
        ph_damping(d, st, l, h, dmp);
        if (!inv6_spd(h, dmp, Ci)) {
            d.st->step_failed = 1;
#pragma unroll
            for (int c = 0; c < 21; ++c) Ci[c] = 0.0;
        }
    }
#pragma unroll
    for (int c = 0; c < 21; ++c) d.cinv[(size_t)c * d.Lpad + l] = Ci[c];
}

__global__ __launch_bounds__(256) void k_ph_linearize_landmarks(Dev d) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double cost = 0.0, xn = 0.0, gm = 0.0;
    if (mask) {
        const uint32_t win = d.lm_win[l];
        LmIn x;
        load_lm(d, l, x);
        const size_t obase = (size_t)(l >> 6) * (TW * LMG) + (l & 63);
        double h[21], g[6];
#pragma unroll
        for (int i = 0; i < 21; ++i) h[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = 0.0;
        for (int s = 0; s < TW; ++s) {
            if (!((mask >> s) & 1u)) continue;
            const uint32_t k = d.win_pose[win * TW + s];
            const size_t oi = obase + (size_t)s * LMG;
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            ObsPh o;
            obs_ph_linearize(d, d.poses + (size_t)k * 12, x.p, x.n, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, false, o);
            cost += o.half_sq;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    g[a] += o.Jl[6 * m + a] * o.r[m];
#pragma unroll
                    for (int b = a; b < 6; ++b) h[q++] += o.Jl[6 * m + a] * o.Jl[6 * m + b];
                }
            }
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
        xn = x.p[0] * x.p[0] + x.p[1] * x.p[1] + x.p[2] * x.p[2] + x.n[0] * x.n[0] + x.n[1] * x.n[1] + x.n[2] * x.n[2];
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
        d.part_lin[blockIdx.x * 4 + 0] = c0;
        d.part_lin[blockIdx.x * 4 + 1] = c1;
        d.part_lin[blockIdx.x * 4 + 2] = c2;
    }
}

__global__ __launch_bounds__(256) void k_ph_linearize_poses(Dev d) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const int k = blockIdx.x;
    if (d.pose_free[k] < 0) return;
    __shared__ double sm[4][27];
    const double *T = d.poses + (size_t)k * 12;
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    const uint32_t b = d.pose_obs_start[k], e = d.pose_obs_start[k + 1];
    for (uint32_t i = b + threadIdx.x; i < e; i += 256) {
        const uint32_t ref = d.pose_obs_ref[i];
        const int l = (int)(ref >> 4), s = (int)(ref & 15u);
        const size_t oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)s * LMG + (l & 63);
        LmIn x;
        load_lm(d, l, x);
        const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
        ObsPh o;
        obs_ph_linearize(d, T, x.p, x.n, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, true, o);
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            int n = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                acc[21 + a] += o.Jp[6 * m + a] * o.r[m];
#pragma unroll
                for (int c = a; c < 6; ++c) acc[n++] += o.Jp[6 * m + a] * o.Jp[6 * m + c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        const double v = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        const double v = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
        if (threadIdx.x < 21) d.hpp[(size_t)k * 21 + threadIdx.x] = v;
        else d.gp[(size_t)k * 6 + (threadIdx.x - 21)] = v;
    }
}

// Output-stationary Schur complement for 6-D landmark blocks: same structure as k_schur_windows with
// W, Y of size 6x6 per (landmark, slot) and K = 6 in the pair products.
constexpr int PH_THREADS = 256;
constexpr int PH_BATCH = 10;      // 120 producer lanes
constexpr int PH_SPLIT = 3;
constexpr int PH_STRIDE = 74;     // W(36) | Y(36) | pad(2): 592 B, 16-byte aligned
constexpr int PH_LDS_DOUBLES = PH_BATCH * TW * PH_STRIDE + PH_BATCH * 28;   // + per-landmark [Ci(21) | g_l(6) | pad]

__global__ __launch_bounds__(PH_THREADS, 2) void k_ph_schur_windows(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    extern __shared__ __align__(16) double ph_lds[];
    double *sWY = ph_lds;
    double *sLM = ph_lds + PH_BATCH * TW * PH_STRIDE;
    const int item = blockIdx.x;
    const uint32_t win = d.slab_win[item];
    const int lb = (int)d.slab_lm_begin[item], le = (int)d.slab_lm_end[item];
    const int t = threadIdx.x;
    const bool producer = t < PH_BATCH * TW;
    const int li = t / TW, s = t - li * TW;
    const bool consumer = t < NPAIR * PH_SPLIT;
    const int grp = t / NPAIR, pr = t - grp * NPAIR;
    const int pa = consumer ? c_ph_pair_a[pr] : 0, pb = consumer ? c_ph_pair_b[pr] : 0;
    double acc[36], racc[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) racc[i] = 0.0;
    bool pose_ok = false;
    uint32_t k = 0xFFFFFFFFu;
    if (producer) {
        k = d.win_pose[win * TW + s];
        pose_ok = (k != 0xFFFFFFFFu) && d.pose_free[k] >= 0;
    }
    for (int l0 = lb; l0 < le; l0 += PH_BATCH) {
        // phase 0: stage C^-1 (21) and g_l (6) of the batch
        for (int i = t; i < PH_BATCH * 27; i += PH_THREADS) {
            const int c = i / PH_BATCH, j = i - c * PH_BATCH, l = l0 + j;
            double v = 0.0;
            if (l < le) v = c < 21 ? d.cinv[(size_t)c * d.Lpad + l] : d.gl[(size_t)(c - 21) * d.Lpad + l];
            sLM[j * 28 + c] = v;
        }
        __syncthreads();
        // phase 1: W = J_p^T J_l and Y = W C^-1 per (landmark, slot)
        if (producer) {
            const int l = l0 + li;
            double *dst = sWY + (li * TW + s) * PH_STRIDE;
            bool live = false;
            if (l < le && pose_ok && ((d.lm_mask[l] >> s) & 1u)) {
                live = true;
                LmIn x;
                load_lm(d, l, x);
                const size_t oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)s * LMG + (l & 63);
                const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
                ObsPh o;
                obs_ph_linearize(d, d.poses + (size_t)k * 12, x.p, x.n, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, true, o);
                const double *Ci = sLM + li * 28;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double w[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int m = 0; m < 7; ++m) v += o.Jp[6 * m + a] * o.Jl[6 * m + c];
                        w[c] = v;
                        dst[6 * a + c] = v;
                    }
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < 6; ++q) v += w[q] * Ci[q <= c ? tri6(q, c) : tri6(c, q)];
                        dst[36 + 6 * a + c] = v;
                    }
                }
            }
            if (!live) {
#pragma unroll
                for (int i = 0; i < 72; ++i) dst[i] = 0.0;
            }
        }
        __syncthreads();
        // phase 2: pair products
        if (consumer) {
            const int nb = min(PH_BATCH, le - l0);
            for (int j = grp; j < nb; j += PH_SPLIT) {
                const double2 *Y2 = reinterpret_cast<const double2 *>(sWY + (j * TW + pa) * PH_STRIDE + 36);
                const double2 *W2 = reinterpret_cast<const double2 *>(sWY + (j * TW + pb) * PH_STRIDE);
                double y[36];
#pragma unroll
                for (int i = 0; i < 18; ++i) { const double2 v = Y2[i]; y[2 * i] = v.x; y[2 * i + 1] = v.y; }
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double w[6];
#pragma unroll
                    for (int i = 0; i < 3; ++i) { const double2 v = W2[3 * c + i]; w[2 * i] = v.x; w[2 * i + 1] = v.y; }
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        double v = acc[6 * a + c];
#pragma unroll
                        for (int q = 0; q < 6; ++q) v = fma(y[6 * a + q], w[q], v);
                        acc[6 * a + c] = v;
                    }
                }
                if (pa == pb) {
                    const double *g = sLM + j * 28 + 21;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        double v = racc[a];
#pragma unroll
                        for (int q = 0; q < 6; ++q) v = fma(y[6 * a + q], g[q], v);
                        racc[a] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    double *part = ph_lds;   // 2 x 78 x 42 doubles
    if (consumer && grp > 0) {
        double *o = part + ((grp - 1) * NPAIR + pr) * 42;
#pragma unroll
        for (int i = 0; i < 36; ++i) o[i] = acc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) o[36 + i] = racc[i];
    }
    __syncthreads();
    if (consumer && grp == 0) {
        const double *p1 = part + pr * 42, *p2 = part + (NPAIR + pr) * 42;
        double *out = d.slab + (size_t)item * SLAB_DOUBLES + (size_t)pr * 36;
#pragma unroll
        for (int i = 0; i < 36; ++i) out[i] = (acc[i] + p1[i]) + p2[i];
        if (pa == pb) {
            double *ro = d.slab + (size_t)item * SLAB_DOUBLES + NPAIR * 36 + pa * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) ro[a] = (racc[a] + p1[36 + a]) + p2[36 + a];
        }
    }
}

__global__ __launch_bounds__(256) void k_ph_backsub_eval(Dev d) {
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
        const uint32_t win = d.lm_win[l];
        const size_t obase = (size_t)(l >> 6) * (TW * LMG) + (l & 63);
        double tt[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) tt[c] = d.gl[(size_t)c * d.Lpad + l];
        for (int s = 0; s < TW; ++s) {
            if (!((mask >> s) & 1u)) continue;
            const uint32_t k = d.win_pose[win * TW + s];
            const int f = d.pose_free[k];
            if (f < 0) continue;
            const size_t oi = obase + (size_t)s * LMG;
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            ObsPh o;
            obs_ph_linearize(d, d.poses + (size_t)k * 12, x.p, x.n, x.mat, d.ou[oi], d.ov[oi], d.od[oi], d.oi[oi], nobs, true, o);
            const double *dp = d.x0 + (size_t)f * 6;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                double jd = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) jd += o.Jp[6 * m + c] * dp[c];
#pragma unroll
                for (int c = 0; c < 6; ++c) tt[c] += o.Jl[6 * m + c] * jd;
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
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (!isfinite(dl[a])) nonfinite = 1.0;
        np_[0] = x.p[0] + dl[0]; np_[1] = x.p[1] + dl[1]; np_[2] = x.p[2] + dl[2];
        unit_plus(x.n, dl + 3, nn);
        dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2] + (nn[0] - x.n[0]) * (nn[0] - x.n[0]) +
             (nn[1] - x.n[1]) * (nn[1] - x.n[1]) + (nn[2] - x.n[2]) * (nn[2] - x.n[2]);
        for (int s = 0; s < TW; ++s) {
            if (!((mask >> s) & 1u)) continue;
            const uint32_t k = d.win_pose[win * TW + s];
            const int f = d.pose_free[k];
            const size_t oi = obase + (size_t)s * LMG;
            const double nobs[3] = {d.onx[oi], d.ony[oi], d.onz[oi]};
            const double u = d.ou[oi], v = d.ov[oi], dd = d.od[oi], inten = d.oi[oi];
            ObsPh o;
            obs_ph_linearize(d, d.poses + (size_t)k * 12, x.p, x.n, x.mat, u, v, dd, inten, nobs, f >= 0, o);
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                double jd = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) jd += o.Jl[6 * m + c] * dl[c];
                if (f >= 0) {
                    const double *dp = d.x0 + (size_t)f * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) jd += o.Jp[6 * m + c] * dp[c];
                }
                mcc -= jd * (o.r[m] + 0.5 * jd);
            }
            ccost += obs_ph_cost(d, d.cand_poses + (size_t)k * 12, np_, nn, x.mat, u, v, dd, inten, nobs);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d.cand_pts[(size_t)c * d.Lpad + l] = np_[c];
        d.cand_nrm[(size_t)c * d.Lpad + l] = nn[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) d.dlm[(size_t)c * d.Lpad + l] = dl[c];
    const double a = block_sum(ccost, sm), b = block_sum(mcc, sm), c = block_sum(dn, sm), e = block_sum(nonfinite, sm);
    if (threadIdx.x == 0) {
        d.part_eval[blockIdx.x * 4 + 0] = a;
        d.part_eval[blockIdx.x * 4 + 1] = b;
        d.part_eval[blockIdx.x * 4 + 2] = c;
        d.part_eval[blockIdx.x * 4 + 3] = e;
    }
}

void launch_ph_linearize(Launcher &L, const Dev &d) {
    LAUNCH(KC_LIN_LM, k_ph_linearize_landmarks, dim3(d.n_lm_blocks), dim3(256), 0, d);
    LAUNCH(KC_LIN_POSE, k_ph_linearize_poses, dim3(d.P), dim3(256), 0, d);
}
void launch_ph_schur(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_ph_invert, dim3(d.n_lm_blocks), dim3(256), 0, d);
    LAUNCH(KC_SCHUR, k_ph_schur_windows, dim3(d.n_slabs), dim3(PH_THREADS), PH_LDS_DOUBLES * sizeof(double), d);
}
void launch_ph_backsub_eval(Launcher &L, const Dev &d) {
    LAUNCH(KC_BACKSUB_EVAL, k_ph_backsub_eval, dim3(d.n_lm_blocks), dim3(256), 0, d);
}
int configure_phong() {
    return hipFuncSetAttribute((const void *)k_ph_schur_windows, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(PH_LDS_DOUBLES * sizeof(double))) == hipSuccess ? 0 : -1;
}

}  // namespace ssba

// Phong-lighting device functions, shared by the batch evaluation kernel
// (ssba_phong.hip) and the config-3 solver kernels (ssba_phong_solver.hip).  Citations: see ssba_phong.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace ssba {

static __device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// hardware estimates + two Newton steps: full fp64 accuracy at a fraction of the IEEE sqrt / divide chains
static __device__ __forceinline__ double ph_rsqrt(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}
static __device__ __forceinline__ double ph_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}

struct PhongGrad { double nc[3], ell[3], cd[3], mat[3]; };   // mat = d/dkd, d/dks, d/dalpha

// clamped intensity for camera-frame normal nc, UNIT light direction ell, UNIT camera direction cd
static __device__ __forceinline__ double phong_core(const double nc[3], const double ell[3], const double cd[3], double kd,
                                             double ks, double alpha, PhongGrad *g) {
    double diffuse = 0.0, specular = 0.0;
    if (g) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g->nc[i] = g->ell[i] = g->cd[i] = g->mat[i] = 0.0;
    }
    const bool finite = isfinite(ell[0]) && isfinite(ell[1]) && isfinite(ell[2]);
    const double ldn = dot3(ell, nc);
    if (finite && !(ldn <= 0.0)) {            // phong.hpp:62-71
        diffuse = kd * ldn;
        if (g) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { g->ell[i] += kd * nc[i]; g->nc[i] += kd * ell[i]; }
            g->mat[0] = ldn;
        }
    }
    double mt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mt[i] = 2.0 * ldn * nc[i] - ell[i];   // phong.hpp:81-84
    const double mu2 = dot3(mt, mt);
    if (!(mu2 <= 0.0)) {                       // phong.hpp:88-90
        const double rmu = ph_rsqrt(mu2);      // reciprocal square roots + multiplications: IEEE sqrt / divide are 150 / 110 dependent cycles each
        const double m[3] = {mt[0] * rmu, mt[1] * rmu, mt[2] * rmu};
        const double s = dot3(m, cd);
        if (!(s <= 0.0)) {                     // phong.hpp:98-100
            // s^alpha, s^(alpha-1) and log s from ONE logarithm and one exponential (pow is both, twice over, at extended
            // precision: the three calls were a third of this function); |alpha| <= 20 keeps exp(alpha log s) within ~2e-15
            const double ls = log(s);
            const double sa = exp(alpha * ls);
            specular = ks * sa;
            if (g) {
                const double gs = ks * alpha * (sa * ph_rcp(s));
                double w[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) w[i] = (cd[i] - m[i] * s) * rmu;   // d s / d m~
                const double nw = dot3(nc, w);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    g->ell[i] += gs * (2.0 * nc[i] * nw - w[i]);
                    g->nc[i] += gs * 2.0 * (ldn * w[i] + ell[i] * nw);
                    g->cd[i] += gs * m[i];
                }
                g->mat[1] = sa;
                g->mat[2] = ks * sa * ls;
            }
        }
    }
    double col = 1.0 * (0.0 + diffuse + specular);   // ambient forced to 0 (phong.hpp:33)
    bool clamped = false;
    if (0.0 >= col) { col = 0.0; clamped = true; }   // fmax(Colour(0), col)  utils.hpp:16-19
    if (1.0 <= col) { col = 1.0; clamped = true; }   // fmin(Colour(1), col)  utils.hpp:22-25
    if (clamped && g) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g->nc[i] = g->ell[i] = g->cd[i] = g->mat[i] = 0.0;
    }
    return col;
}

// g^T (-a^)
static __device__ __forceinline__ void row_times_neg_skew(const double g[3], const double a[3], double out[3]) {
    out[0] = -g[1] * a[2] + g[2] * a[1];
    out[1] = g[0] * a[2] - g[2] * a[0];
    out[2] = -g[0] * a[1] + g[1] * a[0];
}

// plus-Jacobian of UnitVectorPerturbation at delta = 0: (I - x x^T/|x|^2)/|x|
static __device__ __forceinline__ void row_times_unit_plus(const double g[3], const double x[3], double out[3]) {
    const double n2 = dot3(x, x), inv = ph_rsqrt(n2), gx = dot3(g, x) * inv * inv;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j] = (g[j] - gx * x[j]) * inv;
}

static __device__ __forceinline__ void intensity_residual(int light_type, const double *__restrict__ T, const double p[3],
                                                   const double n[3], const double phong[3], double kd,
                                                   const double light[3], double colour, double stiffness,
                                                   double *r, double *J19) {
    const double *R = T + 3;
    double q[3], nc[3], lc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        q[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2] + T[i];
        nc[i] = R[3 * i] * n[0] + R[3 * i + 1] * n[1] + R[3 * i + 2] * n[2];
        lc[i] = R[3 * i] * light[0] + R[3 * i + 1] * light[1] + R[3 * i + 2] * light[2] + (light_type == 0 ? T[i] : 0.0);
    }
    double ell[3], cd[3], v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = light_type == 0 ? lc[i] - q[i] : lc[i];
    const double rrho = ph_rsqrt(dot3(v, v)), rqn = ph_rsqrt(dot3(q, q));
#pragma unroll
    for (int i = 0; i < 3; ++i) { ell[i] = v[i] * rrho; cd[i] = -q[i] * rqn; }
    PhongGrad g;
    const double col = phong_core(nc, ell, cd, kd, phong[1], phong[2], J19 ? &g : nullptr);
    *r = stiffness * (col - colour);
    if (!J19) return;
    const double le = dot3(ell, g.ell), ce = dot3(cd, g.cd);
    double g_l[3], g_q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double gv = (g.ell[i] - ell[i] * le) * rrho;
        const double gc = -(g.cd[i] - cd[i] * ce) * rqn;
        g_l[i] = gv;
        g_q[i] = gc - (light_type == 0 ? gv : 0.0);
    }
    double rq[3], rn[3], rl[3], t3[3];
    row_times_neg_skew(g_q, q, rq);
    row_times_neg_skew(g.nc, nc, rn);
    row_times_neg_skew(g_l, lc, rl);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J19[i] = stiffness * (g_q[i] + (light_type == 0 ? g_l[i] : 0.0));
        J19[3 + i] = stiffness * (rq[i] + rn[i] + rl[i]);
        J19[6 + i] = stiffness * (g_q[0] * R[i] + g_q[1] * R[3 + i] + g_q[2] * R[6 + i]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) t3[j] = g.nc[0] * R[j] + g.nc[1] * R[3 + j] + g.nc[2] * R[6 + j];
    double o3[3];
    row_times_unit_plus(t3, n, o3);
#pragma unroll
    for (int j = 0; j < 3; ++j) J19[9 + j] = stiffness * o3[j];
    J19[12] = 0.0;                       // d/d ka: ambient disabled (phong.hpp:33)
    J19[13] = stiffness * g.mat[1];
    J19[14] = stiffness * g.mat[2];
    J19[15] = stiffness * g.mat[0];
#pragma unroll
    for (int j = 0; j < 3; ++j) t3[j] = g_l[0] * R[j] + g_l[1] * R[3 + j] + g_l[2] * R[6 + j];
    if (light_type == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) J19[16 + j] = stiffness * t3[j];
    } else {
        row_times_unit_plus(t3, light, o3);
#pragma unroll
        for (int j = 0; j < 3; ++j) J19[16 + j] = stiffness * o3[j];
    }
}

static __device__ __forceinline__ void normal_residual(const double *__restrict__ T, const double n[3], const double nobs[3],
                                                const double S[9], double r[3], double *Jpose, double *Jn) {
    const double *R = T + 3;
    double nc[3], e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { nc[i] = R[3 * i] * n[0] + R[3 * i + 1] * n[1] + R[3 * i + 2] * n[2]; e[i] = nc[i] - nobs[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = S[3 * i] * e[0] + S[3 * i + 1] * e[1] + S[3 * i + 2] * e[2];
    if (!Jpose) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double rr[3], sr[3], o3[3];
        row_times_neg_skew(S + 3 * i, nc, rr);
        Jpose[6 * i] = Jpose[6 * i + 1] = Jpose[6 * i + 2] = 0.0;
        Jpose[6 * i + 3] = rr[0]; Jpose[6 * i + 4] = rr[1]; Jpose[6 * i + 5] = rr[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) sr[j] = S[3 * i] * R[j] + S[3 * i + 1] * R[3 + j] + S[3 * i + 2] * R[6 + j];
        row_times_unit_plus(sr, n, o3);
        Jn[3 * i] = o3[0]; Jn[3 * i + 1] = o3[1]; Jn[3 * i + 2] = o3[2];
    }
}


}  // namespace ssba

// Phong-lighting rows of the hot path as gfx950 device functions plus a
// batch evaluation kernel: intensity residual (point / directional light) with its 19 local
// Jacobian entries, normal residual with its 3x6 / 3x3 Jacobians, unit-vector Plus.
//
// Follows /root/reference include/ceres_slam/lighting/phong.hpp:25-51,59-104,136-139,
// lighting/point_light.hpp:76-90, lighting/directional_light.hpp:32-35,82-91,
// intensity_error_point_light.hpp:24-96, intensity_error_directional_light.hpp:24-96,
// normal_error.hpp:22-42, perturbations.hpp:87-103, utils/utils.hpp:16-25 -- with the autodiff
// Jets replaced by hand-derived gradients (same branch choices: where a guard or the [0,1] clamp
// fires the reference's Jet becomes a constant, so every derivative is zero there).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>

#include "../../include/ssba.h"

namespace ssba {

__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

struct PhongGrad { double nc[3], ell[3], cd[3], mat[3]; };   // mat = d/dkd, d/dks, d/dalpha

// clamped intensity for camera-frame normal nc, UNIT light direction ell, UNIT camera direction cd
__device__ __forceinline__ double phong_core(const double nc[3], const double ell[3], const double cd[3], double kd,
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
        const double mu = sqrt(mu2);
        const double m[3] = {mt[0] / mu, mt[1] / mu, mt[2] / mu};
        const double s = dot3(m, cd);
        if (!(s <= 0.0)) {                     // phong.hpp:98-100
            const double sa = pow(s, alpha);
            specular = ks * sa;
            if (g) {
                const double gs = ks * alpha * pow(s, alpha - 1.0);
                double w[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) w[i] = (cd[i] - m[i] * s) / mu;   // d s / d m~
                const double nw = dot3(nc, w);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    g->ell[i] += gs * (2.0 * nc[i] * nw - w[i]);
                    g->nc[i] += gs * 2.0 * (ldn * w[i] + ell[i] * nw);
                    g->cd[i] += gs * m[i];
                }
                g->mat[1] = sa;
                g->mat[2] = ks * sa * log(s);
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
__device__ __forceinline__ void row_times_neg_skew(const double g[3], const double a[3], double out[3]) {
    out[0] = -g[1] * a[2] + g[2] * a[1];
    out[1] = g[0] * a[2] - g[2] * a[0];
    out[2] = -g[0] * a[1] + g[1] * a[0];
}

// plus-Jacobian of UnitVectorPerturbation at delta = 0: (I - x x^T/|x|^2)/|x|
__device__ __forceinline__ void row_times_unit_plus(const double g[3], const double x[3], double out[3]) {
    const double n2 = dot3(x, x), inv = 1.0 / sqrt(n2), gx = dot3(g, x) / n2;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j] = (g[j] - gx * x[j]) * inv;
}

__device__ __forceinline__ void intensity_residual(int light_type, const double *__restrict__ T, const double p[3],
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
    const double rho = sqrt(dot3(v, v)), qn = sqrt(dot3(q, q));
#pragma unroll
    for (int i = 0; i < 3; ++i) { ell[i] = v[i] / rho; cd[i] = -q[i] / qn; }
    PhongGrad g;
    const double col = phong_core(nc, ell, cd, kd, phong[1], phong[2], J19 ? &g : nullptr);
    *r = stiffness * (col - colour);
    if (!J19) return;
    const double le = dot3(ell, g.ell), ce = dot3(cd, g.cd);
    double g_l[3], g_q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double gv = (g.ell[i] - ell[i] * le) / rho;
        const double gc = -(g.cd[i] - cd[i] * ce) / qn;
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

__device__ __forceinline__ void normal_residual(const double *__restrict__ T, const double n[3], const double nobs[3],
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

struct PhongBatch {
    int light_type;
    uint64_t n;
    const double *poses, *points, *normals, *phong, *texture, *colour, *nobs;
    double light[3], stiffness, Sn[9];
    double *r_int, *J_int, *r_nrm, *J_nrm_pose, *J_nrm_n;
};

__global__ __launch_bounds__(256) void k_phong_evaluate(PhongBatch b) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= b.n) return;
    double T[12], p[3], n[3], ph[3], no[3];
#pragma unroll
    for (int c = 0; c < 12; ++c) T[c] = b.poses[12 * i + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = b.points[3 * i + c]; n[c] = b.normals[3 * i + c]; ph[c] = b.phong[3 * i + c]; no[c] = b.nobs[3 * i + c]; }
    double r, J[19];
    intensity_residual(b.light_type, T, p, n, ph, b.texture[i], b.light, b.colour[i], b.stiffness, &r, J);
    b.r_int[i] = r;
#pragma unroll
    for (int c = 0; c < 19; ++c) b.J_int[19 * i + c] = J[c];
    double rn[3], Jp[18], Jn[9];
    normal_residual(T, n, no, b.Sn, rn, Jp, Jn);
#pragma unroll
    for (int c = 0; c < 3; ++c) b.r_nrm[3 * i + c] = rn[c];
#pragma unroll
    for (int c = 0; c < 18; ++c) b.J_nrm_pose[18 * i + c] = Jp[c];
#pragma unroll
    for (int c = 0; c < 9; ++c) b.J_nrm_n[9 * i + c] = Jn[c];
}

}  // namespace ssba

extern "C" int ssba_phong_evaluate(int device, int light_type, uint64_t n, const double *poses, const double *points,
                                   const double *normals, const double *phong, const double *texture,
                                   const double light[3], const double *colour, double stiffness,
                                   const double *normal_obs, const double normal_stiffness[9], double *r_int,
                                   double *J_int, double *r_nrm, double *J_nrm_pose, double *J_nrm_n) {
    using namespace ssba;
    if (!poses || !points || !normals || !phong || !texture || !light || !colour || !normal_obs || !normal_stiffness ||
        !r_int || !J_int || !r_nrm || !J_nrm_pose || !J_nrm_n || (light_type != 0 && light_type != 1))
        return SSBA_ERR_INVALID_ARGUMENT;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return SSBA_ERR_NO_DEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return SSBA_ERR_HIP;
    if (n == 0) return SSBA_OK;
    const size_t in_sizes[7] = {12, 3, 3, 3, 1, 1, 3};
    const double *in_host[7] = {poses, points, normals, phong, texture, colour, normal_obs};
    const size_t out_sizes[5] = {1, 19, 3, 18, 9};
    double *out_host[5] = {r_int, J_int, r_nrm, J_nrm_pose, J_nrm_n};
    double *din[7] = {nullptr}, *dout[5] = {nullptr};
    int rc = SSBA_OK;
    for (int k = 0; k < 7 && !rc; ++k) {
        if (hipMalloc((void **)&din[k], n * in_sizes[k] * sizeof(double)) != hipSuccess) rc = SSBA_ERR_HIP;
        else if (hipMemcpy(din[k], in_host[k], n * in_sizes[k] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = SSBA_ERR_HIP;
    }
    for (int k = 0; k < 5 && !rc; ++k)
        if (hipMalloc((void **)&dout[k], n * out_sizes[k] * sizeof(double)) != hipSuccess) rc = SSBA_ERR_HIP;
    if (!rc) {
        PhongBatch b;
        b.light_type = light_type; b.n = n;
        b.poses = din[0]; b.points = din[1]; b.normals = din[2]; b.phong = din[3]; b.texture = din[4]; b.colour = din[5]; b.nobs = din[6];
        for (int c = 0; c < 3; ++c) b.light[c] = light[c];
        b.stiffness = stiffness;
        for (int c = 0; c < 9; ++c) b.Sn[c] = normal_stiffness[c];
        b.r_int = dout[0]; b.J_int = dout[1]; b.r_nrm = dout[2]; b.J_nrm_pose = dout[3]; b.J_nrm_n = dout[4];
        hipLaunchKernelGGL(k_phong_evaluate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, b);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = SSBA_ERR_HIP;
    }
    for (int k = 0; k < 5 && !rc; ++k)
        if (hipMemcpy(out_host[k], dout[k], n * out_sizes[k] * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = SSBA_ERR_HIP;
    for (int k = 0; k < 7; ++k) if (din[k]) hipFree(din[k]);
    for (int k = 0; k < 5; ++k) if (dout[k]) hipFree(dout[k]);
    return rc;
}

// Unary pose residual blocks of the sun-aided driver (tests/dataset_vo_sun.cpp:80-124):
//   type 0  PoseErrorAutomatic      (include/ceres_slam/pose_error.hpp:22-55): r = S log(T_ref T^-1), the reference's
//           log = [translation ; SO3::log(rotation)] (se3group.hpp:337-342, so3group.hpp:293-348)
//   type 1  SunSensorErrorAutomatic (include/ceres_slam/sun_sensor_error.hpp:35-104): azimuth / zenith of R s_g against the
//           observed direction, wrap-around, outlier thresholds, 2x2 stiffness
// with closed-form local Jacobians (SE3Perturbation) and the Huber corrector of the stereo blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#include "ssba_types.h"

namespace ssba {

static __device__ void pf_so3_log(const double R[9], double phi[3]) {
    const double axis[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double sin_angle = 0.5 * sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const double cos_angle = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    const double angle = atan2(sin_angle, cos_angle);
    if (fabs(angle) <= DBL_EPSILON) { phi[0] = 0.5 * axis[0]; phi[1] = 0.5 * axis[1]; phi[2] = 0.5 * axis[2]; return; }
    for (int i = 0; i < 3; ++i) phi[i] = 0.5 * angle * axis[i] / sin_angle;
}

// inverse right Jacobian of SO(3): log(R Exp(d)) ~ log R + Jr^-1(phi) d
static __device__ void pf_inv_right_jacobian(const double phi[3], double J[9]) {
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2], th = sqrt(th2);
    const double W[9] = {0, -phi[2], phi[1], phi[2], 0, -phi[0], -phi[1], phi[0], 0};
    const double c = th < 1e-5 ? 1.0 / 12.0 + th2 / 720.0 : 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double w2 = 0.0;
            for (int k = 0; k < 3; ++k) w2 += W[3 * i + k] * W[3 * k + j];
            J[3 * i + j] = (i == j ? 1.0 : 0.0) + 0.5 * W[3 * i + j] + c * w2;
        }
}

static __device__ void pf_prior(const double *T, const double *T_ref, const double *S, double r[6], double *J) {
    const double *R = T + 3, *Rr = T_ref + 3;
    double Rres[9], e[6];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rres[3 * i + j] = Rr[3 * i] * R[3 * j] + Rr[3 * i + 1] * R[3 * j + 1] + Rr[3 * i + 2] * R[3 * j + 2];
    for (int i = 0; i < 3; ++i) e[i] = T_ref[i] - (Rres[3 * i] * T[0] + Rres[3 * i + 1] * T[1] + Rres[3 * i + 2] * T[2]);
    pf_so3_log(Rres, e + 3);
    for (int i = 0; i < 6; ++i) {
        double v = 0.0;
        for (int k = 0; k < 6; ++k) v += S[6 * i + k] * e[k];
        r[i] = v;
    }
    if (!J) return;
    double Jr[9];
    pf_inv_right_jacobian(e + 3, Jr);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0.0;     // S * [-R_res 0; 0 -Jr^-1]
            if (j < 3) for (int k = 0; k < 3; ++k) v -= S[6 * i + k] * Rres[3 * k + j];
            else for (int k = 0; k < 3; ++k) v -= S[6 * i + 3 + k] * Jr[3 * k + (j - 3)];
            J[6 * i + j] = v;
        }
}

// RelativePoseErrorAutomatic (include/ceres_slam/relative_pose_error.hpp:22-40): r = S log(T_ref T1 T2^-1).  With
// R_res = R_ref R1 R2^T, v = t1 - R1 R2^T t2, t_res = R_ref v + t_ref and T <- exp(eps) T on either pose:
//   d/d eps1 = [[R_ref, -R_ref v^], [0, Jl^-1(phi) R_ref]],  d/d eps2 = [[-R_res, 0], [0, -Jr^-1(phi)]],  Jl^-1 = (Jr^-1)^T
static __device__ void pf_rel(const double *T1, const double *T2, const double *T_ref, const double *S, double r[6], double *J1, double *J2) {
    const double *R1 = T1 + 3, *R2 = T2 + 3, *Rr = T_ref + 3;
    double R12[9], Rres[9], v[3], e[6];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R12[3 * i + j] = R1[3 * i] * R2[3 * j] + R1[3 * i + 1] * R2[3 * j + 1] + R1[3 * i + 2] * R2[3 * j + 2];
    for (int i = 0; i < 3; ++i) v[i] = T1[i] - (R12[3 * i] * T2[0] + R12[3 * i + 1] * T2[1] + R12[3 * i + 2] * T2[2]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rres[3 * i + j] = Rr[3 * i] * R12[j] + Rr[3 * i + 1] * R12[3 + j] + Rr[3 * i + 2] * R12[6 + j];
    for (int i = 0; i < 3; ++i) e[i] = Rr[3 * i] * v[0] + Rr[3 * i + 1] * v[1] + Rr[3 * i + 2] * v[2] + T_ref[i];
    pf_so3_log(Rres, e + 3);
    for (int i = 0; i < 6; ++i) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += S[6 * i + k] * e[k];
        r[i] = a;
    }
    if (!J1 && !J2) return;
    double Jr[9];
    pf_inv_right_jacobian(e + 3, Jr);
    const double vx[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 3; ++k) {
                if (j < 3) {
                    a += S[6 * i + k] * Rr[3 * k + j];
                    b -= S[6 * i + k] * Rres[3 * k + j];
                } else {
                    double rv = 0.0, jl = 0.0;
                    for (int q = 0; q < 3; ++q) { rv += Rr[3 * k + q] * vx[3 * q + (j - 3)]; jl += Jr[3 * q + k] * Rr[3 * q + (j - 3)]; }
                    a += -S[6 * i + k] * rv + S[6 * i + 3 + k] * jl;
                    b -= S[6 * i + 3 + k] * Jr[3 * k + (j - 3)];
                }
            }
            if (J1) J1[6 * i + j] = a;
            if (J2) J2[6 * i + j] = b;
        }
}

static __device__ void pf_sun(const double *T, const double *dat, const double *S, double r[2], double *J) {
    const double pi = 3.14159265358979323846;
    const double *R = T + 3;
    double oc[3], eg[3], sc[3];
    const double no = sqrt(dat[0] * dat[0] + dat[1] * dat[1] + dat[2] * dat[2]), ne = sqrt(dat[3] * dat[3] + dat[4] * dat[4] + dat[5] * dat[5]);
    for (int i = 0; i < 3; ++i) { oc[i] = dat[i] / no; eg[i] = dat[3 + i] / ne; }
    for (int i = 0; i < 3; ++i) sc[i] = R[3 * i] * eg[0] + R[3 * i + 1] * eg[1] + R[3 * i + 2] * eg[2];
    const double ezen = acos(-sc[1]), eaz = atan2(sc[0], sc[2]);
    const double ozen = acos(-oc[1]), oaz = atan2(oc[0], oc[2]);
    double raz = eaz - oaz, rzen = ezen - ozen;
    if (raz > pi) raz -= 2 * pi; else if (raz < -pi) raz += 2 * pi;
    bool kaz = true, kzen = true;
    if (fabs(raz) > dat[6]) { raz = 0.0; kaz = false; }
    if (fabs(rzen) > dat[7]) { rzen = 0.0; kzen = false; }
    r[0] = S[0] * raz + S[1] * rzen;
    r[1] = S[2] * raz + S[3] * rzen;
    if (!J) return;
    const double x = sc[0], y = sc[1], z = sc[2], d2 = x * x + z * z;
    const double gaz[3] = {z / d2, 0.0, -x / d2}, gzen[3] = {0.0, 1.0 / sqrt(1.0 - y * y), 0.0};
    // g^T (-s_c^)
    double jaz[3] = {-gaz[1] * sc[2] + gaz[2] * sc[1], gaz[0] * sc[2] - gaz[2] * sc[0], -gaz[0] * sc[1] + gaz[1] * sc[0]};
    double jzen[3] = {-gzen[1] * sc[2] + gzen[2] * sc[1], gzen[0] * sc[2] - gzen[2] * sc[0], -gzen[0] * sc[1] + gzen[1] * sc[0]};
    for (int c = 0; c < 3; ++c) { if (!kaz) jaz[c] = 0.0; if (!kzen) jzen[c] = 0.0; }
    for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 6; ++c) J[6 * i + c] = c < 3 ? 0.0 : S[2 * i] * jaz[c - 3] + S[2 * i + 1] * jzen[c - 3];
}

// the other pose of a relative-pose half entry (types 2 / 3), -1 for the unary blocks
static __device__ __forceinline__ int pf_other_pose(const Dev &d, int f) { return d.pf_type[f] >= 2 ? (int)d.pf_data[18 * (size_t)f + 12] : -1; }
static __device__ __forceinline__ int pf_partner(const Dev &d, int f) { return d.pf_type[f] >= 2 ? (int)d.pf_data[18 * (size_t)f + 14] : -1; }
static __device__ __forceinline__ bool pf_counts_cost(const Dev &d, int f) { return d.pf_type[f] < 2 || d.pf_data[18 * (size_t)f + 13] != 0.0; }

// factor f at pose block T (and, for the two halves of a relative-pose block, the other pose's block T_other): corrected
// residual r (dim rows) and the Jacobian J w.r.t. THIS pose (dim x 6, or NULL); returns 1/2 rho(|r|^2)
static __device__ double pf_evaluate(const Dev &d, int f, const double *T, const double *T_other, double r[6], double *J, int *dim_out) {
    const int type = d.pf_type[f], dim = type == 1 ? 2 : 6;
    for (int i = 0; i < 6; ++i) r[i] = 0.0;
    if (type == 0) pf_prior(T, d.pf_data + 18 * (size_t)f, d.pf_S + 36 * (size_t)f, r, J);
    else if (type == 1) pf_sun(T, d.pf_data + 18 * (size_t)f, d.pf_S + 36 * (size_t)f, r, J);
    else if (type == 2) pf_rel(T, T_other, d.pf_data + 18 * (size_t)f, d.pf_S + 36 * (size_t)f, r, J, nullptr);
    else pf_rel(T_other, T, d.pf_data + 18 * (size_t)f, d.pf_S + 36 * (size_t)f, r, nullptr, J);
    if (J) for (int i = 6 * dim; i < 36; ++i) J[i] = 0.0;
    double sq = 0.0;
    for (int i = 0; i < dim; ++i) sq += r[i] * r[i];
    *dim_out = dim;
    const double a = d.pf_huber[f];
    if (a > 0.0 && sq > a * a) {      // HuberLoss + corrector (rho'' <= 0: scale r and J by sqrt(rho'))
        const double rs = sqrt(sq), sc = sqrt(fmax(DBL_MIN, a / rs));
        for (int i = 0; i < dim; ++i) r[i] *= sc;
        if (J) for (int i = 0; i < 6 * dim; ++i) J[i] *= sc;
        return 0.5 * (2.0 * a * rs - a * a);
    }
    return 0.5 * sq;
}

}  // namespace ssba

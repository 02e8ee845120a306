/*
 * ssba_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ssba_oracle.h).
 *
 * CPU restatement of the ceres-slam stereo bundle-adjustment hot path.
 * PARITY UNPINNED at the Ceres boundary (header comment of ssba_oracle.h).
 * File:line citations are into /root/reference.
 */
#include "ssba_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------ */
/* L1 math                                                                    */
/* ------------------------------------------------------------------------ */

/* include/ceres_slam/geometry/se3group.hpp:191-193  p' = R p + t, block layout
 * [t | R row-major] (se3group.hpp:425-429, so3group.hpp:34). */
void orc_se3_transform(const double T[12], const double p[3], double out[3]) {
    const double *t = T, *R = T + 3;
    out[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    out[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    out[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
}

/* include/ceres_slam/geometry/so3group.hpp:273-291 (wedge :248-254).
 * angle <= DBL_EPSILON -> I + phi^ ; else cos*I + (1-cos) a a^T + sin * a^ */
void orc_so3_exp(const double phi[3], double R[9]) {
    double angle = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    if (angle <= DBL_EPSILON) {
        R[0] = 1.0;     R[1] = -phi[2]; R[2] = phi[1];
        R[3] = phi[2];  R[4] = 1.0;     R[5] = -phi[0];
        R[6] = -phi[1]; R[7] = phi[0];  R[8] = 1.0;
        return;
    }
    double a[3] = {phi[0] / angle, phi[1] / angle, phi[2] / angle};
    double cp = cos(angle), sp = sin(angle), omc = 1.0 - cp;
    R[0] = cp + omc * a[0] * a[0];
    R[1] = omc * a[0] * a[1] - sp * a[2];
    R[2] = omc * a[0] * a[2] + sp * a[1];
    R[3] = omc * a[1] * a[0] + sp * a[2];
    R[4] = cp + omc * a[1] * a[1];
    R[5] = omc * a[1] * a[2] - sp * a[0];
    R[6] = omc * a[2] * a[0] - sp * a[1];
    R[7] = omc * a[2] * a[1] + sp * a[0];
    R[8] = cp + omc * a[2] * a[2];
}

/* include/ceres_slam/perturbations.hpp:45-65: T_new = exp(eps) * T with
 * exp(xi) = (xi[0:3], Exp_SO3(xi[3:6])) (se3group.hpp:323-325, NOT the true SE(3)
 * exponential) and the product of se3group.hpp:176-183:
 *   R_new = E R,  t_new = E t + rho. */
void orc_se3_plus(const double T[12], const double eps[6], double out[12]) {
    double E[9];
    orc_so3_exp(eps + 3, E);
    const double *t = T, *R = T + 3;
    double tn[3], Rn[9];
    for (int i = 0; i < 3; ++i) {
        tn[i] = E[3 * i] * t[0] + E[3 * i + 1] * t[1] + E[3 * i + 2] * t[2] + eps[i];
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = E[3 * i] * R[j] + E[3 * i + 1] * R[3 + j] + E[3 * i + 2] * R[6 + j];
    }
    memcpy(out, tn, sizeof tn);
    memcpy(out + 3, Rn, sizeof Rn);
}

/* se3group.hpp:152-158: R^-1 = R^T (so3group.hpp inverse = transpose), t' = -(R^T t) */
void orc_se3_inverse(const double T[12], double out[12]) {
    const double *t = T, *R = T + 3;
    double o[12];
    for (int i = 0; i < 3; ++i) {
        o[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
        for (int j = 0; j < 3; ++j) o[3 + 3 * i + j] = R[3 * j + i];
    }
    memcpy(out, o, sizeof o);
}

/* include/ceres_slam/stereo_camera.hpp:77-108 */
void orc_project(const orc_camera *c, const double q[3], double uvd[3], double J[9]) {
    double one_over_z = 1.0 / q[2];
    uvd[0] = c->fu * q[0] * one_over_z + c->cu;
    uvd[1] = c->fv * q[1] * one_over_z + c->cv;
    uvd[2] = c->fu * c->b * one_over_z;
    if (J) {
        double one_over_z2 = one_over_z * one_over_z;
        J[0] = c->fu * one_over_z; J[1] = 0.0;                J[2] = -c->fu * q[0] * one_over_z2;
        J[3] = 0.0;                J[4] = c->fv * one_over_z; J[5] = -c->fv * q[1] * one_over_z2;
        J[6] = 0.0;                J[7] = 0.0;                J[8] = -c->fu * c->b * one_over_z2;
    }
}

/* include/ceres_slam/stereo_camera.hpp:112-144 */
void orc_triangulate(const orc_camera *c, const double uvd[3], double q[3], double J[9]) {
    double b_over_d = c->b / uvd[2];
    double fu_over_fv = c->fu / c->fv;
    q[0] = (uvd[0] - c->cu) * b_over_d;
    q[1] = (uvd[1] - c->cv) * b_over_d * fu_over_fv;
    q[2] = c->fu * b_over_d;
    if (J) {
        double b_over_d2 = b_over_d / uvd[2];
        J[0] = b_over_d; J[1] = 0.0;                   J[2] = (c->cu - uvd[0]) * b_over_d2;
        J[3] = 0.0;      J[4] = b_over_d * fu_over_fv; J[5] = (c->cv - uvd[1]) * b_over_d2 * fu_over_fv;
        J[6] = 0.0;      J[7] = 0.0;                   J[8] = -c->fu * b_over_d2;
    }
}

/* include/ceres_slam/stereo_reprojection_error.hpp:27-55:
 *     r = S * (project(T_c_g * pt_g) - z).
 * Local Jacobians (what AutoDiffCostFunction<...,3,12,3> x
 * AutoDiffLocalParameterization<SE3Perturbation,12,6> produce at eps = 0, where
 * so3group.hpp:277-280 takes the first-order branch):
 *     q(eps) = (I + phi^)(R p + t) + rho  =>  dq/deps = [ I | -q^ ],  dq/dp = R
 *     Jp = S Jpi [ I | -q^ ]  (3x6),   Jl = S Jpi R  (3x3). */
void orc_stereo_residual(const orc_camera *c, const double T[12], const double p[3],
                         const double z[3], const double S[9], double r[3],
                         double *Jp, double *Jl) {
    double q[3], pred[3], Jpi[9];
    orc_se3_transform(T, p, q);
    orc_project(c, q, pred, (Jp || Jl) ? Jpi : NULL);
    double e[3] = {pred[0] - z[0], pred[1] - z[1], pred[2] - z[2]};
    for (int i = 0; i < 3; ++i) r[i] = S[3 * i] * e[0] + S[3 * i + 1] * e[1] + S[3 * i + 2] * e[2];
    if (!(Jp || Jl)) return;
    double A[9]; /* S * Jpi */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = S[3 * i] * Jpi[j] + S[3 * i + 1] * Jpi[3 + j] + S[3 * i + 2] * Jpi[6 + j];
    if (Jp) {
        /* -q^ = [[0, q2, -q1], [-q2, 0, q0], [q1, -q0, 0]] */
        for (int i = 0; i < 3; ++i) {
            const double *a = A + 3 * i;
            Jp[6 * i + 0] = a[0];
            Jp[6 * i + 1] = a[1];
            Jp[6 * i + 2] = a[2];
            Jp[6 * i + 3] = -a[1] * q[2] + a[2] * q[1];
            Jp[6 * i + 4] = a[0] * q[2] - a[2] * q[0];
            Jp[6 * i + 5] = -a[0] * q[1] + a[1] * q[0];
        }
    }
    if (Jl) {
        const double *R = T + 3;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Jl[3 * i + j] = A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j];
    }
}

/* ---- the same block the way the reference pays for it: forward-mode automatic differentiation ------------------
 * ceres::AutoDiffCostFunction<StereoReprojectionErrorAutomatic, 3, 12, 3> (stereo_reprojection_error.hpp:59-69) runs
 * the functor (:27-55) on Jets with 12 + 3 = 15 derivative lanes, which yields the AMBIENT Jacobians (3x12 w.r.t. the
 * stored [t | R] block, 3x3 w.r.t. the point); Ceres then multiplies the pose part by the 12x6 Jacobian of
 * AutoDiffLocalParameterization<SE3Perturbation, 12, 6> (perturbations.hpp:69-75) at eps = 0.  Restated here with a
 * plain struct of 1 + 15 doubles so that (i) the CPU baseline can be timed in the reference's evaluation mode and
 * (ii) the closed forms above are cross-checked against a mechanical derivation (tests/test_oracle_math.py). */
typedef struct { double v, d[15]; } jet15;
static inline jet15 jet_var(double v, int k) { jet15 a; a.v = v; for (int i = 0; i < 15; ++i) a.d[i] = 0.0; a.d[k] = 1.0; return a; }
static inline jet15 jet_add(jet15 a, jet15 b) { for (int i = 0; i < 15; ++i) a.d[i] += b.d[i]; a.v += b.v; return a; }
static inline jet15 jet_mul(jet15 a, jet15 b) { jet15 c; c.v = a.v * b.v; for (int i = 0; i < 15; ++i) c.d[i] = a.v * b.d[i] + a.d[i] * b.v; return c; }
static inline jet15 jet_div(jet15 a, jet15 b) {      /* jet.h operator/: a/b with derivative (a' - (a/b) b') / b */
    jet15 c; const double inv = 1.0 / b.v; c.v = a.v * inv;
    for (int i = 0; i < 15; ++i) c.d[i] = (a.d[i] - c.v * b.d[i]) * inv;
    return c;
}
static inline jet15 jet_scale(jet15 a, double s) { a.v *= s; for (int i = 0; i < 15; ++i) a.d[i] *= s; return a; }
static inline jet15 jet_shift(jet15 a, double s) { a.v += s; return a; }

/* d Plus(T, eps) / d eps at eps = 0 for the [t | R row-major] block: exp(eps) T with the reference's first-order branch
 * (so3group.hpp:277-280, se3group.hpp:323-325): t' = (I + phi^) t + rho, R' = (I + phi^) R  =>  rows of -t^ / -(R col)^. */
static void se3_plus_jacobian(const double T[12], double P[72]) {
    memset(P, 0, 72 * sizeof(double));
    for (int i = 0; i < 3; ++i) P[6 * i + i] = 1.0;
    const double *t = T, *R = T + 3;
    /* (phi x a)_i = sum_k (-a^)_{ik} phi_k with -a^ = [[0, a2, -a1], [-a2, 0, a0], [a1, -a0, 0]] */
    double a[3];
    for (int blk = 0; blk < 4; ++blk) {
        if (blk == 0) { a[0] = t[0]; a[1] = t[1]; a[2] = t[2]; }
        else { a[0] = R[blk - 1]; a[1] = R[3 + blk - 1]; a[2] = R[6 + blk - 1]; }     /* column blk-1 of R */
        const double na[9] = {0, a[2], -a[1], -a[2], 0, a[0], a[1], -a[0], 0};
        for (int i = 0; i < 3; ++i) {
            const int row = blk == 0 ? i : 3 + 3 * i + (blk - 1);       /* t_i, or R_{i, blk-1} in row-major storage */
            for (int k = 0; k < 3; ++k) P[6 * row + 3 + k] = na[3 * i + k];
        }
    }
}

void orc_stereo_residual_autodiff(const orc_camera *c, const double T[12], const double p[3], const double z[3],
                                  const double S[9], double r[3], double *Jp, double *Jl) {
    jet15 Tj[12], pj[3], q[3], pred[3];
    for (int i = 0; i < 12; ++i) Tj[i] = jet_var(T[i], i);
    for (int i = 0; i < 3; ++i) pj[i] = jet_var(p[i], 12 + i);
    for (int i = 0; i < 3; ++i)       /* SE3Group<T>::transform (se3group.hpp:191-193): R p + t */
        q[i] = jet_add(jet_add(jet_add(jet_mul(Tj[3 + 3 * i], pj[0]), jet_mul(Tj[3 + 3 * i + 1], pj[1])), jet_mul(Tj[3 + 3 * i + 2], pj[2])), Tj[i]);
    /* StereoCamera::project (stereo_camera.hpp:77-84) */
    pred[0] = jet_shift(jet_scale(jet_div(q[0], q[2]), c->fu), c->cu);
    pred[1] = jet_shift(jet_scale(jet_div(q[1], q[2]), c->fv), c->cv);
    { jet15 one = q[2]; one.v = 1.0; for (int i = 0; i < 15; ++i) one.d[i] = 0.0; pred[2] = jet_scale(jet_div(one, q[2]), c->fu * c->b); }
    jet15 e[3], rr[3];
    for (int i = 0; i < 3; ++i) e[i] = jet_shift(pred[i], -z[i]);
    for (int i = 0; i < 3; ++i) rr[i] = jet_add(jet_add(jet_scale(e[0], S[3 * i]), jet_scale(e[1], S[3 * i + 1])), jet_scale(e[2], S[3 * i + 2]));
    for (int i = 0; i < 3; ++i) r[i] = rr[i].v;
    if (Jp) {
        double P[72];
        se3_plus_jacobian(T, P);
        for (int m = 0; m < 3; ++m)
            for (int col = 0; col < 6; ++col) {
                double v = 0.0;
                for (int k = 0; k < 12; ++k) v += rr[m].d[k] * P[6 * k + col];
                Jp[6 * m + col] = v;
            }
    }
    if (Jl)
        for (int m = 0; m < 3; ++m)
            for (int col = 0; col < 3; ++col) Jl[3 * m + col] = rr[m].d[12 + col];
}

/* 0: closed-form Jacobians (default), 1: the Jet restatement above for every stereo block (timing variant) */
static int g_jacobian_mode = 0;
void orc_set_jacobian_mode(int mode) { g_jacobian_mode = mode; }

/* [Ceres 1.x loss_function.cc, HuberLoss::Evaluate; call-site shape
 * tests/dataset_vo_sun.cpp:89-95]  s = |r|^2, b = a^2 */
void orc_huber(double a, double s, double rho[3]) {
    double b = a * a;
    if (s > b) {
        double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = fmax(DBL_MIN, a / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

/* [Ceres 1.x corrector.cc] rescale r and J so that Gauss-Newton on the corrected
 * system is the Triggs correction of the robustified cost. */
static void corrector(const double rho[3], double sq_norm, double r[3], double *Jp, double *Jl) {
    double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
        residual_scaling = sqrt_rho1;
        alpha_sq_norm = 0.0;
    } else {
        double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        double alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1.0 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    /* Jacobians first (they use the uncorrected residual), then the residual */
    double *Js[2] = {Jp, Jl};
    int cols[2] = {6, 3};
    for (int m = 0; m < 2; ++m) {
        double *J = Js[m];
        if (!J) continue;
        int nc = cols[m];
        if (alpha_sq_norm == 0.0) {
            for (int i = 0; i < 3 * nc; ++i) J[i] *= sqrt_rho1;
        } else {
            for (int c = 0; c < nc; ++c) {
                double rtj = r[0] * J[c] + r[1] * J[nc + c] + r[2] * J[2 * nc + c];
                for (int i = 0; i < 3; ++i)
                    J[i * nc + c] = sqrt_rho1 * (J[i * nc + c] - alpha_sq_norm * r[i] * rtj);
            }
        }
    }
    for (int i = 0; i < 3; ++i) r[i] *= residual_scaling;
}

void orc_default_options(orc_options *o) {
    /* Ceres 1.x Solver::Options defaults; the reference drivers override
     * max_num_iterations = 1000, use_nonmonotonic_steps = true, num_threads = 8
     * (tests/dataset_vo.cpp:65-70). */
    o->max_num_iterations = 50;
    o->use_nonmonotonic_steps = 0;
    o->max_consecutive_nonmonotonic_steps = 5;
    o->jacobi_scaling = 1;
    o->num_threads = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->trust_region_strategy_type = 0;
    o->dogleg_type = 0;
}

/* ------------------------------------------------------------------------ */
/* evaluation                                                                 */
/* ------------------------------------------------------------------------ */
/* Two problem shapes share the code below:
 *   stereo only (tests/dataset_vo.cpp):            NR = 3 residuals per observation, LD = 3
 *   stereo + Phong lighting (dataset_ba_phong.cpp): NR = 7 (stereo 3 | intensity 1 | normal 3),
 *     landmark block LD = 6 = [position | normal], the normal through UnitVectorPerturbation;
 *     the shared light / material / texture blocks (dataset_ba_phong.cpp:108-139) are free
 *     when `shared_free` says so: they form a dense "border" of local size
 *     nb = 3 [light] + 3 M [ka, ks, alpha per material] + M [kd per material].
 * Residual-block order inside an observation follows the driver: stereo, intensity, normal. */

/* packed ambient state of the shared blocks: [light 3 | phong 3M | texture M] */
static int shared_size(const orc_problem *p) { return 3 + 4 * (int)p->num_materials; }
static void shared_pack(const orc_problem *p, double *sh) {
    const int M = (int)p->num_materials;
    memcpy(sh, p->light, 3 * sizeof(double));
    memcpy(sh + 3, p->phong, (size_t)3 * M * sizeof(double));
    memcpy(sh + 3 + 3 * M, p->texture, (size_t)M * sizeof(double));
}

static int is_phong(const orc_problem *p) { return p->intensity != NULL; }
static int dim_nr(const orc_problem *p) { return is_phong(p) ? 7 : 3; }
static int dim_ld(const orc_problem *p) { return is_phong(p) ? 6 : 3; }

static void set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static double dot3(const double a[3], const double b[3]);
static void row_times_neg_skew(const double g[3], const double a[3], double out[3]);

/* ---- unary pose residual blocks ------------------------------------------
 * type 0: PoseErrorAutomatic (include/ceres_slam/pose_error.hpp:22-55): r = S log(T_ref T^-1) with the
 *         reference's log = [translation ; SO3::log(rotation)] (se3group.hpp:337-342, so3group.hpp:293-348);
 * type 1: SunSensorErrorAutomatic (include/ceres_slam/sun_sensor_error.hpp:35-104): azimuth / zenith of the
 *         expected sun direction R s_g against the observed one, wrap-around, outlier thresholds, 2x2 stiffness.
 * Local Jacobians (6 columns, SE3Perturbation) in closed form; Huber via the same corrector as the stereo blocks. */
static void so3_log(const double R[9], double phi[3]) {       /* so3group.hpp:293-348 */
    double axis[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double sin_angle = 0.5 * sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const double cos_angle = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    const double angle = atan2(sin_angle, cos_angle);
    if (fabs(angle) <= DBL_EPSILON) {                         /* vee(C - I) */
        phi[0] = 0.5 * (R[7] - R[5]); phi[1] = 0.5 * (R[2] - R[6]); phi[2] = 0.5 * (R[3] - R[1]);
        return;
    }
    for (int i = 0; i < 3; ++i) phi[i] = 0.5 * angle * axis[i] / sin_angle;
}

/* inverse right Jacobian of SO(3): log(R Exp(d)) ~ log R + Jr^-1(phi) d */
static void so3_inv_right_jacobian(const double phi[3], double J[9]) {
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2], th = sqrt(th2);
    const double W[9] = {0, -phi[2], phi[1], phi[2], 0, -phi[0], -phi[1], phi[0], 0};
    double c;
    if (th < 1e-5) c = 1.0 / 12.0 + th2 / 720.0;
    else c = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double w2 = 0.0;
            for (int k = 0; k < 3; ++k) w2 += W[3 * i + k] * W[3 * k + j];
            J[3 * i + j] = (i == j ? 1.0 : 0.0) + 0.5 * W[3 * i + j] + c * w2;
        }
}

void orc_pose_prior_residual(const double T[12], const double T_ref[12], const double S[36], double r[6], double *J) {
    const double *R = T + 3, *Rr = T_ref + 3;
    double Rres[9], e[6], Je[36];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rres[3 * i + j] = Rr[3 * i] * R[3 * j] + Rr[3 * i + 1] * R[3 * j + 1] + Rr[3 * i + 2] * R[3 * j + 2];   /* R_ref R^T */
    for (int i = 0; i < 3; ++i) e[i] = T_ref[i] - (Rres[3 * i] * T[0] + Rres[3 * i + 1] * T[1] + Rres[3 * i + 2] * T[2]);
    so3_log(Rres, e + 3);
    for (int i = 0; i < 6; ++i) {
        double v = 0.0;
        for (int k = 0; k < 6; ++k) v += S[6 * i + k] * e[k];
        r[i] = v;
    }
    if (!J) return;
    /* T <- exp(eps) T:  R_res <- R_res Exp(-phi),  t_res <- t_res - R_res Exp(-phi) rho  */
    double Jr[9];
    so3_inv_right_jacobian(e + 3, Jr);
    memset(Je, 0, sizeof Je);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Je[6 * i + j] = -Rres[3 * i + j]; Je[6 * (3 + i) + 3 + j] = -Jr[3 * i + j]; }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0.0;
            for (int k = 0; k < 6; ++k) v += S[6 * i + k] * Je[6 * k + j];
            J[6 * i + j] = v;
        }
}

/* include/ceres_slam/relative_pose_error.hpp:22-40: r = S log(T_2_1_ref * T_1_0 * T_2_0^-1), log = [t ; axis-angle].
 * With R_res = R_ref R1 R2^T, v = t1 - R1 R2^T t2, t_res = R_ref v + t_ref and the local perturbations T <- exp(eps) T:
 *   d/d eps1 = [[R_ref, -R_ref v^], [0, Jl^-1(phi) R_ref]],  d/d eps2 = [[-R_res, 0], [0, -Jr^-1(phi)]],  Jl^-1 = (Jr^-1)^T */
void orc_relative_pose_residual(const double T1[12], const double T2[12], const double T_ref[12], const double S[36], double r[6],
                                double *J1, double *J2) {
    const double *R1 = T1 + 3, *R2 = T2 + 3, *Rr = T_ref + 3;
    double R12[9], Rres[9], v[3], e[6];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R12[3 * i + j] = R1[3 * i] * R2[3 * j] + R1[3 * i + 1] * R2[3 * j + 1] + R1[3 * i + 2] * R2[3 * j + 2];   /* R1 R2^T */
    for (int i = 0; i < 3; ++i) v[i] = T1[i] - (R12[3 * i] * T2[0] + R12[3 * i + 1] * T2[1] + R12[3 * i + 2] * T2[2]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rres[3 * i + j] = Rr[3 * i] * R12[j] + Rr[3 * i + 1] * R12[3 + j] + Rr[3 * i + 2] * R12[6 + j];
    for (int i = 0; i < 3; ++i) e[i] = Rr[3 * i] * v[0] + Rr[3 * i + 1] * v[1] + Rr[3 * i + 2] * v[2] + T_ref[i];
    so3_log(Rres, e + 3);
    for (int i = 0; i < 6; ++i) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += S[6 * i + k] * e[k];
        r[i] = a;
    }
    if (!J1 && !J2) return;
    double Jr[9], Je1[36], Je2[36];
    so3_inv_right_jacobian(e + 3, Jr);
    memset(Je1, 0, sizeof Je1);
    memset(Je2, 0, sizeof Je2);
    const double vx[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double rv = 0.0, jl = 0.0;
            for (int k = 0; k < 3; ++k) { rv += Rr[3 * i + k] * vx[3 * k + j]; jl += Jr[3 * k + i] * Rr[3 * k + j]; }
            Je1[6 * i + j] = Rr[3 * i + j];
            Je1[6 * i + 3 + j] = -rv;
            Je1[6 * (3 + i) + 3 + j] = jl;                 /* (Jr^-1)^T R_ref */
            Je2[6 * i + j] = -Rres[3 * i + j];
            Je2[6 * (3 + i) + 3 + j] = -Jr[3 * i + j];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 6; ++k) { a += S[6 * i + k] * Je1[6 * k + j]; b += S[6 * i + k] * Je2[6 * k + j]; }
            if (J1) J1[6 * i + j] = a;
            if (J2) J2[6 * i + j] = b;
        }
}

void orc_sun_residual(const double T[12], const double obs_c_in[3], const double exp_g_in[3], const double S[4], double az_thresh,
                      double zen_thresh, double r[2], double *J) {
    const double pi = 3.14159265358979323846;
    const double *R = T + 3;
    double oc[3], eg[3], sc[3];
    const double no = sqrt(dot3(obs_c_in, obs_c_in)), ne = sqrt(dot3(exp_g_in, exp_g_in));     /* normalised in the constructor (:30-31) */
    for (int i = 0; i < 3; ++i) { oc[i] = obs_c_in[i] / no; eg[i] = exp_g_in[i] / ne; }
    for (int i = 0; i < 3; ++i) sc[i] = R[3 * i] * eg[0] + R[3 * i + 1] * eg[1] + R[3 * i + 2] * eg[2];
    const double ezen = acos(-sc[1]), eaz = atan2(sc[0], sc[2]);
    const double ozen = acos(-oc[1]), oaz = atan2(oc[0], oc[2]);
    double raz = eaz - oaz, rzen = ezen - ozen;
    if (raz > pi) raz -= 2 * pi; else if (raz < -pi) raz += 2 * pi;
    int kaz = 1, kzen = 1;
    if (fabs(raz) > az_thresh) { raz = 0.0; kaz = 0; }
    if (fabs(rzen) > zen_thresh) { rzen = 0.0; kzen = 0; }
    r[0] = S[0] * raz + S[1] * rzen;
    r[1] = S[2] * raz + S[3] * rzen;
    if (!J) return;
    /* d s_c / d phi = -s_c^ ;  d az / d s_c = [z, 0, -x] / (x^2 + z^2) ;  d zen / d s_c = [0, 1/sqrt(1-y^2), 0] */
    const double x = sc[0], y = sc[1], z = sc[2], d2 = x * x + z * z;
    const double gaz[3] = {z / d2, 0.0, -x / d2}, gzen[3] = {0.0, 1.0 / sqrt(1.0 - y * y), 0.0};
    double jaz[3], jzen[3];
    row_times_neg_skew(gaz, sc, jaz);
    row_times_neg_skew(gzen, sc, jzen);
    for (int c = 0; c < 3; ++c) { if (!kaz) jaz[c] = 0.0; if (!kzen) jzen[c] = 0.0; }
    for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 6; ++c) J[6 * i + c] = c < 3 ? 0.0 : S[2 * i] * jaz[c - 3] + S[2 * i + 1] * jzen[c - 3];
}

/* all unary factors at `poses`: corrected residuals pf_r (6 per factor, unused rows zero) and Jacobians pf_J (36 per
 * factor) if requested; returns their cost 1/2 sum rho(|r|^2) */
static double pf_eval(const orc_problem *p, const double *poses, double *pf_r, double *pf_J) {
    double cost = 0.0;
    for (uint32_t f = 0; f < p->num_pose_factors; ++f) {
        const double *T = poses + 12 * (size_t)p->pf_pose[f], *dat = p->pf_data + 18 * (size_t)f, *S = p->pf_stiffness + 36 * (size_t)f;
        double r[6] = {0}, J[36] = {0};
        const int type = (int)p->pf_type[f], dim = type == 1 ? 2 : 6;
        /* types 2 / 3: the two halves of a RelativePoseErrorAutomatic block (this pose is its first / second block, the
         * other pose is dat[12]); each half carries the Jacobian of its own pose, dat[13] says which one counts the cost */
        double counts = 1.0;
        if (type == 0) orc_pose_prior_residual(T, dat, S, r, pf_J ? J : NULL);
        else if (type == 1) orc_sun_residual(T, dat, dat + 3, S, dat[6], dat[7], r, pf_J ? J : NULL);
        else {
            const double *To = poses + 12 * (size_t)dat[12];
            if (type == 2) orc_relative_pose_residual(T, To, dat, S, r, pf_J ? J : NULL, NULL);
            else orc_relative_pose_residual(To, T, dat, S, r, NULL, pf_J ? J : NULL);
            counts = dat[13];
        }
        double sq = 0.0;
        for (int i = 0; i < dim; ++i) sq += r[i] * r[i];
        const double a = p->pf_huber ? p->pf_huber[f] : 0.0;
        if (a > 0.0) {
            double rho[3];
            orc_huber(a, sq, rho);
            cost += counts * 0.5 * rho[0];
            const double sc = sqrt(rho[1]);             /* corrector with rho'' <= 0: scale r and J by sqrt(rho') */
            for (int i = 0; i < dim; ++i) r[i] *= sc;
            if (pf_J) for (int i = 0; i < 6 * dim; ++i) J[i] *= sc;
        } else {
            cost += counts * 0.5 * sq;
        }
        if (pf_r) memcpy(pf_r + 6 * (size_t)f, r, sizeof r);
        if (pf_J) memcpy(pf_J + 36 * (size_t)f, J, sizeof J);
    }
    return cost;
}

/* Evaluate all residual blocks.  r (N*NR), Jp (N*NR*6), Jl (N*NR*LD) may be NULL.
 * Returns cost = 1/2 sum rho(|r|^2); r/J are the loss-CORRECTED quantities, as
 * Ceres's ResidualBlock::Evaluate hands them to the minimiser. */
static double evaluate(const orc_problem *p, const double *poses, const double *points,
                       const double *normals, const double *sh, double *r_out, double *Jp_out, double *Jl_out,
                       double *Jb_out) {
    const int Mm = (int)p->num_materials;
    const double *light = sh ? sh : p->light, *phong = sh ? sh + 3 : p->phong, *texture = sh ? sh + 3 + 3 * Mm : p->texture;
    const int64_t N = p->num_obs;
    const int nr = dim_nr(p), ld = dim_ld(p), ph = is_phong(p);
    double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double r[7], Jp[42], Jl[42];
        const int wantJ = (Jp_out != NULL);
        const double *T = poses + 12 * (int64_t)p->obs_pose[i];
        const int64_t j = (int64_t)p->obs_point[i];
        double r3[3], Jp3[18], Jl3[9];
        (g_jacobian_mode && wantJ ? orc_stereo_residual_autodiff : orc_stereo_residual)(
            &p->cam, T, points + 3 * j, p->obs_uvd + 3 * i, p->obs_stiffness ? p->obs_stiffness + 9 * i : p->stiffness, r3,
            wantJ ? Jp3 : NULL, wantJ ? Jl3 : NULL);
        double sq = r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2];
        if (p->huber_a > 0.0) {
            double rho[3];
            orc_huber(p->huber_a, sq, rho);
            cost += 0.5 * rho[0];
            corrector(rho, sq, r3, wantJ ? Jp3 : NULL, wantJ ? Jl3 : NULL);
        } else {
            cost += 0.5 * sq;
        }
        memcpy(r, r3, sizeof r3);
        if (wantJ) {
            memset(Jp, 0, sizeof Jp);
            memset(Jl, 0, sizeof Jl);
            for (int m = 0; m < 3; ++m) {
                for (int c = 0; c < 6; ++c) Jp[6 * m + c] = Jp3[6 * m + c];
                for (int c = 0; c < 3; ++c) Jl[ld * m + c] = Jl3[3 * m + c];
            }
        }
        if (ph) {
            const uint32_t mat = p->material_of_point[j];
            double ri, J19[19], rn[3], Jnp[18], Jnn[9];
            orc_intensity_residual(p->light_type, T, points + 3 * j, normals + 3 * j, phong + 3 * mat,
                                   texture[mat], light, p->intensity[i], p->int_stiffness, &ri,
                                   wantJ ? J19 : NULL);
            if (wantJ && Jb_out) memcpy(Jb_out + 7 * i, J19 + 12, 7 * sizeof(double));   /* [phong 3 | kd | light 3] */
            orc_normal_residual(T, normals + 3 * j, p->normal_obs + 3 * i, p->normal_stiffness, rn,
                                wantJ ? Jnp : NULL, wantJ ? Jnn : NULL);
            r[3] = ri;
            r[4] = rn[0]; r[5] = rn[1]; r[6] = rn[2];
            cost += 0.5 * (ri * ri + rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2]);
            if (wantJ) {
                for (int c = 0; c < 6; ++c) Jp[6 * 3 + c] = J19[c];
                for (int c = 0; c < 6; ++c) Jl[6 * 3 + c] = J19[6 + c];      /* [point 3 | normal 3] */
                for (int m = 0; m < 3; ++m) {
                    for (int c = 0; c < 6; ++c) Jp[6 * (4 + m) + c] = Jnp[6 * m + c];
                    for (int c = 0; c < 3; ++c) Jl[6 * (4 + m) + 3 + c] = Jnn[3 * m + c];
                }
            }
        }
        if (wantJ && p->positions_constant)      /* constant position blocks: their Jacobian columns leave the problem */
            for (int m = 0; m < nr; ++m) Jl[ld * m] = Jl[ld * m + 1] = Jl[ld * m + 2] = 0.0;
        if (r_out) memcpy(r_out + nr * i, r, (size_t)nr * sizeof(double));
        if (wantJ) {
            memcpy(Jp_out + (size_t)nr * 6 * i, Jp, (size_t)nr * 6 * sizeof(double));
            memcpy(Jl_out + (size_t)nr * ld * i, Jl, (size_t)nr * ld * sizeof(double));
        }
    }
    if (p->num_pose_factors) cost += pf_eval(p, poses, NULL, NULL);
    return cost;
}

double orc_cost(const orc_problem *p, int num_threads) {
    set_threads(num_threads);
    return evaluate(p, p->poses, p->points, p->normals, NULL, NULL, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------ */
/* graph bookkeeping                                                          */
/* ------------------------------------------------------------------------ */

typedef struct {
    int P, L, nfree, nr, ld;
    int64_t N;
    int *free_idx;       /* P: index among free poses or -1 (constant / unobserved) */
    int *free_pose;      /* nfree -> pose id                                      */
    uint8_t *pt_active;  /* L: has at least one observation                       */
    int64_t *pose_start; /* P+1 CSR over observations (pose-major)                */
    int64_t *pose_obs;   /* N                                                     */
    int64_t *pt_start;   /* L+1 CSR (landmark-major)                              */
    int64_t *pt_obs;     /* N                                                     */
    int bw_poses;        /* max |free_idx(a)-free_idx(b)| over co-observing poses */
    int *lo_pose;        /* nfree: lowest free index coupled to this free pose (<= itself): row profile of S */
    /* free shared blocks ("border"): column offsets in the local border vector, -1 = constant */
    int M, nb, b_light, b_phong, b_tex;
    int *pf_start, *pf_list;   /* P+1 CSR over the unary pose factors */
    int pos_const;             /* position blocks held constant: not part of x */
} graph_t;

/* border column of entry q of the per-observation border Jacobian [phong 3 | kd | light 3] */
static int bcol(const graph_t *g, uint32_t mat, int q) {
    if (q < 3) return g->b_phong < 0 ? -1 : g->b_phong + 3 * (int)mat + q;
    if (q == 3) return g->b_tex < 0 ? -1 : g->b_tex + (int)mat;
    return g->b_light < 0 ? -1 : g->b_light + (q - 4);
}

static void graph_free(graph_t *g) {
    free(g->free_idx); free(g->free_pose); free(g->pt_active);
    free(g->pose_start); free(g->pose_obs); free(g->pt_start); free(g->pt_obs);
    free(g->pf_start); free(g->pf_list); free(g->lo_pose);
}

static void graph_build(const orc_problem *p, graph_t *g) {
    memset(g, 0, sizeof *g);
    int P = p->num_poses, L = p->num_points;
    int64_t N = p->num_obs;
    g->P = P; g->L = L; g->N = N;
    g->nr = dim_nr(p); g->ld = dim_ld(p);
    g->pos_const = p->positions_constant != 0;
    g->M = is_phong(p) ? (int)p->num_materials : 0;
    g->b_light = g->b_phong = g->b_tex = -1;
    if (is_phong(p)) {
        if (p->shared_free & 1u) { g->b_light = g->nb; g->nb += 3; }
        if (p->shared_free & 2u) { g->b_phong = g->nb; g->nb += 3 * g->M; }
        if (p->shared_free & 4u) { g->b_tex = g->nb; g->nb += g->M; }
    }
    g->pose_start = calloc((size_t)P + 1, sizeof(int64_t));
    g->pt_start = calloc((size_t)L + 1, sizeof(int64_t));
    g->pose_obs = malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    g->pt_obs = malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < N; ++i) {
        g->pose_start[p->obs_pose[i] + 1]++;
        g->pt_start[p->obs_point[i] + 1]++;
    }
    for (int k = 0; k < P; ++k) g->pose_start[k + 1] += g->pose_start[k];
    for (int j = 0; j < L; ++j) g->pt_start[j + 1] += g->pt_start[j];
    int64_t *pc = malloc((size_t)(P > 0 ? P : 1) * sizeof(int64_t)), *lc = malloc((size_t)(L > 0 ? L : 1) * sizeof(int64_t));
    memcpy(pc, g->pose_start, (size_t)P * sizeof(int64_t));
    memcpy(lc, g->pt_start, (size_t)L * sizeof(int64_t));
    for (int64_t i = 0; i < N; ++i) { /* stable: keeps the reference's file order */
        g->pose_obs[pc[p->obs_pose[i]]++] = i;
        g->pt_obs[lc[p->obs_point[i]]++] = i;
    }
    free(pc); free(lc);
    g->free_idx = malloc((size_t)(P > 0 ? P : 1) * sizeof(int));
    g->free_pose = malloc((size_t)(P > 0 ? P : 1) * sizeof(int));
    g->pt_active = malloc((size_t)(L > 0 ? L : 1));
    g->pf_start = calloc((size_t)P + 1, sizeof(int));
    g->pf_list = malloc((size_t)(p->num_pose_factors > 0 ? p->num_pose_factors : 1) * sizeof(int));
    for (uint32_t f = 0; f < p->num_pose_factors; ++f) g->pf_start[p->pf_pose[f] + 1]++;
    for (int k = 0; k < P; ++k) g->pf_start[k + 1] += g->pf_start[k];
    {
        int *cur = malloc((size_t)(P > 0 ? P : 1) * sizeof(int));
        memcpy(cur, g->pf_start, (size_t)P * sizeof(int));
        for (uint32_t f = 0; f < p->num_pose_factors; ++f) g->pf_list[cur[p->pf_pose[f]]++] = (int)f;
        free(cur);
    }
    int nf = 0;
    for (int k = 0; k < P; ++k) {
        int in_problem = g->pose_start[k + 1] > g->pose_start[k] || g->pf_start[k + 1] > g->pf_start[k];
        int is_const = p->pose_const && p->pose_const[k];
        if (in_problem && !is_const) { g->free_idx[k] = nf; g->free_pose[nf++] = k; }
        else g->free_idx[k] = -1;
    }
    g->nfree = nf;
    g->lo_pose = malloc((size_t)(nf > 0 ? nf : 1) * sizeof(int));
    for (int f = 0; f < nf; ++f) g->lo_pose[f] = f;
    int bw = 0;
    for (int j = 0; j < L; ++j) {
        g->pt_active[j] = g->pt_start[j + 1] > g->pt_start[j];
        int lo = 1 << 30, hi = -1;
        for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
            int f = g->free_idx[p->obs_pose[g->pt_obs[e]]];
            if (f < 0) continue;
            if (f < lo) lo = f;
            if (f > hi) hi = f;
        }
        if (hi >= 0 && hi - lo > bw) bw = hi - lo;
        for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
            int f = g->free_idx[p->obs_pose[g->pt_obs[e]]];
            if (f >= 0 && lo < g->lo_pose[f]) g->lo_pose[f] = lo;
        }
    }
    for (uint32_t f = 0; f < p->num_pose_factors; ++f)      /* relative-pose blocks couple their two poses */
        if (p->pf_type[f] == 2) {
            const int fa = g->free_idx[p->pf_pose[f]], fb = g->free_idx[(int)p->pf_data[18 * (size_t)f + 12]];
            if (fa >= 0 && fb >= 0 && abs(fa - fb) > bw) bw = abs(fa - fb);
            if (fa >= 0 && fb >= 0) {
                const int hi = fa > fb ? fa : fb, lo = fa > fb ? fb : fa;
                if (lo < g->lo_pose[hi]) g->lo_pose[hi] = lo;
            }
        }
    g->bw_poses = bw;
}

/* ------------------------------------------------------------------------ */
/* small dense helpers                                                        */
/* ------------------------------------------------------------------------ */

/* inverse of an n x n (n <= 6) SPD matrix through its Cholesky factor
 * [Ceres: InvertPSDMatrix -> LLT solve for fixed-size blocks] */
static int inv_spd(int n, const double *C, double *Ci) {
    double Lm[36] = {0}, M[36] = {0};
    for (int j = 0; j < n; ++j) {
        double d = C[n * j + j];
        for (int k = 0; k < j; ++k) d -= Lm[n * j + k] * Lm[n * j + k];
        if (!(d > 0.0) || !isfinite(d)) return -1;
        Lm[n * j + j] = sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = C[n * i + j];
            for (int k = 0; k < j; ++k) s -= Lm[n * i + k] * Lm[n * j + k];
            Lm[n * i + j] = s / Lm[n * j + j];
        }
    }
    for (int j = 0; j < n; ++j) {   /* M = L^-1 */
        M[n * j + j] = 1.0 / Lm[n * j + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= Lm[n * i + k] * M[n * k + j];
            M[n * i + j] = s / Lm[n * i + i];
        }
    }
    for (int a = 0; a < n; ++a)     /* C^-1 = M^T M */
        for (int b = 0; b < n; ++b) {
            double s = 0.0;
            for (int k = (a > b ? a : b); k < n; ++k) s += M[n * k + a] * M[n * k + b];
            Ci[n * a + b] = s;
        }
    return 0;
}

/* Profile (envelope) Cholesky.  Row i of the lower triangle is stored from its first structural column first[i] to the
 * diagonal: entry (i, j) at A[rp[i] + j - first[i]].  Cholesky fill stays inside the profile (George & Liu), so a banded
 * trajectory costs what a band solver costs, and a loop closure -- a few late poses coupled to the first ones -- only
 * makes those few rows long instead of widening a band to the whole matrix.  Every entry is ONE sum in ascending k,
 * the same arithmetic as the band version this replaces (entries of the band outside the profile were exact zeros). */
typedef struct {
    int n;
    int *first;      /* n                                        */
    size_t *rp;      /* n + 1  row offsets                        */
    size_t *cp;      /* n + 1  column lists (rows k > i with first[k] <= i), for the backward substitution */
    int *crow;
} profile_t;

static void profile_free(profile_t *pr) {
    free(pr->first); free(pr->rp); free(pr->cp); free(pr->crow);
    memset(pr, 0, sizeof *pr);
}

/* first[] given: row offsets and the column lists */
static void profile_finish(profile_t *pr) {
    const int n = pr->n;
    pr->rp = malloc(((size_t)n + 1) * sizeof(size_t));
    pr->cp = calloc((size_t)n + 2, sizeof(size_t));
    pr->rp[0] = 0;
    for (int i = 0; i < n; ++i) {
        pr->rp[i + 1] = pr->rp[i] + (size_t)(i - pr->first[i] + 1);
        for (int j = pr->first[i]; j < i; ++j) pr->cp[j + 2]++;
    }
    for (int j = 0; j < n; ++j) pr->cp[j + 2] += pr->cp[j + 1];
    pr->crow = malloc((pr->cp[n + 1] > 0 ? pr->cp[n + 1] : 1) * sizeof(int));
    for (int i = 0; i < n; ++i)        /* rows ascending: every column list ends up ascending */
        for (int j = pr->first[i]; j < i; ++j) pr->crow[pr->cp[j + 1]++] = i;
    /* cp[j + 1] now marks the end of column j's list, i.e. cp[j] .. cp[j + 1] is the list of column j */
}

static int profile_cholesky(double *A, const profile_t *pr) {
    const int n = pr->n;
    for (int i = 0; i < n; ++i) {
        double *Ai = A + pr->rp[i];
        const int fi = pr->first[i];
        for (int j = fi; j <= i; ++j) {
            const double *Aj = A + pr->rp[j];
            const int fj = pr->first[j];
            double s = Ai[j - fi];
            for (int k = fi > fj ? fi : fj; k < j; ++k) s -= Ai[k - fi] * Aj[k - fj];
            if (j < i) {
                Ai[j - fi] = s / Aj[j - fj];
            } else {
                if (!(s > 0.0) || !isfinite(s)) return -1;
                Ai[j - fi] = sqrt(s);
            }
        }
    }
    return 0;
}

static void profile_solve(const double *A, const profile_t *pr, double *x) {
    const int n = pr->n;
    for (int i = 0; i < n; ++i) {
        const double *Ai = A + pr->rp[i];
        const int fi = pr->first[i];
        double s = x[i];
        for (int k = fi; k < i; ++k) s -= Ai[k - fi] * x[k];
        x[i] = s / Ai[i - fi];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (size_t q = pr->cp[i]; q < pr->cp[i + 1]; ++q) {
            const int k = pr->crow[q];
            s -= A[pr->rp[k] + (size_t)(i - pr->first[k])] * x[k];
        }
        x[i] = s / A[pr->rp[i] + (size_t)(i - pr->first[i])];
    }
}

/* ------------------------------------------------------------------------ */
/* linearisation + LM step                                                    */
/* ------------------------------------------------------------------------ */

typedef struct {
    double *r;       /* N*NR     corrected residuals                 */
    double *Jp;      /* N*NR*6   corrected, unscaled                 */
    double *Jl;      /* N*NR*LD                                      */
    double *g_p;     /* nfree*6  unscaled gradient J^T r             */
    double *g_l;     /* L*LD                                         */
    double *sq_p;    /* nfree*6  squared column norms of unscaled J  */
    double *sq_l;    /* L*LD                                         */
    double *Jb;      /* N*7 border entries of the intensity row, or NULL */
    double *g_b, *sq_b;   /* nb                                      */
    double *pf_r, *pf_J;  /* 6 / 36 per unary pose factor             */
    double cost;
} lin_t;

static void lin_alloc(lin_t *w, const graph_t *g) {
    size_t N = (size_t)(g->N > 0 ? g->N : 1), nf = (size_t)(g->nfree > 0 ? g->nfree : 1),
           L = (size_t)(g->L > 0 ? g->L : 1);
    w->r = malloc(N * g->nr * sizeof(double));
    w->Jp = malloc(N * g->nr * 6 * sizeof(double));
    w->Jl = malloc(N * g->nr * g->ld * sizeof(double));
    w->g_p = malloc(nf * 6 * sizeof(double));
    w->g_l = malloc(L * g->ld * sizeof(double));
    w->sq_p = malloc(nf * 6 * sizeof(double));
    w->sq_l = malloc(L * g->ld * sizeof(double));
    w->Jb = g->nb ? malloc(N * 7 * sizeof(double)) : NULL;
    w->g_b = calloc((size_t)g->nb + 1, sizeof(double));
    w->sq_b = calloc((size_t)g->nb + 1, sizeof(double));
    w->pf_r = calloc((size_t)(g->pf_start[g->P] + 1) * 6, sizeof(double));
    w->pf_J = calloc((size_t)(g->pf_start[g->P] + 1) * 36, sizeof(double));
}
static void lin_free(lin_t *w) {
    free(w->r); free(w->Jp); free(w->Jl); free(w->g_p); free(w->g_l); free(w->sq_p); free(w->sq_l);
    free(w->Jb); free(w->g_b); free(w->sq_b); free(w->pf_r); free(w->pf_J);
}

/* [Ceres evaluator: residuals, cost, Jacobian, gradient = J^T r at x] */
static void linearize(const orc_problem *p, const graph_t *g, const double *poses,
                      const double *points, const double *normals, const double *sh, lin_t *w) {
    const int nr = g->nr, ld = g->ld;
    w->cost = evaluate(p, poses, points, normals, sh, w->r, w->Jp, w->Jl, w->Jb);
    if (p->num_pose_factors) pf_eval(p, poses, w->pf_r, w->pf_J);
    if (g->nb) {   /* border gradient and squared column norms: only the intensity row (3) touches it */
        memset(w->g_b, 0, (size_t)g->nb * sizeof(double));
        memset(w->sq_b, 0, (size_t)g->nb * sizeof(double));
        for (int64_t i = 0; i < g->N; ++i) {
            const uint32_t mat = p->material_of_point[p->obs_point[i]];
            const double ri = w->r[(size_t)nr * i + 3];
            for (int q = 0; q < 7; ++q) {
                const int c = bcol(g, mat, q);
                if (c < 0) continue;
                const double v = w->Jb[7 * i + q];
                w->g_b[c] += v * ri;
                w->sq_b[c] += v * v;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int f = 0; f < g->nfree; ++f) {
        int k = g->free_pose[f];
        double gp[6] = {0}, sq[6] = {0};
        for (int64_t e = g->pose_start[k]; e < g->pose_start[k + 1]; ++e) {
            int64_t i = g->pose_obs[e];
            const double *J = w->Jp + (size_t)nr * 6 * i, *r = w->r + (size_t)nr * i;
            for (int m = 0; m < nr; ++m)
                for (int c = 0; c < 6; ++c) {
                    gp[c] += J[6 * m + c] * r[m];
                    sq[c] += J[6 * m + c] * J[6 * m + c];
                }
        }
        for (int e = g->pf_start[k]; e < g->pf_start[k + 1]; ++e) {
            const double *J = w->pf_J + 36 * (size_t)g->pf_list[e], *r = w->pf_r + 6 * (size_t)g->pf_list[e];
            for (int m = 0; m < 6; ++m)
                for (int c = 0; c < 6; ++c) { gp[c] += J[6 * m + c] * r[m]; sq[c] += J[6 * m + c] * J[6 * m + c]; }
        }
        memcpy(w->g_p + 6 * f, gp, sizeof gp);
        memcpy(w->sq_p + 6 * f, sq, sizeof sq);
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < g->L; ++j) {
        double gl[6] = {0}, sq[6] = {0};
        for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
            int64_t i = g->pt_obs[e];
            const double *J = w->Jl + (size_t)nr * ld * i, *r = w->r + (size_t)nr * i;
            for (int m = 0; m < nr; ++m)
                for (int c = 0; c < ld; ++c) {
                    gl[c] += J[ld * m + c] * r[m];
                    sq[c] += J[ld * m + c] * J[ld * m + c];
                }
        }
        memcpy(w->g_l + (size_t)ld * j, gl, (size_t)ld * sizeof(double));
        memcpy(w->sq_l + (size_t)ld * j, sq, (size_t)ld * sizeof(double));
    }
}

double orc_linearize(const orc_problem *p, double *g_p, double *g_l, double *H_pp,
                     double *H_ll, int num_threads) {
    set_threads(num_threads);
    const int64_t N = p->num_obs;
    const int nr = dim_nr(p), ld = dim_ld(p);
    size_t n = (size_t)(N > 0 ? N : 1);
    double *r = malloc(n * nr * sizeof(double)), *Jp = malloc(n * nr * 6 * sizeof(double)),
           *Jl = malloc(n * nr * ld * sizeof(double));
    double cost = evaluate(p, p->poses, p->points, p->normals, NULL, r, Jp, Jl, NULL);
    memset(g_p, 0, (size_t)p->num_poses * 6 * sizeof(double));
    memset(g_l, 0, (size_t)p->num_points * ld * sizeof(double));
    memset(H_pp, 0, (size_t)p->num_poses * 36 * sizeof(double));
    memset(H_ll, 0, (size_t)p->num_points * ld * ld * sizeof(double));
    for (int64_t i = 0; i < N; ++i) { /* serial, reference residual-block order */
        const double *a = Jp + (size_t)nr * 6 * i, *b = Jl + (size_t)nr * ld * i, *ri = r + (size_t)nr * i;
        double *gp = g_p + 6 * (size_t)p->obs_pose[i], *gl = g_l + (size_t)ld * p->obs_point[i];
        double *hp = H_pp + 36 * (size_t)p->obs_pose[i], *hl = H_ll + (size_t)ld * ld * p->obs_point[i];
        for (int m = 0; m < nr; ++m) {
            for (int c = 0; c < 6; ++c) {
                gp[c] += a[6 * m + c] * ri[m];
                for (int d = 0; d < 6; ++d) hp[6 * c + d] += a[6 * m + c] * a[6 * m + d];
            }
            for (int c = 0; c < ld; ++c) {
                gl[c] += b[ld * m + c] * ri[m];
                for (int d = 0; d < ld; ++d) hl[ld * c + d] += b[ld * m + c] * b[ld * m + d];
            }
        }
    }
    free(r); free(Jp); free(Jl);
    return cost;
}

/* Jacobi scaling [Ceres trust_region_minimizer.cc IterationZero]:
 *   scale = 1 / (1 + sqrt(squared column norm)) computed once at iteration 0 */
static void jacobi_scale(const graph_t *g, const lin_t *w, int enabled, double *sp, double *sl, double *sb) {
    for (int i = 0; i < g->nfree * 6; ++i) sp[i] = enabled ? 1.0 / (1.0 + sqrt(w->sq_p[i])) : 1.0;
    for (int i = 0; i < g->L * g->ld; ++i) sl[i] = enabled ? 1.0 / (1.0 + sqrt(w->sq_l[i])) : 1.0;
    for (int i = 0; i < g->nb; ++i) sb[i] = enabled ? 1.0 / (1.0 + sqrt(w->sq_b[i])) : 1.0;
}

typedef struct {
    double *S;    /* profile storage (profile_t pr): row i from its first column to the diagonal */
    profile_t pr;
    double *rhs;  /* n                                                     */
    double *Ci;   /* L*LD*LD inverse of damped landmark blocks (scaled)     */
    double *W;    /* N*6*LD  Jp_s^T Jl_s                                    */
    double *gl_s; /* L*LD    scaled landmark gradient                       */
    int n, bw;
    double t_schur, t_solve;
    /* border (free shared blocks), scaled coordinates */
    int nb;
    double *Spb;  /* n x nb   S_pb = H_pb - sum W C^-1 V                    */
    double *Sbb;  /* nb x nb  H_bb + D_b^2 - sum V^T C^-1 V                 */
    double *rhs_b;/* nb       g_b - sum V^T C^-1 g_l                        */
    double *V;    /* L*LD*7   H_lb of landmark j: columns [phong 3 | kd | light 3] of its material */
} schur_t;

/* Build the Schur-complemented reduced camera system in SCALED coordinates
 * [Ceres schur_eliminator_impl.h restated; LM diagonal of
 * levenberg_marquardt_strategy.cc: D^2 = clamp(diag(J_s^T J_s), min, max)/radius]. */
static int build_reduced(const orc_problem *p, const graph_t *g, const lin_t *w,
                         const double *sp, const double *sl, const double *sb, double radius,
                         const orc_options *o, schur_t *sc) {
    const int nf = g->nfree, L = g->L, nr = g->nr, ld = g->ld, nb = g->nb;
    const int n = 6 * nf;
    int bw = 6 * (g->bw_poses + 1) - 1;
    if (bw > n - 1) bw = n - 1;
    if (bw < 0) bw = 0;
    sc->n = n; sc->bw = bw;
    sc->pr.n = n;
    sc->pr.first = malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int f = 0; f < nf; ++f)
        for (int c = 0; c < 6; ++c) sc->pr.first[6 * f + c] = 6 * g->lo_pose[f];
    profile_finish(&sc->pr);
#define SIDX(i, j) (sc->pr.rp[(i)] + (size_t)((j) - sc->pr.first[(i)]))
    sc->S = calloc(sc->pr.rp[n] > 0 ? sc->pr.rp[n] : 1, sizeof(double));
    sc->rhs = calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    sc->Ci = malloc((size_t)(L > 0 ? L : 1) * ld * ld * sizeof(double));
    sc->W = malloc((size_t)(g->N > 0 ? g->N : 1) * 6 * ld * sizeof(double));
    sc->gl_s = malloc((size_t)(L > 0 ? L : 1) * ld * sizeof(double));
    sc->nb = nb;
    if (nb) {
        sc->Spb = calloc((size_t)(n > 0 ? n : 1) * nb, sizeof(double));
        sc->Sbb = calloc((size_t)nb * nb, sizeof(double));
        sc->rhs_b = calloc((size_t)nb, sizeof(double));
        sc->V = calloc((size_t)(L > 0 ? L : 1) * ld * 7, sizeof(double));
    }
    int bad = 0;

    /* landmark blocks C_j = sum Jl_s^T Jl_s + D_l^2 and their inverses */
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int j = 0; j < L; ++j) {
        double C[36] = {0};
        const double *s = sl + (size_t)ld * j;
        for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
            const double *J = w->Jl + (size_t)nr * ld * g->pt_obs[e];
            for (int m = 0; m < nr; ++m)
                for (int a = 0; a < ld; ++a)
                    for (int b = 0; b < ld; ++b) C[ld * a + b] += J[ld * m + a] * J[ld * m + b] * s[a] * s[b];
        }
        for (int a = 0; a < ld; ++a) {
            double d = w->sq_l[(size_t)ld * j + a] * s[a] * s[a];
            d = fmin(fmax(d, o->min_lm_diagonal), o->max_lm_diagonal);
            C[(ld + 1) * a] += d / radius;
            sc->gl_s[(size_t)ld * j + a] = w->g_l[(size_t)ld * j + a] * s[a];
        }
        if (!g->pt_active[j]) { memset(sc->Ci + (size_t)ld * ld * j, 0, (size_t)ld * ld * sizeof(double)); continue; }
        if (inv_spd(ld, C, sc->Ci + (size_t)ld * ld * j)) bad |= 1;
    }
    if (bad) return -1;

    if (nb) {
        /* H_bb + D_b^2 and g_b (serial: reference residual-block order), V_j = H_lb of each landmark */
        for (int64_t i = 0; i < g->N; ++i) {
            const uint32_t mat = p->material_of_point[p->obs_point[i]];
            const double *jb = w->Jb + 7 * i;
            for (int q = 0; q < 7; ++q) {
                const int c = bcol(g, mat, q);
                if (c < 0) continue;
                for (int q2 = 0; q2 < 7; ++q2) {
                    const int c2 = bcol(g, mat, q2);
                    if (c2 >= 0) sc->Sbb[(size_t)c * nb + c2] += jb[q] * sb[c] * jb[q2] * sb[c2];
                }
            }
        }
        for (int c = 0; c < nb; ++c) {
            double d = w->sq_b[c] * sb[c] * sb[c];
            d = fmin(fmax(d, o->min_lm_diagonal), o->max_lm_diagonal);
            sc->Sbb[(size_t)c * nb + c] += d / radius;
            sc->rhs_b[c] = w->g_b[c] * sb[c];
        }
#pragma omp parallel for schedule(static)
        for (int j = 0; j < L; ++j) {
            if (!g->pt_active[j]) continue;
            const uint32_t mat = p->material_of_point[j];
            double *V = sc->V + (size_t)ld * 7 * j;
            for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
                const int64_t i = g->pt_obs[e];
                const double *Jl3 = w->Jl + (size_t)nr * ld * i + (size_t)ld * 3;   /* intensity row */
                for (int a = 0; a < ld; ++a)
                    for (int q = 0; q < 7; ++q) {
                        const int c = bcol(g, mat, q);
                        if (c >= 0) V[7 * a + q] += Jl3[a] * sl[(size_t)ld * j + a] * w->Jb[7 * i + q] * sb[c];
                    }
            }
        }
        /* S_bb -= sum_j V^T C^-1 V ; rhs_b -= sum_j V^T C^-1 g_l   (serial over landmarks) */
        for (int j = 0; j < L; ++j) {
            if (!g->pt_active[j]) continue;
            const uint32_t mat = p->material_of_point[j];
            const double *V = sc->V + (size_t)ld * 7 * j, *Ci = sc->Ci + (size_t)ld * ld * j;
            double CV[42], Cg[6];
            for (int a = 0; a < ld; ++a) {
                for (int q = 0; q < 7; ++q) {
                    double v = 0.0;
                    for (int b = 0; b < ld; ++b) v += Ci[ld * a + b] * V[7 * b + q];
                    CV[7 * a + q] = v;
                }
                double v = 0.0;
                for (int b = 0; b < ld; ++b) v += Ci[ld * a + b] * sc->gl_s[(size_t)ld * j + b];
                Cg[a] = v;
            }
            for (int q = 0; q < 7; ++q) {
                const int c = bcol(g, mat, q);
                if (c < 0) continue;
                double vg = 0.0;
                for (int a = 0; a < ld; ++a) vg += V[7 * a + q] * Cg[a];
                sc->rhs_b[c] -= vg;
                for (int q2 = 0; q2 < 7; ++q2) {
                    const int c2 = bcol(g, mat, q2);
                    if (c2 < 0) continue;
                    double v = 0.0;
                    for (int a = 0; a < ld; ++a) v += V[7 * a + q] * CV[7 * a + q2];
                    sc->Sbb[(size_t)c * nb + c2] -= v;
                }
            }
        }
    }

    /* W_i = Jp_s^T Jl_s (6 x LD) for observations of free poses */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < g->N; ++i) {
        int f = g->free_idx[p->obs_pose[i]];
        double *Wi = sc->W + (size_t)6 * ld * i;
        if (f < 0) { memset(Wi, 0, (size_t)6 * ld * sizeof(double)); continue; }
        const double *a = w->Jp + (size_t)nr * 6 * i, *b = w->Jl + (size_t)nr * ld * i;
        const double *s6 = sp + 6 * f, *sL = sl + (size_t)ld * p->obs_point[i];
        for (int c = 0; c < 6; ++c)
            for (int d = 0; d < ld; ++d) {
                double v = 0.0;
                for (int m = 0; m < nr; ++m) v += a[6 * m + c] * b[ld * m + d];
                Wi[ld * c + d] = v * s6[c] * sL[d];
            }
    }

    /* block-rows of S: each free pose owns its row (lower band) */
#pragma omp parallel for schedule(dynamic, 4)
    for (int f = 0; f < nf; ++f) {
        int k = g->free_pose[f];
        double B[36] = {0}, gr[6];
        const double *s6 = sp + 6 * f;
        for (int c = 0; c < 6; ++c) gr[c] = w->g_p[6 * f + c] * s6[c];
        for (int64_t e = g->pose_start[k]; e < g->pose_start[k + 1]; ++e) {
            int64_t i = g->pose_obs[e];
            const double *J = w->Jp + (size_t)nr * 6 * i;
            for (int m = 0; m < nr; ++m)
                for (int c = 0; c < 6; ++c)
                    for (int d = 0; d <= c; ++d) B[6 * c + d] += J[6 * m + c] * J[6 * m + d] * s6[c] * s6[d];
        }
        for (int e = g->pf_start[k]; e < g->pf_start[k + 1]; ++e) {
            const double *J = w->pf_J + 36 * (size_t)g->pf_list[e];
            for (int m = 0; m < 6; ++m)
                for (int c = 0; c < 6; ++c)
                    for (int d = 0; d <= c; ++d) B[6 * c + d] += J[6 * m + c] * J[6 * m + d] * s6[c] * s6[d];
        }
        for (int c = 0; c < 6; ++c) {
            double d = w->sq_p[6 * f + c] * s6[c] * s6[c];
            d = fmin(fmax(d, o->min_lm_diagonal), o->max_lm_diagonal);
            B[7 * c] += d / radius;
        }
        for (int c = 0; c < 6; ++c)
            for (int d = 0; d <= c; ++d)
                sc->S[SIDX(6 * f + c, 6 * f + d)] += B[6 * c + d];
        for (int64_t e = g->pose_start[k]; e < g->pose_start[k + 1]; ++e) {
            int64_t i = g->pose_obs[e];
            int j = (int)p->obs_point[i];
            const double *Wi = sc->W + (size_t)6 * ld * i, *Ci = sc->Ci + (size_t)ld * ld * j;
            double Y[36]; /* W_i C^-1 (6 x LD) */
            for (int c = 0; c < 6; ++c)
                for (int d = 0; d < ld; ++d) {
                    double v = 0.0;
                    for (int q = 0; q < ld; ++q) v += Wi[ld * c + q] * Ci[ld * q + d];
                    Y[ld * c + d] = v;
                }
            for (int c = 0; c < 6; ++c)
                for (int q = 0; q < ld; ++q) gr[c] -= Y[ld * c + q] * sc->gl_s[(size_t)ld * j + q];
            if (nb) {   /* S_pb rows of this pose: J_p^T J_b (intensity row) - Y V_j */
                const uint32_t mat = p->material_of_point[j];
                const double *Jp3 = w->Jp + (size_t)nr * 6 * i + 18, *V = sc->V + (size_t)ld * 7 * j;
                for (int q = 0; q < 7; ++q) {
                    const int cb = bcol(g, mat, q);
                    if (cb < 0) continue;
                    for (int c = 0; c < 6; ++c) {
                        double v = Jp3[c] * s6[c] * w->Jb[7 * i + q] * sb[cb];
                        for (int a = 0; a < ld; ++a) v -= Y[ld * c + a] * V[7 * a + q];
                        sc->Spb[(size_t)(6 * f + c) * nb + cb] += v;
                    }
                }
            }
            for (int64_t e2 = g->pt_start[j]; e2 < g->pt_start[j + 1]; ++e2) {
                int64_t i2 = g->pt_obs[e2];
                int f2 = g->free_idx[p->obs_pose[i2]];
                if (f2 < 0 || f2 > f) continue;
                const double *W2 = sc->W + (size_t)6 * ld * i2;
                for (int c = 0; c < 6; ++c) {
                    int dmax = (f2 == f) ? c : 5;
                    for (int d = 0; d <= dmax; ++d) {
                        double v = 0.0;
                        for (int q = 0; q < ld; ++q) v += Y[ld * c + q] * W2[ld * d + q];
                        sc->S[SIDX(6 * f + c, 6 * f2 + d)] -= v;
                    }
                }
            }
        }
        memcpy(sc->rhs + 6 * f, gr, sizeof gr);
    }
    /* relative-pose blocks: the off-diagonal block J_a^T J_b of their two poses (lower band) */
    for (uint32_t e = 0; e < p->num_pose_factors; ++e) {
        if (p->pf_type[e] != 2) continue;
        const int other = (int)p->pf_data[18 * (size_t)e + 14];      /* index of the second half */
        const int fa = g->free_idx[p->pf_pose[e]], fb = g->free_idx[p->pf_pose[other]];
        if (fa < 0 || fb < 0) continue;
        const double *Ja = w->pf_J + 36 * (size_t)e, *Jb = w->pf_J + 36 * (size_t)other;
        const int hi = fa > fb ? fa : fb, lo = fa > fb ? fb : fa;
        const double *Jh = fa > fb ? Ja : Jb, *Jl2 = fa > fb ? Jb : Ja;
        for (int c = 0; c < 6; ++c)
            for (int d = 0; d < 6; ++d) {
                double v = 0.0;
                for (int m = 0; m < 6; ++m) v += Jh[6 * m + c] * Jl2[6 * m + d];
                sc->S[SIDX(6 * hi + c, 6 * lo + d)] += v * sp[6 * hi + c] * sp[6 * lo + d];
            }
    }
    return 0;
}

#undef SIDX
static void schur_free(schur_t *sc) {
    profile_free(&sc->pr);
    free(sc->S); free(sc->rhs); free(sc->Ci); free(sc->W); free(sc->gl_s);
    free(sc->Spb); free(sc->Sbb); free(sc->rhs_b); free(sc->V);
}

/* -(J d)^T (r + J d / 2) for an arbitrary step (dp: P*6, dl: L*LD), and |J d|^2 */
static void step_products(const orc_problem *p, const graph_t *g, const lin_t *w, const double *dp,
                          const double *dl, const double *db, double *mcc_out, double *jd_sq_out) {
    const int nr = g->nr, ld = g->ld;
    double mcc = 0.0, sq = 0.0;
#pragma omp parallel for reduction(+ : mcc, sq) schedule(static)
    for (int64_t i = 0; i < g->N; ++i) {
        const double *a = w->Jp + (size_t)nr * 6 * i, *b = w->Jl + (size_t)nr * ld * i, *r = w->r + (size_t)nr * i;
        const double *d6 = dp + 6 * (size_t)p->obs_pose[i], *dL = dl + (size_t)ld * p->obs_point[i];
        const int fr = g->free_idx[p->obs_pose[i]] >= 0;
        for (int m = 0; m < nr; ++m) {
            double jd = 0.0;
            for (int c = 0; c < ld; ++c) jd += b[ld * m + c] * dL[c];
            if (fr)
                for (int c = 0; c < 6; ++c) jd += a[6 * m + c] * d6[c];
            if (m == 3 && g->nb && db) {
                const uint32_t mat = p->material_of_point[p->obs_point[i]];
                for (int q = 0; q < 7; ++q) {
                    const int c = bcol(g, mat, q);
                    if (c >= 0) jd += w->Jb[7 * i + q] * db[c];
                }
            }
            mcc -= jd * (r[m] + 0.5 * jd);
            sq += jd * jd;
        }
    }
    for (int k = 0; k < g->P; ++k) {
        if (g->free_idx[k] < 0) continue;
        for (int e = g->pf_start[k]; e < g->pf_start[k + 1]; ++e) {
            const double *J = w->pf_J + 36 * (size_t)g->pf_list[e], *r = w->pf_r + 6 * (size_t)g->pf_list[e];
            for (int m = 0; m < 6; ++m) {
                double jd = 0.0;
                for (int c = 0; c < 6; ++c) jd += J[6 * m + c] * dp[6 * (size_t)k + c];
                mcc -= jd * (r[m] + 0.5 * jd);
                sq += jd * jd;
            }
        }
    }
    for (uint32_t e = 0; e < p->num_pose_factors; ++e) {      /* relative-pose blocks: (J_a d_a + J_b d_b) is one row vector */
        if (p->pf_type[e] != 2) continue;
        const int other = (int)p->pf_data[18 * (size_t)e + 14], ka = (int)p->pf_pose[e], kb = (int)p->pf_pose[other];
        if (g->free_idx[ka] < 0 || g->free_idx[kb] < 0) continue;
        const double *Ja = w->pf_J + 36 * (size_t)e, *Jb = w->pf_J + 36 * (size_t)other;
        for (int m = 0; m < 6; ++m) {
            double ja = 0.0, jb = 0.0;
            for (int c = 0; c < 6; ++c) { ja += Ja[6 * m + c] * dp[6 * (size_t)ka + c]; jb += Jb[6 * m + c] * dp[6 * (size_t)kb + c]; }
            mcc -= ja * jb;
            sq += 2.0 * ja * jb;
        }
    }
    if (mcc_out) *mcc_out = mcc;
    if (jd_sq_out) *jd_sq_out = sq;
}

/* One LM step [Ceres LevenbergMarquardtStrategy::ComputeStep + SchurComplementSolver].
 * Outputs the UNSCALED step (delta = scale .* step_scaled) and the model cost
 * change  -(J d)^T (r + J d / 2)  [TrustRegionMinimizer::ComputeTrustRegionStep]. */
/* dense Cholesky solve A x = b in place (A n x n row-major, destroyed); -1 on breakdown */
static int dense_spd_solve(double *A, int n, double *b) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 0;
}

static int lm_step(const orc_problem *p, const graph_t *g, const lin_t *w, const double *sp,
                   const double *sl, const double *sb, double radius, const orc_options *o, double *dp,
                   double *dl, double *db, double *model_cost_change, double *t_schur, double *t_solve) {
    const int ld = g->ld, nb = g->nb;
    schur_t sc;
    memset(&sc, 0, sizeof sc);
    double t0 = now_s();
    int rc = build_reduced(p, g, w, sp, sl, sb, radius, o, &sc);
    double t1 = now_s();
    if (t_schur) *t_schur += t1 - t0;
    if (rc) { schur_free(&sc); return -1; }
    if (sc.n > 0) {
        if (profile_cholesky(sc.S, &sc.pr)) { schur_free(&sc); return -1; }
        profile_solve(sc.S, &sc.pr, sc.rhs);
    }
    double *yb = calloc((size_t)nb + 1, sizeof(double));
    if (nb) {
        /* arrowhead system [S_pp S_pb; S_pb^T S_bb]: Z = S_pp^-1 S_pb, border Schur complement, then
         * y_p = S_pp^-1 rhs_p - Z y_b */
        const int n = sc.n;
        double *Z = malloc((size_t)(n > 0 ? n : 1) * nb * sizeof(double)), *col = malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
        for (int c = 0; c < nb; ++c) {
            for (int i = 0; i < n; ++i) col[i] = sc.Spb[(size_t)i * nb + c];
            if (n > 0) profile_solve(sc.S, &sc.pr, col);
            for (int i = 0; i < n; ++i) Z[(size_t)i * nb + c] = col[i];
        }
        double *T = malloc((size_t)nb * nb * sizeof(double));
        for (int a = 0; a < nb; ++a) {
            for (int c = 0; c < nb; ++c) {
                double v = sc.Sbb[(size_t)a * nb + c];
                for (int i = 0; i < n; ++i) v -= sc.Spb[(size_t)i * nb + a] * Z[(size_t)i * nb + c];
                T[(size_t)a * nb + c] = v;
            }
            double v = sc.rhs_b[a];
            for (int i = 0; i < n; ++i) v -= sc.Spb[(size_t)i * nb + a] * sc.rhs[i];
            yb[a] = v;
        }
        for (int a = 0; a < nb; ++a)       /* symmetrise against rounding */
            for (int c = 0; c < a; ++c) T[(size_t)a * nb + c] = T[(size_t)c * nb + a] = 0.5 * (T[(size_t)a * nb + c] + T[(size_t)c * nb + a]);
        int bad = dense_spd_solve(T, nb, yb);
        if (!bad)
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < nb; ++c) sc.rhs[i] -= Z[(size_t)i * nb + c] * yb[c];
        free(Z); free(col); free(T);
        if (bad) { free(yb); schur_free(&sc); return -1; }
    }
    double t2 = now_s();
    if (t_solve) *t_solve += t2 - t1;
    /* y_p = rhs ; y_l = C^-1 (g_l - sum W^T y_p); step = -y; delta = scale * step */
    const double *yp = sc.rhs;
    int ok = 1;
#pragma omp parallel for schedule(static) reduction(& : ok)
    for (int j = 0; j < g->L; ++j) {
        double t[6];
        for (int d = 0; d < ld; ++d) t[d] = sc.gl_s[(size_t)ld * j + d];
        for (int64_t e = g->pt_start[j]; e < g->pt_start[j + 1]; ++e) {
            int64_t i = g->pt_obs[e];
            int f = g->free_idx[p->obs_pose[i]];
            if (f < 0) continue;
            const double *Wi = sc.W + (size_t)6 * ld * i, *y = yp + 6 * f;
            for (int d = 0; d < ld; ++d)
                for (int c = 0; c < 6; ++c) t[d] -= Wi[ld * c + d] * y[c];
        }
        if (nb && g->pt_active[j]) {
            const uint32_t mat = p->material_of_point[j];
            const double *V = sc.V + (size_t)ld * 7 * j;
            for (int q = 0; q < 7; ++q) {
                const int c = bcol(g, mat, q);
                if (c < 0) continue;
                for (int d = 0; d < ld; ++d) t[d] -= V[7 * d + q] * yb[c];
            }
        }
        const double *Ci = sc.Ci + (size_t)ld * ld * j;
        for (int a = 0; a < ld; ++a) {
            double y = 0.0;
            for (int q = 0; q < ld; ++q) y += Ci[ld * a + q] * t[q];
            double v = -y * sl[(size_t)ld * j + a];
            if (!g->pt_active[j]) v = 0.0;
            if (!isfinite(v)) ok = 0;
            dl[(size_t)ld * j + a] = v;
        }
    }
    memset(dp, 0, (size_t)g->P * 6 * sizeof(double));
    for (int f = 0; f < g->nfree; ++f)
        for (int c = 0; c < 6; ++c) {
            double v = -yp[6 * f + c] * sp[6 * f + c];
            if (!isfinite(v)) ok = 0;
            dp[6 * g->free_pose[f] + c] = v;
        }
    double dbl[64 + 1];   /* local copy when the caller does not want the border step */
    double *dbo = db ? db : dbl;
    if (nb > 64) ok = 0;
    for (int c = 0; c < nb && c < 64; ++c) {
        double v = -yb[c] * sb[c];
        if (!isfinite(v)) ok = 0;
        dbo[c] = v;
    }
    free(yb);
    schur_free(&sc);
    if (!ok) return -1;
    step_products(p, g, w, dp, dl, nb ? dbo : NULL, model_cost_change, NULL);
    return 0;
}

/* |J d1|^2, |J d2|^2 and (J d1).(J d2) for two arbitrary steps (dp: P*6, dl: L*LD, db: nb or NULL) */
static void jd_products(const orc_problem *p, const graph_t *g, const lin_t *w, const double *dp1, const double *dl1,
                        const double *db1, const double *dp2, const double *dl2, const double *db2, double out[3]) {
    const int nr = g->nr, ld = g->ld;
    double s11 = 0.0, s22 = 0.0, s12 = 0.0;
#pragma omp parallel for reduction(+ : s11, s22, s12) schedule(static)
    for (int64_t i = 0; i < g->N; ++i) {
        const double *a = w->Jp + (size_t)nr * 6 * i, *b = w->Jl + (size_t)nr * ld * i;
        const size_t k = p->obs_pose[i], j = p->obs_point[i];
        const int fr = g->free_idx[k] >= 0;
        for (int m = 0; m < nr; ++m) {
            double j1 = 0.0, j2 = 0.0;
            for (int c = 0; c < ld; ++c) { j1 += b[ld * m + c] * dl1[ld * j + c]; j2 += b[ld * m + c] * dl2[ld * j + c]; }
            if (fr)
                for (int c = 0; c < 6; ++c) { j1 += a[6 * m + c] * dp1[6 * k + c]; j2 += a[6 * m + c] * dp2[6 * k + c]; }
            if (m == 3 && g->nb) {
                const uint32_t mat = p->material_of_point[j];
                for (int q = 0; q < 7; ++q) {
                    const int c = bcol(g, mat, q);
                    if (c >= 0) { j1 += w->Jb[7 * i + q] * db1[c]; j2 += w->Jb[7 * i + q] * db2[c]; }
                }
            }
            s11 += j1 * j1; s22 += j2 * j2; s12 += j1 * j2;
        }
    }
    for (int k = 0; k < g->P; ++k) {
        if (g->free_idx[k] < 0) continue;
        for (int e = g->pf_start[k]; e < g->pf_start[k + 1]; ++e) {
            const double *J = w->pf_J + 36 * (size_t)g->pf_list[e];
            for (int m = 0; m < 6; ++m) {
                double j1 = 0.0, j2 = 0.0;
                for (int c = 0; c < 6; ++c) { j1 += J[6 * m + c] * dp1[6 * (size_t)k + c]; j2 += J[6 * m + c] * dp2[6 * (size_t)k + c]; }
                s11 += j1 * j1; s22 += j2 * j2; s12 += j1 * j2;
            }
        }
    }
    for (uint32_t e = 0; e < p->num_pose_factors; ++e) {
        if (p->pf_type[e] != 2) continue;
        const int other = (int)p->pf_data[18 * (size_t)e + 14], ka = (int)p->pf_pose[e], kb = (int)p->pf_pose[other];
        if (g->free_idx[ka] < 0 || g->free_idx[kb] < 0) continue;
        const double *Ja = w->pf_J + 36 * (size_t)e, *Jb = w->pf_J + 36 * (size_t)other;
        for (int m = 0; m < 6; ++m) {
            double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
            for (int c = 0; c < 6; ++c) {
                a1 += Ja[6 * m + c] * dp1[6 * (size_t)ka + c]; a2 += Ja[6 * m + c] * dp2[6 * (size_t)ka + c];
                b1 += Jb[6 * m + c] * dp1[6 * (size_t)kb + c]; b2 += Jb[6 * m + c] * dp2[6 * (size_t)kb + c];
            }
            s11 += 2.0 * a1 * b1; s22 += 2.0 * a2 * b2; s12 += a1 * b2 + b1 * a2;
        }
    }
    out[0] = s11; out[1] = s22; out[2] = s12;
}

/* Real parts of ALL roots of a polynomial (coefficients highest degree first), as Ceres's
 * FindPolynomialRoots(polynomial, &real, NULL) hands them to its callers [polynomial.cc]: leading zeros
 * removed, closed forms for degree 1 and 2 (the numerically stable quadratic formula), otherwise the
 * eigenvalues of the companion matrix -- restated here with the Aberth-Ehrlich iteration, which
 * converges to the same roots.  Returns the number of roots or -1. */
static int poly_roots_real(const double *coef_in, int ncoef, double *re) {
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
    if (deg > 8) return -1;
    double m[9], bound = 0.0;
    for (int i = 0; i <= deg; ++i) {
        m[i] = c[i] / c[0];
        if (!isfinite(m[i])) return -1;
        if (i && fabs(m[i]) > bound) bound = fabs(m[i]);
    }
    double complex z[8];
    for (int i = 0; i < deg; ++i) z[i] = (1.0 + bound) * cexp(I * (2.0 * 3.14159265358979323846 * i / deg + 0.4));
    int polished = 0;
    for (int it = 0; it < 500; ++it) {
        double worst = 0.0;
        for (int i = 0; i < deg; ++i) {
            double complex pv = m[0], dv = 0.0;
            for (int k = 1; k <= deg; ++k) { dv = dv * z[i] + pv; pv = pv * z[i] + m[k]; }
            if (cabs(pv) == 0.0) continue;
            double complex ratio = pv / dv, sum = 0.0;
            if (cabs(dv) == 0.0) ratio = 1e-3 * (1.0 + cabs(z[i]));
            for (int k = 0; k < deg; ++k)
                if (k != i) sum += 1.0 / (z[i] - z[k]);
            const double complex step = ratio / (1.0 - ratio * sum);
            z[i] -= step;
            const double rel = cabs(step) / (1.0 + cabs(z[i]));
            if (rel > worst) worst = rel;
        }
        if (polished) break;                  /* cubic convergence: one sweep after the 1e-13 sweep reaches rounding level */
        if (worst < 1e-13) polished = 1;
    }
    for (int i = 0; i < deg; ++i) {
        if (!isfinite(creal(z[i]))) return -1;
        re[i] = creal(z[i]);
    }
    return deg;
}

int orc_poly_roots_real(const double *coef, int ncoef, double *re) { return poly_roots_real(coef, ncoef, re); }

/* [Ceres 1.x dogleg_strategy.cc] state kept between iterations */
typedef struct {
    double radius, mu, alpha, dogleg_step_norm, gradient_norm, gn_norm, g_dot_gn;
    int reuse;
    double *gn_p, *gn_l;   /* Gauss-Newton step, unscaled (P*6, L*LD)                    */
    double *v_p, *v_l;     /* s^2 g / D^2: the unscaled image of the scaled gradient / D */
    double gn_b[64 + 1], v_b[64 + 1];   /* the same for the border of free shared blocks */
    /* SUBSPACE_DOGLEG: orthonormal basis of span{gradient_, gauss_newton_step_} expressed in those two
     * vectors (u_i = e[i][0] * gradient_ + e[i][1] * gauss_newton_step_), model g and B */
    int one_dim;
    double e[2][2], sub_g[2], sub_B[3];
} dogleg_t;

/* DoglegStrategy::ComputeTraditionalDoglegStep: coefficients of delta = beta * delta_gn + gamma * v */
static void traditional_dogleg(dogleg_t *dg, double *beta_out, double *gamma_out) {
    double beta, gamma;
    const double r = dg->radius;
    if (dg->gn_norm <= r) {                               /* case 1: GN step inside the region */
        beta = 1.0; gamma = 0.0;
        dg->dogleg_step_norm = dg->gn_norm;
    } else if (dg->gradient_norm * dg->alpha >= r) {      /* case 2: Cauchy point outside */
        beta = 0.0; gamma = -r / dg->gradient_norm;
        dg->dogleg_step_norm = r;
    } else {                                              /* case 3: on the dogleg */
        const double b_dot_a = -dg->alpha * dg->g_dot_gn;
        const double a_sq = pow(dg->alpha * dg->gradient_norm, 2.0);
        const double bma_sq = a_sq - 2.0 * b_dot_a + pow(dg->gn_norm, 2.0);
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma_sq * (r * r - a_sq));
        const double bt = (cc <= 0.0) ? (dd - cc) / bma_sq : (r * r - a_sq) / (dd + cc);
        beta = bt; gamma = -dg->alpha * (1.0 - bt);
        const double a = gamma, b = beta;
        dg->dogleg_step_norm = sqrt(a * a * dg->gradient_norm * dg->gradient_norm + 2.0 * a * b * dg->g_dot_gn +
                                    b * b * dg->gn_norm * dg->gn_norm);
    }
    *beta_out = beta; *gamma_out = gamma;
}

/* DoglegStrategy::ComputeSubspaceModel from the Gram matrices of (gradient_, gauss_newton_step_):
 * inner products in the D-scaled space (pn2, qn2, pq) and of their Jacobian images (jj = |Jv|^2,
 * |J dgn|^2, Jv.J dgn).  ColPivHouseholderQR pivots the longer column first; rank threshold
 * epsilon * min(rows, cols).  Returns 0 if the rank is 0. */
static int subspace_model(dogleg_t *dg, double pn2, double qn2, double pq, const double jj[3]) {
    const int a_is_p = pn2 >= qn2;
    const double an2 = a_is_p ? pn2 : qn2, bn2 = a_is_p ? qn2 : pn2;
    if (!(an2 > 0.0)) return 0;
    const double an = sqrt(an2), proj = pq / an;
    double wn2 = bn2 - proj * proj;
    if (wn2 < 0.0) wn2 = 0.0;
    const double wn = sqrt(wn2);
    const int ia = a_is_p ? 0 : 1, ib = 1 - ia;
    dg->e[0][ia] = 1.0 / an; dg->e[0][ib] = 0.0;
    dg->one_dim = wn <= 2.0 * DBL_EPSILON * an;
    if (dg->one_dim) return 1;
    dg->e[1][ia] = -proj / (an * wn); dg->e[1][ib] = 1.0 / wn;
    for (int i = 0; i < 2; ++i) dg->sub_g[i] = dg->e[i][0] * pn2 + dg->e[i][1] * pq;      /* u_i . gradient_ */
    int n = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = i; j < 2; ++j)
            dg->sub_B[n++] = dg->e[i][0] * dg->e[j][0] * jj[0] + (dg->e[i][0] * dg->e[j][1] + dg->e[i][1] * dg->e[j][0]) * jj[2] +
                             dg->e[i][1] * dg->e[j][1] * jj[1];
    return 1;
}

/* DoglegStrategy::FindMinimumOnTrustRegionBoundary */
static int subspace_boundary_minimum(const dogleg_t *dg, double min_out[2]) {
    const double B00 = dg->sub_B[0], B01 = dg->sub_B[1], B11 = dg->sub_B[2], g0 = dg->sub_g[0], g1 = dg->sub_g[1];
    const double detB = B00 * B11 - B01 * B01, trB = B00 + B11, r2 = dg->radius * dg->radius;
    /* B_adj = [B11 -B01; -B01 B00] */
    const double ag0 = B11 * g0 - B01 * g1, ag1 = -B01 * g0 + B00 * g1;
    double poly[5];
    poly[0] = r2;
    poly[1] = 2.0 * r2 * trB;
    poly[2] = r2 * (trB * trB + 2.0 * detB) - (g0 * g0 + g1 * g1);
    poly[3] = -2.0 * ((g0 * ag0 + g1 * ag1) - r2 * detB * trB);
    poly[4] = r2 * detB * detB - (ag0 * ag0 + ag1 * ag1);
    double roots[8];
    const int nroots = poly_roots_real(poly, 5, roots);
    min_out[0] = min_out[1] = 0.0;
    if (nroots < 0) return 0;
    double best = DBL_MAX;
    int found = 0;
    for (int i = 0; i < nroots; ++i) {
        /* x = -(B + y I)^-1 g, partial-pivot LU of the 2x2 */
        double a = B00 + roots[i], b = B01, c = B01, d = B11 + roots[i], r0 = g0, r1 = g1;
        if (fabs(c) > fabs(a)) { double t; t = a; a = c; c = t; t = b; b = d; d = t; t = r0; r0 = r1; r1 = t; }
        const double l = c / a, u = d - l * b;
        const double x1 = (r1 - l * r0) / u, x0 = (r0 - b * x1) / a;
        const double x[2] = {-x0, -x1};
        const double nx = sqrt(x[0] * x[0] + x[1] * x[1]);
        if (nx > 0) {
            const double sx[2] = {dg->radius / nx * x[0], dg->radius / nx * x[1]};
            const double f = 0.5 * (sx[0] * (B00 * sx[0] + B01 * sx[1]) + sx[1] * (B01 * sx[0] + B11 * sx[1])) + g0 * sx[0] + g1 * sx[1];
            found = 1;
            if (f < best) { best = f; min_out[0] = x[0]; min_out[1] = x[1]; }
        }
    }
    return found;
}

/* One dogleg step.  In Ceres the strategy sees the Jacobi-scaled Jacobian J_s = J diag(s):
 *   D^2 = clamp(diag(J_s^T J_s)), gradient_ = J_s^T r ./ D, alpha = |gradient_|^2 / |J_s (gradient_ ./ D)|^2,
 *   Gauss-Newton: (J_s^T J_s + mu D^2) y = J_s^T r, gauss_newton_step_ = -D .* y,
 *   step = interpolation in the D-scaled space, ./ D, then .* s by the minimiser.
 * In unscaled coordinates: delta = beta * delta_gn + gamma * v with v = s^2 g / D^2, for the traditional
 * and for the subspace dogleg (its basis vectors are combinations of gradient_ and gauss_newton_step_). */
static int dogleg_step(const orc_problem *p, const graph_t *g, const lin_t *w, const double *sp,
                       const double *sl, const double *sb, const orc_options *o, dogleg_t *dg, double *dp, double *dl,
                       double *db, double *mcc, double *t_schur, double *t_solve) {
    const int nf = g->nfree, L = g->L, ld = g->ld, nb = g->nb;
    const int subspace = o->dogleg_type == 1;
    if (!dg->reuse) {
        dg->reuse = 1;
        /* Gauss-Newton step with the regulariser mu * D^2: same damped system as LM with 1/radius = mu */
        double mcc_gn;
        if (lm_step(p, g, w, sp, sl, sb, 1.0 / dg->mu, o, dg->gn_p, dg->gn_l, dg->gn_b, &mcc_gn, t_schur, t_solve)) return -1;
        double gsq = 0.0, nsq = 0.0, dot = 0.0;
        memset(dg->v_p, 0, (size_t)g->P * 6 * sizeof(double));
        for (int f = 0; f < nf; ++f)
            for (int c = 0; c < 6; ++c) {
                const int k = g->free_pose[f];
                const double s = sp[6 * f + c], gq = w->g_p[6 * f + c], gn = dg->gn_p[6 * k + c];
                const double D2 = fmin(fmax(w->sq_p[6 * f + c] * s * s, o->min_lm_diagonal), o->max_lm_diagonal);
                gsq += s * s * gq * gq / D2;
                nsq += D2 * gn * gn / (s * s);
                dot += gq * gn;
                dg->v_p[6 * k + c] = s * s * gq / D2;
            }
        for (int j = 0; j < L; ++j)
            for (int c = 0; c < ld; ++c) {
                const size_t ix = (size_t)ld * j + c;
                const double s = sl[ix], gq = w->g_l[ix], gn = dg->gn_l[ix];
                const double D2 = fmin(fmax(w->sq_l[ix] * s * s, o->min_lm_diagonal), o->max_lm_diagonal);
                if (!g->pt_active[j]) { dg->v_l[ix] = 0.0; continue; }
                gsq += s * s * gq * gq / D2;
                nsq += D2 * gn * gn / (s * s);
                dot += gq * gn;
                dg->v_l[ix] = s * s * gq / D2;
            }
        for (int c = 0; c < nb; ++c) {
            const double s = sb[c], gq = w->g_b[c], gn = dg->gn_b[c];
            const double D2 = fmin(fmax(w->sq_b[c] * s * s, o->min_lm_diagonal), o->max_lm_diagonal);
            gsq += s * s * gq * gq / D2;
            nsq += D2 * gn * gn / (s * s);
            dot += gq * gn;
            dg->v_b[c] = s * s * gq / D2;
        }
        double jj[3];
        jd_products(p, g, w, dg->v_p, dg->v_l, dg->v_b, dg->gn_p, dg->gn_l, dg->gn_b, jj);
        dg->gradient_norm = sqrt(gsq);
        dg->gn_norm = sqrt(nsq);
        dg->g_dot_gn = dot;           /* gradient_ . gauss_newton_step_ */
        dg->alpha = gsq / jj[0];      /* ComputeCauchyPoint */
        if (subspace && !subspace_model(dg, gsq, nsq, dot, jj)) return -1;   /* ComputeSubspaceModel */
    }
    double beta, gamma;
    if (!subspace) {
        traditional_dogleg(dg, &beta, &gamma);
    } else {
        /* ComputeSubspaceDoglegStep */
        double m2[2];
        if (dg->gn_norm <= dg->radius) {
            beta = 1.0; gamma = 0.0;
            dg->dogleg_step_norm = dg->gn_norm;
        } else if (dg->one_dim) {
            beta = 0.0; gamma = -dg->radius / dg->gradient_norm;
            dg->dogleg_step_norm = dg->radius;
        } else if (!subspace_boundary_minimum(dg, m2)) {
            traditional_dogleg(dg, &beta, &gamma);   /* "Taking traditional dogleg step instead" */
        } else {
            gamma = m2[0] * dg->e[0][0] + m2[1] * dg->e[1][0];   /* coefficient of gradient_ -> v */
            beta = m2[0] * dg->e[0][1] + m2[1] * dg->e[1][1];    /* coefficient of gauss_newton_step_ */
            dg->dogleg_step_norm = dg->radius;
        }
    }
    for (int i = 0; i < g->P * 6; ++i) dp[i] = beta * dg->gn_p[i] + gamma * dg->v_p[i];
    for (int i = 0; i < L * ld; ++i) dl[i] = beta * dg->gn_l[i] + gamma * dg->v_l[i];
    for (int c = 0; c < nb; ++c) db[c] = beta * dg->gn_b[c] + gamma * dg->v_b[c];
    step_products(p, g, w, dp, dl, nb ? db : NULL, mcc, NULL);
    return 0;
}

int orc_border_size(const orc_problem *p) {
    graph_t g;
    graph_build(p, &g);
    int nb = g.nb;
    graph_free(&g);
    return nb;
}

int orc_lm_step(const orc_problem *p, double radius, const orc_options *o, double *delta_p,
                double *delta_l, double *model_cost_change) {
    return orc_lm_step_border(p, radius, o, delta_p, delta_l, NULL, model_cost_change);
}

int orc_lm_step_border(const orc_problem *p, double radius, const orc_options *o, double *delta_p,
                       double *delta_l, double *delta_b, double *model_cost_change) {
    set_threads(o->num_threads);
    graph_t g;
    graph_build(p, &g);
    lin_t w;
    lin_alloc(&w, &g);
    linearize(p, &g, p->poses, p->points, p->normals, NULL, &w);
    double *sp = malloc((size_t)(g.nfree > 0 ? g.nfree : 1) * 6 * sizeof(double));
    double *sl = malloc((size_t)(g.L > 0 ? g.L : 1) * g.ld * sizeof(double));
    double sb[64 + 1];
    if (g.nb > 64) { lin_free(&w); graph_free(&g); free(sp); free(sl); return -1; }
    jacobi_scale(&g, &w, o->jacobi_scaling, sp, sl, sb);
    int rc = lm_step(p, &g, &w, sp, sl, sb, radius, o, delta_p, delta_l, delta_b, model_cost_change, NULL, NULL);
    free(sp); free(sl);
    lin_free(&w);
    graph_free(&g);
    return rc;
}

int orc_reduced_system(const orc_problem *p, double radius, const orc_options *o, double *S,
                       double *rhs, int32_t *free_pose_index) {
    set_threads(o->num_threads);
    graph_t g;
    graph_build(p, &g);
    lin_t w;
    lin_alloc(&w, &g);
    linearize(p, &g, p->poses, p->points, p->normals, NULL, &w);
    double *sp = malloc((size_t)(g.nfree > 0 ? g.nfree : 1) * 6 * sizeof(double));
    double *sl = malloc((size_t)(g.L > 0 ? g.L : 1) * g.ld * sizeof(double));
    double sb[64 + 1];
    if (g.nb > 64) { lin_free(&w); graph_free(&g); free(sp); free(sl); return -1; }
    jacobi_scale(&g, &w, o->jacobi_scaling, sp, sl, sb);
    schur_t sc;
    memset(&sc, 0, sizeof sc);
    int rc = build_reduced(p, &g, &w, sp, sl, sb, radius, o, &sc);
    if (!rc) {
        /* un-scale: S_unscaled = diag(1/s) S_s diag(1/s), rhs_unscaled = -(1/s) rhs_s
         * so that S_unscaled * delta = rhs_unscaled (delta = -s .* y).  With free shared blocks the
         * system is the (n + nb) arrowhead [S_pp S_pb; S_pb^T S_bb], leading dimension n + nb. */
        int n = sc.n, nb = sc.nb, nt = n + nb;
        for (int i = 0; i < nt; ++i)
            for (int j = 0; j < nt; ++j) S[(size_t)i * nt + j] = 0.0;
        for (int i = 0; i < n; ++i) rhs[i] = -sc.rhs[i] / sp[i];
        for (int i = 0; i < n; ++i) {
            for (int j = sc.pr.first[i]; j <= i; ++j) {
                double v = sc.S[sc.pr.rp[i] + (size_t)(j - sc.pr.first[i])] / (sp[i] * sp[j]);
                S[(size_t)i * nt + j] = v;
                S[(size_t)j * nt + i] = v;
            }
        }
        for (int c = 0; c < nb; ++c) {
            rhs[n + c] = -sc.rhs_b[c] / sb[c];
            for (int i = 0; i < n; ++i) {
                double v = sc.Spb[(size_t)i * nb + c] / (sp[i] * sb[c]);
                S[(size_t)i * nt + n + c] = v;
                S[(size_t)(n + c) * nt + i] = v;
            }
            for (int c2 = 0; c2 < nb; ++c2) S[(size_t)(n + c) * nt + n + c2] = sc.Sbb[(size_t)c * nb + c2] / (sb[c] * sb[c2]);
        }
        for (int k = 0; k < g.P; ++k) free_pose_index[k] = g.free_idx[k];
    }
    schur_free(&sc);
    free(sp); free(sl);
    lin_free(&w);
    graph_free(&g);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* trust-region minimiser [Ceres 1.13/1.14 trust_region_minimizer.cc]         */
/* ------------------------------------------------------------------------ */

/* Evaluator::Plus: SE3Perturbation on free poses, Euclidean on the active points,
 * UnitVectorPerturbation on their normals (perturbations.hpp:87-103).  The landmark step is
 * LD wide: [d position | d normal]. */
static void plus_all(const orc_problem *p, const graph_t *g, const double *poses, const double *points, const double *normals,
                     const double *sh, const double *dp, const double *dl, const double *db, double *poses_out,
                     double *points_out, double *normals_out, double *sh_out) {
    const int ld = g->ld;
    if (sh_out) {
        /* shared blocks: Euclidean Plus, UnitVectorPerturbation on a directional light
         * (dataset_ba_phong.cpp:201-204) */
        const int M = g->M;
        memcpy(sh_out, sh, (size_t)(3 + 4 * M) * sizeof(double));
        if (g->b_light >= 0) {
            if (p->light_type == ORC_DIRECTIONAL_LIGHT) orc_unit_vector_plus(sh, db + g->b_light, sh_out);
            else for (int c = 0; c < 3; ++c) sh_out[c] = sh[c] + db[g->b_light + c];
        }
        if (g->b_phong >= 0) for (int c = 0; c < 3 * M; ++c) sh_out[3 + c] = sh[3 + c] + db[g->b_phong + c];
        if (g->b_tex >= 0) for (int c = 0; c < M; ++c) sh_out[3 + 3 * M + c] = sh[3 + 3 * M + c] + db[g->b_tex + c];
        if (p->use_bounds) {
            /* ParameterBlock::Plus projects onto the box [Ceres parameter_block.h]; the driver's bounds:
             * ka, ks in [0,1], alpha >= 1, kd in [0,1] (dataset_ba_phong.cpp:143-181) */
            if (g->b_phong >= 0)
                for (int m = 0; m < M; ++m) {
                    double *q = sh_out + 3 + 3 * m;
                    q[0] = fmin(fmax(q[0], 0.0), 1.0);
                    q[1] = fmin(fmax(q[1], 0.0), 1.0);
                    q[2] = fmax(q[2], 1.0);
                }
            if (g->b_tex >= 0)
                for (int m = 0; m < M; ++m) sh_out[3 + 3 * M + m] = fmin(fmax(sh_out[3 + 3 * M + m], 0.0), 1.0);
        }
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->P; ++k) {
        if (g->free_idx[k] >= 0) orc_se3_plus(poses + 12 * k, dp + 6 * k, poses_out + 12 * k);
        else memcpy(poses_out + 12 * k, poses + 12 * k, 12 * sizeof(double));
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < g->L; ++j) {
        for (int a = 0; a < 3; ++a)
            points_out[3 * j + a] = g->pt_active[j] ? points[3 * j + a] + dl[(size_t)ld * j + a] : points[3 * j + a];
        if (ld == 6) {
            if (g->pt_active[j]) orc_unit_vector_plus(normals + 3 * j, dl + (size_t)ld * j + 3, normals_out + 3 * j);
            else memcpy(normals_out + 3 * j, normals + 3 * j, 3 * sizeof(double));
        }
    }
}

/* ambient-space norms over the reduced program's parameter blocks */
static double x_sq_diff(const graph_t *g, const double *pa, const double *qa, const double *na, const double *sa,
                        const double *pb, const double *qb, const double *nb, const double *sbb, double *max_abs) {
    double s = 0.0, m = 0.0;
    if (g->nb) {
        const int M = g->M;
        const int lo[3] = {0, 3, 3 + 3 * M}, hi[3] = {3, 3 + 3 * M, 3 + 4 * M}, on[3] = {g->b_light >= 0, g->b_phong >= 0, g->b_tex >= 0};
        for (int b = 0; b < 3; ++b)
            if (on[b])
                for (int c = lo[b]; c < hi[b]; ++c) {
                    double d = sa[c] - (sbb ? sbb[c] : 0.0);
                    s += d * d;
                    if (fabs(d) > m) m = fabs(d);
                }
    }
    for (int f = 0; f < g->nfree; ++f) {
        int k = g->free_pose[f];
        for (int c = 0; c < 12; ++c) {
            double d = pa[12 * k + c] - (pb ? pb[12 * k + c] : 0.0);
            s += d * d;
            if (fabs(d) > m) m = fabs(d);
        }
    }
    for (int j = 0; j < g->L; ++j) {
        if (!g->pt_active[j]) continue;
        for (int c = 0; c < 3 && !g->pos_const; ++c) {
            double d = qa[3 * j + c] - (qb ? qb[3 * j + c] : 0.0);
            s += d * d;
            if (fabs(d) > m) m = fabs(d);
        }
        if (g->ld == 6)
            for (int c = 0; c < 3; ++c) {
                double d = na[3 * j + c] - (nb ? nb[3 * j + c] : 0.0);
                s += d * d;
                if (fabs(d) > m) m = fabs(d);
            }
    }
    if (max_abs) *max_abs = m;
    return s;
}

/* TrustRegionStepEvaluator (Conn, Gould & Toint alg. 10.1.2) */
typedef struct {
    int max_nonmono, num_nonmono;
    double minimum_cost, current_cost, reference_cost, candidate_cost;
    double acc_reference_mcc, acc_candidate_mcc;
} step_eval_t;

static void se_init(step_eval_t *e, double cost, int max_nonmono) {
    e->max_nonmono = max_nonmono; e->num_nonmono = 0;
    e->minimum_cost = e->current_cost = e->reference_cost = e->candidate_cost = cost;
    e->acc_reference_mcc = e->acc_candidate_mcc = 0.0;
}
static double se_quality(const step_eval_t *e, double cost, double mcc) {
    double rd = (e->current_cost - cost) / mcc;
    double hrd = (e->reference_cost - cost) / (e->acc_reference_mcc + mcc);
    return rd > hrd ? rd : hrd;
}
static void se_accepted(step_eval_t *e, double cost, double mcc) {
    e->current_cost = cost;
    e->acc_candidate_mcc += mcc;
    e->acc_reference_mcc += mcc;
    if (e->current_cost < e->minimum_cost) {
        e->minimum_cost = e->current_cost;
        e->num_nonmono = 0;
        e->candidate_cost = e->current_cost;
        e->acc_candidate_mcc = 0.0;
    } else {
        ++e->num_nonmono;
        if (e->current_cost > e->candidate_cost) {
            e->candidate_cost = e->current_cost;
            e->acc_candidate_mcc = 0.0;
        }
    }
    if (e->num_nonmono == e->max_nonmono) {
        e->reference_cost = e->candidate_cost;
        e->acc_reference_mcc = e->acc_candidate_mcc;
    }
}

static void log_push(orc_iteration_log *log, orc_summary *s, double cost, double cost_change,
                     double gmax, double step_norm, double rd, double radius, int ok) {
    int i = s->num_iterations++;
    if (!log || i >= log->capacity) return;
    log->cost[i] = cost; log->cost_change[i] = cost_change; log->gradient_max_norm[i] = gmax;
    log->step_norm[i] = step_norm; log->relative_decrease[i] = rd;
    log->trust_region_radius[i] = radius; log->step_is_successful[i] = ok;
}

/* ---- projected Armijo line search of the bounds-constrained trust-region loop ----------------
 * [Ceres 1.13 TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch::DoSearch with the Solver::Options
 * defaults: CUBIC interpolation, sufficient decrease 1e-4, step contraction in [1e-3, 0.6], 20
 * iterations, min step size 1e-9; line_search.cc, polynomial.cc] */
typedef struct { double x, value, gradient; int value_ok, gradient_ok; } ls_sample_t;

static double poly_eval(const double *c, int n, double x) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = v * x + c[i];
    return v;
}

/* FindInterpolatingPolynomial: full-pivot elimination of the Vandermonde-type system */
static int ls_fit(const ls_sample_t *smp, int ns, double *coef) {
    int nc = 0;
    for (int i = 0; i < ns; ++i) nc += smp[i].value_ok + smp[i].gradient_ok;
    const int deg = nc - 1;
    double A[6][7];
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (smp[i].value_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = pow(smp[i].x, deg - j);
            A[row][nc] = smp[i].value;
            ++row;
        }
        if (smp[i].gradient_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = j < deg ? (deg - j) * pow(smp[i].x, deg - j - 1) : 0.0;
            A[row][nc] = smp[i].gradient;
            ++row;
        }
    }
    int perm[6];
    for (int i = 0; i < nc; ++i) perm[i] = i;
    for (int k = 0; k < nc; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int i = k; i < nc; ++i)
            for (int j = k; j < nc; ++j)
                if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); pr = i; pc = j; }
        if (!(best > 0.0)) { for (int i = k; i < nc; ++i) A[i][nc] = 0.0; break; }
        for (int j = 0; j <= nc; ++j) { double t = A[k][j]; A[k][j] = A[pr][j]; A[pr][j] = t; }
        for (int i = 0; i < nc; ++i) { double t = A[i][k]; A[i][k] = A[i][pc]; A[i][pc] = t; }
        { int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        for (int i = k + 1; i < nc; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= nc; ++j) A[i][j] -= f * A[k][j];
        }
    }
    double y[6];
    for (int i = nc - 1; i >= 0; --i) {
        double v = A[i][nc];
        for (int j = i + 1; j < nc; ++j) v -= A[i][j] * y[j];
        y[i] = A[i][i] != 0.0 ? v / A[i][i] : 0.0;
    }
    for (int i = 0; i < nc; ++i) coef[perm[i]] = y[i];
    return nc;
}

/* MinimizeInterpolatingPolynomial */
static double ls_minimize(const ls_sample_t *smp, int ns, double x_min, double x_max) {
    double coef[6];
    const int nc = ls_fit(smp, ns, coef);
    double best_x = 0.5 * (x_min + x_max), best_v = poly_eval(coef, nc, best_x), v;
    if ((v = poly_eval(coef, nc, x_min)) < best_v) { best_v = v; best_x = x_min; }
    if ((v = poly_eval(coef, nc, x_max)) < best_v) { best_v = v; best_x = x_max; }
    if (nc > 2) {
        double der[6], roots[8];
        for (int i = 0; i < nc - 1; ++i) der[i] = (nc - 1 - i) * coef[i];
        const int nr = poly_roots_real(der, nc - 1, roots);
        for (int i = 0; i < nr; ++i) {
            if (roots[i] < x_min || roots[i] > x_max) continue;
            if ((v = poly_eval(coef, nc, roots[i])) < best_v) { best_v = v; best_x = roots[i]; }
        }
    }
    for (int i = 0; i < ns; ++i)
        if (smp[i].value_ok && smp[i].x >= x_min && smp[i].x <= x_max && smp[i].value < best_v) { best_v = smp[i].value; best_x = smp[i].x; }
    return best_x;
}

/* LineSearch::InterpolatingPolynomialMinimizingStepSize, CUBIC */
static double ls_next_step(const ls_sample_t *lower, const ls_sample_t *prev, const ls_sample_t *cur, double min_step, double max_step) {
    if (!cur->value_ok) return fmin(fmax(cur->x * 0.5, min_step), max_step);
    ls_sample_t smp[3];
    int ns = 0;
    smp[ns++] = *lower;
    smp[ns++] = *cur;
    if (prev->value_ok) smp[ns++] = *prev;
    return ls_minimize(smp, ns, min_step, max_step);
}

/* ArmijoLineSearch::DoSearch as a resumable state machine: the caller evaluates phi(x) = cost(Plus(x0,
 * x * delta)) and phi'(x) = delta . gradient(x) at the requested step and feeds the sample back. */
typedef struct {
    ls_sample_t initial, previous, current;
    double dir_max_norm;
    int num_iterations, done, success;
    double optimal_step;
} armijo_t;

static void armijo_begin(armijo_t *a, double initial_cost, double initial_gradient, double dir_max_norm) {
    memset(a, 0, sizeof *a);
    a->initial.x = 0.0; a->initial.value = initial_cost; a->initial.gradient = initial_gradient;
    a->initial.value_ok = a->initial.gradient_ok = 1;
    a->current.x = 1.0;                 /* step_size_estimate = 1.0 */
    a->dir_max_norm = dir_max_norm;
}
/* feed the evaluation of a->current.x; afterwards either a->done or a->current.x is the next request */
static void armijo_feed(armijo_t *a, double value, double gradient) {
    const double sufficient_decrease = 1e-4, max_step_contraction = 1e-3, min_step_contraction = 0.6, min_step_size = 1e-9;
    const int max_num_iterations = 20;
    a->current.value = value; a->current.gradient = gradient;
    a->current.value_ok = isfinite(value);
    a->current.gradient_ok = a->current.value_ok && isfinite(gradient);
    if (a->current.value_ok && !(a->current.value > a->initial.value + sufficient_decrease * a->initial.gradient * a->current.x)) {
        a->optimal_step = a->current.x; a->success = 1; a->done = 1;
        return;
    }
    ++a->num_iterations;
    if (a->num_iterations >= max_num_iterations) { a->done = 1; return; }
    const double step = ls_next_step(&a->initial, &a->previous, &a->current, max_step_contraction * a->current.x,
                                     min_step_contraction * a->current.x);
    if (step * a->dir_max_norm < min_step_size) { a->done = 1; return; }
    a->previous = a->current;
    a->current.x = step;
    a->current.value_ok = a->current.gradient_ok = 0;
}

int orc_armijo_trace(const double *values, const double *gradients, int n, double initial_cost, double initial_gradient,
                     double dir_max_norm, double *steps_out, double *optimal_step) {
    /* test hook: replays the state machine on a given sequence of evaluations; returns #requests made */
    armijo_t a;
    armijo_begin(&a, initial_cost, initial_gradient, dir_max_norm);
    int k = 0;
    while (!a.done && k < n) {
        steps_out[k] = a.current.x;
        armijo_feed(&a, values[k], gradients[k]);
        ++k;
    }
    *optimal_step = a.success ? a.optimal_step : -1.0;
    return k;
}

int orc_solve(orc_problem *p, const orc_options *o, orc_summary *s, orc_iteration_log *log) {
    set_threads(o->num_threads);
    memset(s, 0, sizeof *s);
    double t_start = now_s();
    graph_t g;
    graph_build(p, &g);
    const int P = g.P, L = g.L, ld = g.ld, ph = (ld == 6);
    lin_t w;
    lin_alloc(&w, &g);
    size_t szP = (size_t)(P > 0 ? P : 1), szL = (size_t)(L > 0 ? L : 1);
    double *x_pose = malloc(szP * 12 * sizeof(double)), *x_pt = malloc(szL * 3 * sizeof(double)), *x_n = malloc(szL * 3 * sizeof(double));
    double *c_pose = malloc(szP * 12 * sizeof(double)), *c_pt = malloc(szL * 3 * sizeof(double)), *c_n = malloc(szL * 3 * sizeof(double));
    double *best_pose = malloc(szP * 12 * sizeof(double)), *best_pt = malloc(szL * 3 * sizeof(double)), *best_n = malloc(szL * 3 * sizeof(double));
    double *dp = calloc(szP * 6, sizeof(double)), *dl = calloc(szL * ld, sizeof(double));
    double *ngp = calloc(szP * 6, sizeof(double)), *ngl = calloc(szL * ld, sizeof(double));
    double *sp = malloc((size_t)(g.nfree > 0 ? g.nfree : 1) * 6 * sizeof(double));
    double *sl = malloc(szL * ld * sizeof(double));
    /* shared blocks (light / materials / textures): packed ambient state, border step, scale */
    const int nb = g.nb, nsh = ph ? shared_size(p) : 1;
    double *x_sh = calloc((size_t)nsh, sizeof(double)), *c_sh = calloc((size_t)nsh, sizeof(double)), *best_sh = calloc((size_t)nsh, sizeof(double));
    double db[64 + 1] = {0}, ngb[64 + 1] = {0}, sb[64 + 1];
    if (nb > 64) {   /* unsupported here: report FAILURE */
        s->termination_type = ORC_FAILURE;
        free(x_pose); free(x_pt); free(x_n); free(c_pose); free(c_pt); free(c_n); free(best_pose); free(best_pt); free(best_n);
        free(dp); free(dl); free(ngp); free(ngl); free(sp); free(sl); free(x_sh); free(c_sh); free(best_sh);
        lin_free(&w); graph_free(&g);
        return -1;
    }
    if (ph) { shared_pack(p, x_sh); memcpy(c_sh, x_sh, (size_t)nsh * sizeof(double)); memcpy(best_sh, x_sh, (size_t)nsh * sizeof(double)); }
    memcpy(x_pose, p->poses, (size_t)P * 12 * sizeof(double));
    memcpy(x_pt, p->points, (size_t)L * 3 * sizeof(double));
    if (ph) memcpy(x_n, p->normals, (size_t)L * 3 * sizeof(double));
    else memset(x_n, 0, szL * 3 * sizeof(double));
    memcpy(best_pose, x_pose, (size_t)P * 12 * sizeof(double));
    memcpy(best_pt, x_pt, (size_t)L * 3 * sizeof(double));
    memcpy(best_n, x_n, szL * 3 * sizeof(double));

    /* ---- IterationZero ---- */
    const int constrained = ph && p->use_bounds && (g.b_phong >= 0 || g.b_tex >= 0);
    lin_t w2;                 /* line-search evaluations keep the linearisation at x intact */
    memset(&w2, 0, sizeof w2);
    double *ls_pose = NULL, *ls_pt = NULL, *ls_n = NULL, *ls_sh = NULL, *sdp = NULL, *sdl = NULL;
    if (constrained) {
        lin_alloc(&w2, &g);
        ls_pose = malloc(szP * 12 * sizeof(double)); ls_pt = malloc(szL * 3 * sizeof(double)); ls_n = malloc(szL * 3 * sizeof(double));
        ls_sh = calloc((size_t)nsh, sizeof(double));
        sdp = calloc(szP * 6, sizeof(double)); sdl = calloc(szL * ld, sizeof(double));
        /* "x = Plus(x, 0)": project the initial point onto the feasible set */
        plus_all(p, &g, x_pose, x_pt, x_n, x_sh, dp, dl, db, c_pose, c_pt, c_n, c_sh);
        memcpy(x_sh, c_sh, (size_t)nsh * sizeof(double));
        memcpy(best_sh, x_sh, (size_t)nsh * sizeof(double));
    }
    double t0 = now_s();
    linearize(p, &g, x_pose, x_pt, x_n, ph ? x_sh : NULL, &w);
    s->linearize_time_s += now_s() - t0;
    jacobi_scale(&g, &w, o->jacobi_scaling, sp, sl, sb);
    double x_cost = w.cost, minimum_cost = x_cost;
    double x_norm = sqrt(x_sq_diff(&g, x_pose, x_pt, x_n, x_sh, NULL, NULL, NULL, NULL, NULL));
    s->initial_cost = x_cost;
    /* projected gradient: |x - Plus(x, -g)|_inf [EvaluateGradientAndJacobian] */
    double gmax;
#define GRADIENT_MAX_NORM()                                                              \
    do {                                                                                 \
        for (int f = 0; f < g.nfree; ++f)                                                \
            for (int c = 0; c < 6; ++c) ngp[6 * g.free_pose[f] + c] = -w.g_p[6 * f + c]; \
        for (int i = 0; i < ld * L; ++i) ngl[i] = -w.g_l[i];                             \
        for (int i = 0; i < nb; ++i) ngb[i] = -w.g_b[i];                                 \
        plus_all(p, &g, x_pose, x_pt, x_n, x_sh, ngp, ngl, ngb, c_pose, c_pt, c_n, nb ? c_sh : NULL); \
        x_sq_diff(&g, x_pose, x_pt, x_n, x_sh, c_pose, c_pt, c_n, c_sh, &gmax);          \
    } while (0)
    GRADIENT_MAX_NORM();
    double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
    const int dogleg = o->trust_region_strategy_type == 1;
    dogleg_t dg;
    memset(&dg, 0, sizeof dg);
    dg.radius = radius; dg.mu = 1e-8;      /* DoglegStrategy: mu_ = min_mu_ = 1e-8, max 1.0, factor 10 */
    if (dogleg) {
        dg.gn_p = calloc(szP * 6, sizeof(double)); dg.gn_l = calloc(szL * ld, sizeof(double));
        dg.v_p = calloc(szP * 6, sizeof(double)); dg.v_l = calloc(szL * ld, sizeof(double));
    }
    step_eval_t se;
    se_init(&se, x_cost, o->use_nonmonotonic_steps ? o->max_consecutive_nonmonotonic_steps : 0);
    int iteration = 0, num_invalid = 0;
    int term = -1;
    /* iteration 0 is recorded with step_is_successful = false and therefore counts
     * as an "unsuccessful step" in the Ceres 1.13 summary */
    log_push(log, s, x_cost, 0.0, gmax, 0.0, 0.0, radius, 0);
    s->num_unsuccessful_steps = 1;
    int last_successful = 0;

    for (;;) {
        /* ---- FinalizeIterationAndCheckIfMinimizerCanContinue ---- */
        if (last_successful && x_cost < minimum_cost) {
            minimum_cost = x_cost;
            memcpy(best_pose, x_pose, (size_t)P * 12 * sizeof(double));
            memcpy(best_pt, x_pt, (size_t)L * 3 * sizeof(double));
            memcpy(best_n, x_n, (size_t)L * 3 * sizeof(double));
            memcpy(best_sh, x_sh, (size_t)nsh * sizeof(double));
        }
        if (iteration >= o->max_num_iterations) { term = ORC_NO_CONVERGENCE; break; }
        if (gmax <= o->gradient_tolerance) { term = ORC_CONVERGENCE; break; }
        if (radius <= o->min_trust_region_radius) { term = ORC_CONVERGENCE; break; }
        ++iteration;
        last_successful = 0;

        /* ---- ComputeTrustRegionStep ---- */
        double mcc = 0.0;
        int rc = dogleg ? dogleg_step(p, &g, &w, sp, sl, sb, o, &dg, dp, dl, db, &mcc, &s->schur_time_s, &s->solve_time_s)
                        : lm_step(p, &g, &w, sp, sl, sb, radius, o, dp, dl, db, &mcc, &s->schur_time_s, &s->solve_time_s);
        int step_is_valid = (rc == 0) && (mcc > 0.0);
        if (!step_is_valid) {
            /* HandleInvalidStep */
            if (++num_invalid >= o->max_num_consecutive_invalid_steps) {
                term = ORC_FAILURE;
                log_push(log, s, x_cost, 0.0, gmax, 0.0, 0.0, radius, 0);
                ++s->num_unsuccessful_steps;
                break;
            }
            if (dogleg) {               /* DoglegStrategy::StepIsInvalid */
                dg.mu *= 10.0; dg.reuse = 0;
            } else {
                radius /= decrease_factor;  /* LM: StepIsInvalid() == StepRejected */
                decrease_factor *= 2.0;
            }
            log_push(log, s, x_cost, 0.0, gmax, 0.0, 0.0, radius, 0);
            ++s->num_unsuccessful_steps;
            continue;
        }
        num_invalid = 0;

        if (constrained) {
            /* ---- DoLineSearch(x, gradient, cost, &delta): projected Armijo search along delta ---- */
            double g0 = 0.0, dmax = 0.0;
            for (int f = 0; f < g.nfree; ++f)
                for (int c = 0; c < 6; ++c) {
                    const double dv = dp[6 * g.free_pose[f] + c];
                    g0 += w.g_p[6 * f + c] * dv;
                    if (fabs(dv) > dmax) dmax = fabs(dv);
                }
            for (int j = 0; j < L; ++j) {
                if (!g.pt_active[j]) continue;
                for (int c = 0; c < ld; ++c) {
                    g0 += w.g_l[(size_t)ld * j + c] * dl[(size_t)ld * j + c];
                    if (fabs(dl[(size_t)ld * j + c]) > dmax) dmax = fabs(dl[(size_t)ld * j + c]);
                }
            }
            for (int c = 0; c < nb; ++c) { g0 += w.g_b[c] * db[c]; if (fabs(db[c]) > dmax) dmax = fabs(db[c]); }
            armijo_t arm;
            armijo_begin(&arm, x_cost, g0, dmax);
            while (!arm.done) {
                const double a = arm.current.x;
                double sdb[64 + 1];
                for (int i = 0; i < P * 6; ++i) sdp[i] = a * dp[i];
                for (int i = 0; i < L * ld; ++i) sdl[i] = a * dl[i];
                for (int c = 0; c < nb; ++c) sdb[c] = a * db[c];
                plus_all(p, &g, x_pose, x_pt, x_n, x_sh, sdp, sdl, sdb, ls_pose, ls_pt, ls_n, ls_sh);
                linearize(p, &g, ls_pose, ls_pt, ls_n, ls_sh, &w2);
                double gd = 0.0;      /* direction . gradient at the trial point (its own tangent space) */
                for (int f = 0; f < g.nfree; ++f)
                    for (int c = 0; c < 6; ++c) gd += w2.g_p[6 * f + c] * dp[6 * g.free_pose[f] + c];
                for (int j = 0; j < L; ++j)
                    if (g.pt_active[j])
                        for (int c = 0; c < ld; ++c) gd += w2.g_l[(size_t)ld * j + c] * dl[(size_t)ld * j + c];
                for (int c = 0; c < nb; ++c) gd += w2.g_b[c] * db[c];
                armijo_feed(&arm, w2.cost, gd);
                ++s->num_line_search_steps;
            }
            if (arm.success) {
                for (int i = 0; i < P * 6; ++i) dp[i] *= arm.optimal_step;
                for (int i = 0; i < L * ld; ++i) dl[i] *= arm.optimal_step;
                for (int c = 0; c < nb; ++c) db[c] *= arm.optimal_step;
            }
        }

        /* ---- ComputeCandidatePointAndEvaluateCost ---- */
        t0 = now_s();
        plus_all(p, &g, x_pose, x_pt, x_n, x_sh, dp, dl, db, c_pose, c_pt, c_n, nb ? c_sh : NULL);
        double candidate_cost = evaluate(p, c_pose, c_pt, c_n, ph ? c_sh : NULL, NULL, NULL, NULL, NULL);
        if (!isfinite(candidate_cost)) candidate_cost = DBL_MAX;
        s->update_time_s += now_s() - t0;

        /* ---- ParameterToleranceReached ---- */
        double step_norm = sqrt(x_sq_diff(&g, x_pose, x_pt, x_n, x_sh, c_pose, c_pt, c_n, c_sh, NULL));
        if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
            term = ORC_CONVERGENCE;
            break;
        }
        /* ---- FunctionToleranceReached ---- */
        double cost_change = x_cost - candidate_cost;
        if (fabs(cost_change) <= o->function_tolerance * x_cost) {
            term = ORC_CONVERGENCE;
            break;
        }
        /* ---- IsStepSuccessful ---- */
        double rd = se_quality(&se, candidate_cost, mcc);
        if (rd > o->min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(x_pose, c_pose, (size_t)P * 12 * sizeof(double));
            memcpy(x_pt, c_pt, (size_t)L * 3 * sizeof(double));
            if (ph) memcpy(x_n, c_n, (size_t)L * 3 * sizeof(double));
            if (nb) memcpy(x_sh, c_sh, (size_t)nsh * sizeof(double));
            x_norm = sqrt(x_sq_diff(&g, x_pose, x_pt, x_n, x_sh, NULL, NULL, NULL, NULL, NULL));
            t0 = now_s();
            linearize(p, &g, x_pose, x_pt, x_n, ph ? x_sh : NULL, &w);
            s->linearize_time_s += now_s() - t0;
            x_cost = w.cost;
            GRADIENT_MAX_NORM();
            if (dogleg) {
                /* DoglegStrategy::StepAccepted */
                if (rd < 0.25) dg.radius *= 0.5;
                if (rd > 0.75) dg.radius = fmax(dg.radius, 3.0 * dg.dogleg_step_norm);
                dg.mu = fmax(1e-8, 2.0 * dg.mu / 10.0);
                dg.reuse = 0;
                radius = dg.radius;
            } else {
                /* LevenbergMarquardtStrategy::StepAccepted */
                radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rd - 1.0, 3));
                radius = fmin(o->max_trust_region_radius, radius);
                decrease_factor = 2.0;
            }
            se_accepted(&se, candidate_cost, mcc);
            last_successful = 1;
            ++s->num_successful_steps;
            log_push(log, s, x_cost, cost_change, gmax, step_norm, rd, radius, 1);
        } else {
            /* HandleUnsuccessfulStep: StepRejected */
            if (dogleg) {
                dg.radius *= 0.5; dg.reuse = 1; radius = dg.radius;
            } else {
                radius /= decrease_factor;
                decrease_factor *= 2.0;
            }
            ++s->num_unsuccessful_steps;
            log_push(log, s, candidate_cost, cost_change, gmax, step_norm, rd, radius, 0);
        }
    }
#undef GRADIENT_MAX_NORM
    s->termination_type = term;
    /* solver.cc SetSummaryFinalCost: min over recorded iteration costs */
    s->final_cost = s->initial_cost;
    if (log) {
        int n = s->num_iterations < log->capacity ? s->num_iterations : log->capacity;
        for (int i = 0; i < n; ++i)
            if (log->cost[i] < s->final_cost) s->final_cost = log->cost[i];
    } else if (minimum_cost < s->final_cost) {
        s->final_cost = minimum_cost;
    }
    /* user state <- lowest-cost iterate (solution usable unless FAILURE) */
    if (term != ORC_FAILURE) {
        memcpy(p->poses, best_pose, (size_t)P * 12 * sizeof(double));
        memcpy(p->points, best_pt, (size_t)L * 3 * sizeof(double));
        if (ph) memcpy(p->normals, best_n, (size_t)L * 3 * sizeof(double));
        if (nb) {
            const int M = g.M;
            if (g.b_light >= 0) memcpy(p->light, best_sh, 3 * sizeof(double));
            if (g.b_phong >= 0) memcpy(p->phong, best_sh + 3, (size_t)3 * M * sizeof(double));
            if (g.b_tex >= 0) memcpy(p->texture, best_sh + 3 + 3 * M, (size_t)M * sizeof(double));
        }
    }
    free(x_pose); free(x_pt); free(x_n); free(c_pose); free(c_pt); free(c_n);
    free(best_pose); free(best_pt); free(best_n);
    free(dp); free(dl); free(ngp); free(ngl); free(sp); free(sl);
    free(x_sh); free(c_sh); free(best_sh);
    if (constrained) { lin_free(&w2); free(ls_pose); free(ls_pt); free(ls_n); free(ls_sh); free(sdp); free(sdl); }
    free(dg.gn_p); free(dg.gn_l); free(dg.v_p); free(dg.v_l);
    lin_free(&w);
    graph_free(&g);
    s->total_time_s = now_s() - t_start;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Phong lighting rows                                 */
/* ------------------------------------------------------------------------ */

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Forward value and, optionally, the gradients of the clamped Phong intensity w.r.t. the
 * camera-frame normal nc, UNIT light direction ell and UNIT camera direction cd, plus the
 * partials w.r.t. (kd, ks, alpha).  lighting/phong.hpp:25-51 (shade: ambient = 0),
 * :59-74 (diffuse guard ldn <= 0), :77-104 (mirror direction, guards |m|^2 <= 0 and
 * s <= 0, ks * pow(s, alpha)), :136-139 + utils/utils.hpp:16-25 (clamp through the templated
 * fmax/fmin: at the clamp the whole Jet is replaced by a constant => zero derivatives). */
static double phong_core(const double nc[3], const double ell[3], const double cd[3], double kd,
                         double ks, double alpha, double g_nc[3], double g_ell[3], double g_cd[3],
                         double g_mat[3] /* d/dkd, d/dks, d/dalpha */) {
    double diffuse = 0.0, specular = 0.0;
    int want = g_nc != NULL;
    if (want) {
        for (int i = 0; i < 3; ++i) g_nc[i] = g_ell[i] = g_cd[i] = g_mat[i] = 0.0;
    }
    const int finite = isfinite(ell[0]) && isfinite(ell[1]) && isfinite(ell[2]);
    const double ldn = dot3(ell, nc);
    const int diff_on = finite && !(ldn <= 0.0);
    if (diff_on) {
        diffuse = kd * ldn;
        if (want) {
            for (int i = 0; i < 3; ++i) { g_ell[i] += kd * nc[i]; g_nc[i] += kd * ell[i]; }
            g_mat[0] = ldn;
        }
    }
    /* mirror direction m~ = 2 (n.l) n - l */
    double mt[3];
    for (int i = 0; i < 3; ++i) mt[i] = 2.0 * ldn * nc[i] - ell[i];
    const double mu2 = dot3(mt, mt);
    if (!(mu2 <= 0.0)) {
        const double mu = sqrt(mu2);
        double m[3] = {mt[0] / mu, mt[1] / mu, mt[2] / mu};
        const double s = dot3(m, cd);
        if (!(s <= 0.0)) {
            const double sa = pow(s, alpha);
            specular = ks * sa;
            if (want) {
                const double gs = ks * alpha * pow(s, alpha - 1.0);
                /* w = (I - m m^T) cd / mu = d s / d m~ */
                double w[3];
                for (int i = 0; i < 3; ++i) w[i] = (cd[i] - m[i] * s) / mu;
                const double nw = dot3(nc, w);
                for (int i = 0; i < 3; ++i) {
                    g_ell[i] += gs * (2.0 * nc[i] * nw - w[i]);
                    g_nc[i] += gs * 2.0 * (ldn * w[i] + ell[i] * nw);
                    g_cd[i] += gs * m[i];
                }
                g_mat[1] = sa;
                g_mat[2] = ks * sa * log(s);
            }
        }
    }
    double col = 1.0 * (0.0 + diffuse + specular);
    int clamped = 0;
    if (0.0 >= col) { col = 0.0; clamped = 1; }     /* fmax(Colour(0), col) */
    if (1.0 <= col) { col = 1.0; clamped = 1; }     /* fmin(Colour(1), col) */
    if (clamped && want)
        for (int i = 0; i < 3; ++i) g_nc[i] = g_ell[i] = g_cd[i] = g_mat[i] = 0.0;
    return col;
}

double orc_phong_shade(const double n[3], const double ldir[3], const double cdir[3], double kd,
                       double ks, double alpha) {
    return phong_core(n, ldir, cdir, kd, ks, alpha, NULL, NULL, NULL, NULL);
}

/* Intensity of a camera-frame vertex (q, nc) under a light given in the camera frame
 * (position lc, or un-normalised direction dc), camera at the origin; gradients w.r.t. q, nc
 * and the light vector.  point_light.hpp:76-90, directional_light.hpp:32-35,82-91. */
static double light_core(int light_type, const double q[3], const double nc[3], const double lv[3],
                         double kd, double ks, double alpha, double g_q[3], double g_nc[3],
                         double g_l[3], double g_mat[3]) {
    double ell[3], cd[3], rho;
    if (light_type == ORC_POINT_LIGHT) {
        double v[3] = {lv[0] - q[0], lv[1] - q[1], lv[2] - q[2]};
        rho = sqrt(dot3(v, v));
        for (int i = 0; i < 3; ++i) ell[i] = v[i] / rho;
    } else {
        rho = sqrt(dot3(lv, lv));
        for (int i = 0; i < 3; ++i) ell[i] = lv[i] / rho;
    }
    const double qn = sqrt(dot3(q, q));
    for (int i = 0; i < 3; ++i) cd[i] = -q[i] / qn;
    if (!g_q) return phong_core(nc, ell, cd, kd, ks, alpha, NULL, NULL, NULL, NULL);
    double g_ell[3], g_cd[3];
    const double col = phong_core(nc, ell, cd, kd, ks, alpha, g_nc, g_ell, g_cd, g_mat);
    /* d ell / d v = (I - ell ell^T)/rho ; d cd / d q = -(I - cd cd^T)/|q| */
    const double le = dot3(ell, g_ell), ce = dot3(cd, g_cd);
    for (int i = 0; i < 3; ++i) {
        const double gv = (g_ell[i] - ell[i] * le) / rho;
        const double gc = -(g_cd[i] - cd[i] * ce) / qn;
        g_l[i] = gv;
        g_q[i] = gc - (light_type == ORC_POINT_LIGHT ? gv : 0.0);
    }
    return col;
}

double orc_light_shade(int light_type, const double p[3], const double n[3], const double light[3],
                       double kd, double ks, double alpha) {
    return light_core(light_type, p, n, light, kd, ks, alpha, NULL, NULL, NULL, NULL);
}

void orc_unit_vector_plus(const double x[3], const double delta[3], double out[3]) {
    const double s = dot3(delta, x) / dot3(x, x);
    double y[3] = {x[0] + delta[0] - s * x[0], x[1] + delta[1] - s * x[1], x[2] + delta[2] - s * x[2]};
    const double nrm = sqrt(dot3(y, y));
    for (int i = 0; i < 3; ++i) out[i] = y[i] / nrm;
}

/* plus-Jacobian of UnitVectorPerturbation at delta = 0: (I - x^ x^^T)/|x| (x^ = x/|x|) */
static void unit_plus_jacobian(const double x[3], double P[9]) {
    const double n2 = dot3(x, x), nrm = sqrt(n2);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P[3 * i + j] = ((i == j ? 1.0 : 0.0) - x[i] * x[j] / n2) / nrm;
}

static void rot_vec(const double T[12], const double v[3], double out[3]) {   /* se3group.hpp:242-244 */
    const double *R = T + 3;
    for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
/* row-vector g (1x3) times [ I | -a^ ]  ->  (g, g x (-a)^...) : returns the 3 rotational entries
 * of g^T [-a^] = (a x g)^T... explicitly: g^T (-a^) = ( -g1 a2 + g2 a1,  g0 a2 - g2 a0, -g0 a1 + g1 a0 ) */
static void row_times_neg_skew(const double g[3], const double a[3], double out[3]) {
    out[0] = -g[1] * a[2] + g[2] * a[1];
    out[1] = g[0] * a[2] - g[2] * a[0];
    out[2] = -g[0] * a[1] + g[1] * a[0];
}

void orc_intensity_residual(int light_type, const double T[12], const double p[3],
                            const double n[3], const double phong[3], double kd,
                            const double light[3], double colour, double stiffness, double *r,
                            double *J19) {
    double q[3], nc[3], lc[3];
    orc_se3_transform(T, p, q);
    rot_vec(T, n, nc);
    if (light_type == ORC_POINT_LIGHT) orc_se3_transform(T, light, lc);
    else rot_vec(T, light, lc);
    if (!J19) {
        *r = stiffness * (light_core(light_type, q, nc, lc, kd, phong[1], phong[2], NULL, NULL, NULL, NULL) - colour);
        return;
    }
    double g_q[3], g_nc[3], g_l[3], g_mat[3];
    const double col = light_core(light_type, q, nc, lc, kd, phong[1], phong[2], g_q, g_nc, g_l, g_mat);
    *r = stiffness * (col - colour);
    const double *R = T + 3;
    /* pose: dq/deps = [I | -q^], dnc/deps = [0 | -nc^], dlc/deps = [I | -lc^] (point) or [0 | -lc^] */
    double rq[3], rn[3], rl[3];
    row_times_neg_skew(g_q, q, rq);
    row_times_neg_skew(g_nc, nc, rn);
    row_times_neg_skew(g_l, lc, rl);
    for (int i = 0; i < 3; ++i) {
        J19[i] = stiffness * (g_q[i] + (light_type == ORC_POINT_LIGHT ? g_l[i] : 0.0));
        J19[3 + i] = stiffness * (rq[i] + rn[i] + rl[i]);
    }
    /* point: g_q^T R */
    for (int j = 0; j < 3; ++j) J19[6 + j] = stiffness * (g_q[0] * R[j] + g_q[1] * R[3 + j] + g_q[2] * R[6 + j]);
    /* normal: g_nc^T R P(n) */
    double gR[3], P[9];
    for (int j = 0; j < 3; ++j) gR[j] = g_nc[0] * R[j] + g_nc[1] * R[3 + j] + g_nc[2] * R[6 + j];
    unit_plus_jacobian(n, P);
    for (int j = 0; j < 3; ++j) J19[9 + j] = stiffness * (gR[0] * P[j] + gR[1] * P[3 + j] + gR[2] * P[6 + j]);
    /* phong params (ka, ks, alpha): ambient is disabled (phong.hpp:33) => d/dka = 0 */
    J19[12] = 0.0;
    J19[13] = stiffness * g_mat[1];
    J19[14] = stiffness * g_mat[2];
    /* texture kd */
    J19[15] = stiffness * g_mat[0];
    /* light: g_l^T R, through the unit-vector plus-Jacobian when directional */
    for (int j = 0; j < 3; ++j) gR[j] = g_l[0] * R[j] + g_l[1] * R[3 + j] + g_l[2] * R[6 + j];
    if (light_type == ORC_POINT_LIGHT) {
        for (int j = 0; j < 3; ++j) J19[16 + j] = stiffness * gR[j];
    } else {
        unit_plus_jacobian(light, P);
        for (int j = 0; j < 3; ++j) J19[16 + j] = stiffness * (gR[0] * P[j] + gR[1] * P[3 + j] + gR[2] * P[6 + j]);
    }
}

void orc_normal_residual(const double T[12], const double n[3], const double n_obs[3],
                         const double S[9], double r[3], double *Jpose, double *Jn) {
    double nc[3];
    rot_vec(T, n, nc);
    const double e[3] = {nc[0] - n_obs[0], nc[1] - n_obs[1], nc[2] - n_obs[2]};
    for (int i = 0; i < 3; ++i) r[i] = S[3 * i] * e[0] + S[3 * i + 1] * e[1] + S[3 * i + 2] * e[2];
    if (Jpose) {
        for (int i = 0; i < 3; ++i) {
            double rr[3];
            row_times_neg_skew(S + 3 * i, nc, rr);
            Jpose[6 * i] = Jpose[6 * i + 1] = Jpose[6 * i + 2] = 0.0;
            Jpose[6 * i + 3] = rr[0]; Jpose[6 * i + 4] = rr[1]; Jpose[6 * i + 5] = rr[2];
        }
    }
    if (Jn) {
        const double *R = T + 3;
        double P[9], SR[9];
        unit_plus_jacobian(n, P);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) SR[3 * i + j] = S[3 * i] * R[j] + S[3 * i + 1] * R[3 + j] + S[3 * i + 2] * R[6 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Jn[3 * i + j] = SR[3 * i] * P[j] + SR[3 * i + 1] * P[3 + j] + SR[3 * i + 2] * P[6 + j];
    }
}

/* ------------------------------------------------------------------------ */
/* Front end: VO initial guess                        */
/* ------------------------------------------------------------------------ */
/* std::mt19937 (the generator of point_cloud_aligner.cpp:71-73, seeded with 42) */
void orc_mt19937_seed(orc_mt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
uint32_t orc_mt19937_next(orc_mt19937 *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
/* std::uniform_int_distribution<unsigned>(0, n - 1)(rng) -- implementation-defined by the C++ standard; the two
 * libstdc++ algorithms are restated: variant 1 = GCC >= 11 (Lemire's nearly divisionless method on the 32-bit
 * generator, bits/uniform_int_dist.h _S_nd), variant 0 = GCC <= 10 (scaling + rejection). */
uint32_t orc_uniform_uint(orc_mt19937 *g, uint32_t n, int variant) {
    if (variant == 1) {
        uint64_t product = (uint64_t)orc_mt19937_next(g) * (uint64_t)n;
        uint32_t low = (uint32_t)product;
        if (low < n) {
            const uint32_t threshold = (uint32_t)(-n) % n;
            while (low < threshold) {
                product = (uint64_t)orc_mt19937_next(g) * (uint64_t)n;
                low = (uint32_t)product;
            }
        }
        return (uint32_t)(product >> 32);
    }
    const uint64_t urngrange = 0xFFFFFFFFull, uerange = (uint64_t)n;
    const uint64_t scaling = urngrange / uerange, past = uerange * scaling;
    uint64_t ret;
    do ret = orc_mt19937_next(g); while (ret >= past);
    return (uint32_t)(ret / scaling);
}

/* 3 unique indices per RANSAC iteration (point_cloud_aligner.cpp:82-91); the generator is re-seeded with 42 in
 * every call of compute_transformation_and_inliers (:71-73) */
void orc_ransac_samples(uint32_t n, uint32_t num_iters, int variant, uint32_t *idx3) {
    orc_mt19937 g;
    orc_mt19937_seed(&g, 42u);
    for (uint32_t it = 0; it < num_iters; ++it) {
        uint32_t a = orc_uniform_uint(&g, n, variant), b, c;
        b = orc_uniform_uint(&g, n, variant);
        while (b == a) b = orc_uniform_uint(&g, n, variant);
        c = orc_uniform_uint(&g, n, variant);
        while (c == a || c == b) c = orc_uniform_uint(&g, n, variant);
        idx3[3 * it] = a; idx3[3 * it + 1] = b; idx3[3 * it + 2] = c;
    }
}

/* symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations: A = V diag(w) V^T, eigenvalues descending */
static void jacobi_eig3(const double A[9], double w[3], double V[9]) {
    double a[9];
    memcpy(a, A, sizeof a);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {        /* A <- A J */
                    const double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - s * akq; a[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {        /* A <- J^T A */
                    const double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - s * aqk; a[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    double d[3] = {a[0], a[4], a[8]};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (d[ord[j]] > d[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[9];
    for (int c = 0; c < 3; ++c) { w[c] = d[ord[c]]; for (int r = 0; r < 3; ++r) Vs[3 * r + c] = V[3 * r + ord[c]]; }
    memcpy(V, Vs, sizeof Vs);
}

static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

/* PointCloudAligner::compute_transformation (point_cloud_aligner.cpp:12-62): centroids, W_1_0 = mean of
 * (p1 - c1)(p0 - c0)^T, C_1_0 = U diag(1, 1, det U det V) V^T from the SVD of W, r = c1 - C c0.  For the
 * 3-point samples of the RANSAC loop W has rank <= 2; the product is then fixed by the two leading singular
 * pairs: C = u1 v1^T + u2 v2^T + (u1 x u2)(v1 x v2)^T, whatever signs / third vectors an SVD routine returns.
 * Full-rank W (n > 3) goes through the same formula with the det correction.  T = [t | R row-major]. */
void orc_align_points(const double *pts0, const double *pts1, int n, double T[12]) {
    double c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) { c0[c] += pts0[3 * i + c]; c1[c] += pts1[3 * i + c]; }
    for (int c = 0; c < 3; ++c) { c0[c] /= (double)n; c1[c] /= (double)n; }
    double W[9] = {0};
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) W[3 * r + c] += (pts1[3 * i + r] - c1[r]) * (pts0[3 * i + c] - c0[c]);
    for (int i = 0; i < 9; ++i) W[i] /= (double)n;
    double WtW[9], w[3], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += W[3 * k + r] * W[3 * k + c];
            WtW[3 * r + c] = v;
        }
    jacobi_eig3(WtW, w, V);
    double v1[3] = {V[0], V[3], V[6]}, v2[3] = {V[1], V[4], V[7]}, v3[3], u1[3], u2[3], u3[3];
    for (int r = 0; r < 3; ++r) {
        u1[r] = W[3 * r] * v1[0] + W[3 * r + 1] * v1[1] + W[3 * r + 2] * v1[2];
        u2[r] = W[3 * r] * v2[0] + W[3 * r + 1] * v2[1] + W[3 * r + 2] * v2[2];
    }
    double n1 = sqrt(dot3(u1, u1)), n2;
    for (int r = 0; r < 3; ++r) u1[r] /= n1;
    /* re-orthogonalise u2 against u1 (it is orthogonal in exact arithmetic) */
    const double d12 = dot3(u1, u2);
    for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
    n2 = sqrt(dot3(u2, u2));
    for (int r = 0; r < 3; ++r) u2[r] /= n2;
    /* third pair: with V = [v1 v2 v1 x v2] (det +1) and U = [u1 u2 +-(u1 x u2)] the product
     * diag(1, 1, det U det V) maps the third term to (u1 x u2)(v1 x v2)^T whatever the sign */
    cross3(v1, v2, v3);
    cross3(u1, u2, u3);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[3 + 3 * r + c] = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
    for (int r = 0; r < 3; ++r) T[r] = c1[r] - (T[3 + 3 * r] * c0[0] + T[4 + 3 * r] * c0[1] + T[5 + 3 * r] * c0[2]);
}

/* PointCloudAligner::compute_transformation_and_inliers (point_cloud_aligner.cpp:64-136) with the sample
 * indices given (orc_ransac_samples): returns the number of inliers of the best hypothesis (first maximum),
 * its transformation and the inlier flags. */
int orc_ransac_align(const orc_camera *cam, const double *pts0, const double *pts1, int n, const uint32_t *idx3,
                     int num_iters, double thresh, double T_best[12], uint8_t *inlier) {
    int best = 0;   /* best_inlier_idx starts empty: a hypothesis needs more than 0 inliers to be taken */
    memset(T_best, 0, 12 * sizeof(double));
    T_best[3] = T_best[7] = T_best[11] = 1.0;     /* default-constructed SE3: identity */
    if (inlier) memset(inlier, 0, (size_t)n);
    uint8_t *cur = malloc((size_t)(n > 0 ? n : 1));
    for (int it = 0; it < num_iters; ++it) {
        double s0[9], s1[9], T[12];
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) { s0[3 * k + c] = pts0[3 * idx3[3 * it + k] + c]; s1[3 * k + c] = pts1[3 * idx3[3 * it + k] + c]; }
        orc_align_points(s0, s1, 3, T);
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            double q[3], a[3], b[3];
            orc_se3_transform(T, pts0 + 3 * i, q);
            orc_project(cam, pts1 + 3 * i, a, NULL);
            orc_project(cam, q, b, NULL);
            const double e = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
            cur[i] = e < thresh;
            cnt += cur[i];
        }
        if (cnt > best) {
            best = cnt;
            memcpy(T_best, T, sizeof T);
            if (inlier) memcpy(inlier, cur, (size_t)n);
        }
    }
    free(cur);
    return best;
}

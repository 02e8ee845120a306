// Scalar logic of the projected line search Ceres 1.x runs inside its trust-region loop when
// parameter bounds are set [trust_region_minimizer.cc DoLineSearch -> line_search.cc
// ArmijoLineSearch::DoSearch, polynomial.cc], with the Solver::Options defaults the reference drivers
// leave untouched: CUBIC interpolation, sufficient decrease 1e-4, step contraction in [1e-3, 0.6], at most
// 20 iterations, minimum step size 1e-9.  The device evaluates phi(a) = cost(Plus(x, a * delta)) and
// phi'(a) = delta . gradient(Plus(x, a * delta)); this state machine says which step to try next.  It runs on the device
// (one lane of k_ph_ls_reduce, ssba_phong_solver.hip) and, for the searches the device hands back, on the host
// (ssba_api.hip: finish_pending_search) -- the same code for both, and the same BITS: no libm beyond sqrt / fabs / fmin /
// fmax, no contraction of a * b + c into one rounding (the device would, the host build would not), so that a search gives
// the same step whichever side runs it (tests/test_gpu_phong_solve.py: any split of the evaluations reproduces the solve).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace ssba {

#define SSBA_HD __host__ __device__
#define SSBA_NO_CONTRACT _Pragma("clang fp contract(off)")
// arrays that are indexed at run time: in LDS on the device (ONE lane of a work-group runs this code; registers cannot be
// indexed, so the arrays would live in scratch memory behind ~1 us loads), on the stack on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define SSBA_WORK __shared__
#else
#define SSBA_WORK
#endif

struct LsSample { double x, value, gradient; int value_ok, gradient_ok; };

SSBA_HD inline bool ls_finite(double v) { return fabs(v) <= 1.7976931348623157e308; }     // (NaN compares false)

struct LsCx { double re, im; };
SSBA_HD inline LsCx cx(double re, double im = 0.0) { return LsCx{re, im}; }
SSBA_HD inline LsCx operator+(LsCx a, LsCx b) { SSBA_NO_CONTRACT return cx(a.re + b.re, a.im + b.im); }
SSBA_HD inline LsCx operator-(LsCx a, LsCx b) { SSBA_NO_CONTRACT return cx(a.re - b.re, a.im - b.im); }
SSBA_HD inline LsCx operator*(LsCx a, LsCx b) { SSBA_NO_CONTRACT return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
SSBA_HD inline LsCx operator/(LsCx a, LsCx b) {      // Smith's scaling: no overflow of |b|^2
    SSBA_NO_CONTRACT
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, den = b.re + b.im * r;
        return cx((a.re + a.im * r) / den, (a.im - a.re * r) / den);
    }
    const double r = b.re / b.im, den = b.re * r + b.im;
    return cx((a.re * r + a.im) / den, (a.im * r - a.re) / den);
}
SSBA_HD inline double cx_abs(LsCx a) {               // scaled: no overflow of the squares
    SSBA_NO_CONTRACT
    const double p = fmax(fabs(a.re), fabs(a.im)), q = fmin(fabs(a.re), fabs(a.im));
    if (p == 0.0) return 0.0;
    const double r = q / p;
    return p * sqrt(1.0 + r * r);
}
SSBA_HD inline double ls_ipow(double x, int k) {
    SSBA_NO_CONTRACT
    double v = 1.0;
    for (int i = 0; i < k; ++i) v *= x;
    return v;
}
// unit vectors at the angles 2 pi i / deg + 0.4 (the root finder's start), deg = 3 .. 6: constants, so that host and device start
// from the same bits (their cos / sin do not agree to the last one)
SSBA_HD inline void ls_start_dir(int deg, int i, double *c, double *sn) {
    const double t[4][6][2] = {
    {{0.9210609940028851, 0.3894183423086505}, {-0.7977766741403581, 0.60295304808712}, {-0.12328431986252686, -0.9923713903957702}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}},
    {{0.9210609940028851, 0.3894183423086505}, {-0.38941834230865036, 0.9210609940028851}, {-0.9210609940028852, -0.3894183423086503}, {0.38941834230865063, -0.921060994002885}, {0.0, 0.0}, {0.0, 0.0}},
    {{0.9210609940028851, 0.3894183423086505}, {-0.08573535201472558, 0.9963179459464289}, {-0.9740483555854224, 0.22634001188772324}, {-0.5162596384230086, -0.8564321255857609}, {0.6549823520202719, -0.7556441745570415}, {0.0, 0.0}},
    {{0.9210609940028851, 0.3894183423086505}, {0.12328431986252673, 0.9923713903957703}, {-0.7977766741403581, 0.60295304808712}, {-0.9210609940028852, -0.3894183423086503}, {-0.12328431986252686, -0.9923713903957702}, {0.7977766741403586, -0.6029530480871194}}};
    *c = t[deg - 3][i][0];
    *sn = t[deg - 3][i][1];
}

SSBA_HD inline double ls_poly_eval(const double *c, int n, double x) {
    SSBA_NO_CONTRACT
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = v * x + c[i];
    return v;
}

// Radius of a disc that holds every root of the monic polynomial m[0] = 1, m[1..deg] (Fujiwara: |z| <= 2 max_k |m_k|^(1/k)),
// with the k-th roots bounded from above through square roots alone (x >= 1: the largest power of two <= k, x < 1: the
// smallest one >= k) -- the start circle of the Aberth iteration.  Cauchy's 1 + max |m_k| can be 1e10 x the roots here
// (the dogleg's quartic has m_4 ~ lambda^4 for roots ~ lambda), which costs ~40 sweeps of slow contraction on one lane.
SSBA_HD inline double ls_root_radius(const double *m, int deg) {
    SSBA_NO_CONTRACT
    double R = 0.0, cauchy = 0.0;
    for (int k = 1; k <= deg; ++k) {
        double x = fabs(m[k]);
        cauchy = fmax(cauchy, x);
        const int nsq = x >= 1.0 ? (k >= 4 ? 2 : k >= 2 ? 1 : 0) : (k >= 5 ? 3 : k >= 3 ? 2 : k >= 2 ? 1 : 0);
        for (int j = 0; j < nsq; ++j) x = sqrt(x);
        R = fmax(R, x);
    }
    return R > 0.0 ? fmin(1.0 + cauchy, 2.0 * R) : 1.0;
}

// real parts of all roots (FindPolynomialRoots(p, &real, NULL)); degree <= 5 after dropping leading zeros
SSBA_HD inline int ls_poly_roots_real(const double *coef_in, int ncoef, double *re) {
    SSBA_NO_CONTRACT
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
    if (deg > 6) return -1;
    SSBA_WORK double m[8];
    for (int i = 0; i <= deg; ++i) {
        m[i] = c[i] / c[0];
        if (!ls_finite(m[i])) return -1;
    }
    const double radius = ls_root_radius(m, deg);
    SSBA_WORK LsCx z[8];
    for (int i = 0; i < deg; ++i) {
        double c, sn;
        ls_start_dir(deg, i, &c, &sn);
        z[i] = cx(radius * c, radius * sn);
    }
    int polished = 0;
    for (int it = 0; it < 500; ++it) {     // Aberth-Ehrlich (Ceres: eigenvalues of the companion matrix)
        double worst = 0.0;
        for (int i = 0; i < deg; ++i) {
            LsCx pv = cx(m[0]), dv = cx(0.0);
            for (int k = 1; k <= deg; ++k) { dv = dv * z[i] + pv; pv = pv * z[i] + cx(m[k]); }
            if (cx_abs(pv) == 0.0) continue;
            LsCx ratio = cx_abs(dv) == 0.0 ? cx(1e-3 * (1.0 + cx_abs(z[i]))) : pv / dv, sum = cx(0.0);
            for (int k = 0; k < deg; ++k)
                if (k != i) sum = sum + cx(1.0) / (z[i] - z[k]);
            const LsCx step = ratio / (cx(1.0) - ratio * sum);
            z[i] = z[i] - step;
            worst = fmax(worst, cx_abs(step) / (1.0 + cx_abs(z[i])));
        }
        if (polished) break;                  /* cubic convergence: one sweep after the 1e-13 sweep reaches rounding level */
        if (worst < 1e-13) polished = 1;
    }
    for (int i = 0; i < deg; ++i) {
        if (!ls_finite(z[i].re)) return -1;
        re[i] = z[i].re;
    }
    return deg;
}

// FindInterpolatingPolynomial (full-pivot elimination) + MinimizeInterpolatingPolynomial
SSBA_HD inline double ls_minimize(const LsSample *smp, int ns, double x_min, double x_max) {
    SSBA_NO_CONTRACT
    int nc = 0;
    for (int i = 0; i < ns; ++i) nc += (smp[i].value_ok ? 1 : 0) + (smp[i].gradient_ok ? 1 : 0);
    const int deg = nc - 1;
    SSBA_WORK double A[6][7];
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (smp[i].value_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = ls_ipow(smp[i].x, deg - j);
            A[row][nc] = smp[i].value;
            ++row;
        }
        if (smp[i].gradient_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = j < deg ? (deg - j) * ls_ipow(smp[i].x, deg - j - 1) : 0.0;
            A[row][nc] = smp[i].gradient;
            ++row;
        }
    }
    SSBA_WORK int perm[6];
    for (int i = 0; i < nc; ++i) perm[i] = i;
    for (int k = 0; k < nc; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int i = k; i < nc; ++i)
            for (int j = k; j < nc; ++j)
                if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); pr = i; pc = j; }
        if (!(best > 0.0)) { for (int i = k; i < nc; ++i) A[i][nc] = 0.0; break; }
        for (int j = 0; j <= nc; ++j) { const double t = A[k][j]; A[k][j] = A[pr][j]; A[pr][j] = t; }
        for (int i = 0; i < nc; ++i) { const double t = A[i][k]; A[i][k] = A[i][pc]; A[i][pc] = t; }
        { const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        for (int i = k + 1; i < nc; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= nc; ++j) A[i][j] -= f * A[k][j];
        }
    }
    SSBA_WORK double y[6];
    SSBA_WORK double coef[6];
    for (int i = nc - 1; i >= 0; --i) {
        double v = A[i][nc];
        for (int j = i + 1; j < nc; ++j) v -= A[i][j] * y[j];
        y[i] = A[i][i] != 0.0 ? v / A[i][i] : 0.0;
    }
    for (int i = 0; i < nc; ++i) coef[perm[i]] = y[i];
    double best_x = 0.5 * (x_min + x_max), best_v = ls_poly_eval(coef, nc, best_x), v;
    if ((v = ls_poly_eval(coef, nc, x_min)) < best_v) { best_v = v; best_x = x_min; }
    if ((v = ls_poly_eval(coef, nc, x_max)) < best_v) { best_v = v; best_x = x_max; }
    if (nc > 2) {
        SSBA_WORK double der[6];
        SSBA_WORK double roots[8];
        for (int i = 0; i < nc - 1; ++i) der[i] = (nc - 1 - i) * coef[i];
        const int nr = ls_poly_roots_real(der, nc - 1, roots);
        for (int i = 0; i < nr; ++i) {
            if (roots[i] < x_min || roots[i] > x_max) continue;
            if ((v = ls_poly_eval(coef, nc, roots[i])) < best_v) { best_v = v; best_x = roots[i]; }
        }
    }
    for (int i = 0; i < ns; ++i)
        if (smp[i].value_ok && smp[i].x >= x_min && smp[i].x <= x_max && smp[i].value < best_v) { best_v = smp[i].value; best_x = smp[i].x; }
    return best_x;
}

struct Armijo {
    LsSample initial, previous, current;
    double dir_max_norm, optimal_step;
    int num_iterations, num_feeds;
    int done, success;

    SSBA_HD void begin(double initial_cost, double initial_gradient, double dmax) {
        const LsSample none{0.0, 0.0, 0.0, 0, 0};
        initial = previous = current = none;
        initial.value = initial_cost; initial.gradient = initial_gradient;
        initial.value_ok = initial.gradient_ok = 1;
        current.x = 1.0;        // step_size_estimate
        dir_max_norm = dmax;
        optimal_step = 1.0;
        num_iterations = num_feeds = 0;
        done = success = 0;
    }
    // feed the evaluation at current.x; afterwards either done, or current.x is the next step to evaluate
    SSBA_HD void feed(double value, double gradient) {
        SSBA_NO_CONTRACT
        const double sufficient_decrease = 1e-4, max_step_contraction = 1e-3, min_step_contraction = 0.6, min_step_size = 1e-9;
        const int max_num_iterations = 20;
        ++num_feeds;
        current.value = value; current.gradient = gradient;
        current.value_ok = ls_finite(value) ? 1 : 0;
        current.gradient_ok = current.value_ok && ls_finite(gradient) ? 1 : 0;
        if (current.value_ok && !(current.value > initial.value + sufficient_decrease * initial.gradient * current.x)) {
            optimal_step = current.x; success = 1; done = 1;
            return;
        }
        if (++num_iterations >= max_num_iterations) { done = 1; return; }
        const double lo = max_step_contraction * current.x, hi = min_step_contraction * current.x;
        double step;
        if (!current.value_ok) {
            step = fmin(fmax(current.x * 0.5, lo), hi);
        } else {
            SSBA_WORK LsSample smp[3];
            int ns = 0;
            smp[ns++] = initial;
            smp[ns++] = current;
            if (previous.value_ok) smp[ns++] = previous;
            step = ls_minimize(smp, ns, lo, hi);
        }
        if (step * dir_max_norm < min_step_size) { done = 1; return; }
        previous = current;
        current = LsSample{step, 0.0, 0.0, 0, 0};
    }
};

}  // namespace ssba

// Host-side scalar logic of the projected line search Ceres 1.x runs inside its trust-region loop when
// parameter bounds are set [trust_region_minimizer.cc DoLineSearch -> line_search.cc
// ArmijoLineSearch::DoSearch, polynomial.cc], with the Solver::Options defaults the reference drivers
// leave untouched: CUBIC interpolation, sufficient decrease 1e-4, step contraction in [1e-3, 0.6], at most
// 20 iterations, minimum step size 1e-9.  The device evaluates phi(a) = cost(Plus(x, a * delta)) and
// phi'(a) = delta . gradient(Plus(x, a * delta)); this state machine says which step to try next.
#pragma once
#include <cmath>
#include <complex>
#include <cstring>

namespace ssba {

struct LsSample { double x = 0, value = 0, gradient = 0; bool value_ok = false, gradient_ok = false; };

inline double ls_poly_eval(const double *c, int n, double x) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = v * x + c[i];
    return v;
}

// real parts of all roots (FindPolynomialRoots(p, &real, NULL)); degree <= 5 after dropping leading zeros
inline int ls_poly_roots_real(const double *coef_in, int ncoef, double *re) {
    int lead = 0;
    while (lead < ncoef - 1 && coef_in[lead] == 0.0) ++lead;
    const double *c = coef_in + lead;
    const int deg = ncoef - lead - 1;
    if (deg < 0) return -1;
    if (deg == 0) return 0;
    if (deg == 1) { re[0] = -c[1] / c[0]; return 1; }
    if (deg == 2) {
        const double a = c[0], b = c[1], cc = c[2];
        const double D = b * b - 4 * a * cc, sq = std::sqrt(std::fabs(D));
        if (D >= 0) {
            if (b >= 0) { re[0] = (-b - sq) / (2.0 * a); re[1] = (2.0 * cc) / (-b - sq); }
            else { re[0] = (2.0 * cc) / (-b + sq); re[1] = (-b + sq) / (2.0 * a); }
        } else {
            re[0] = re[1] = -b / (2.0 * a);
        }
        return 2;
    }
    if (deg > 6) return -1;
    double m[8], bound = 0.0;
    for (int i = 0; i <= deg; ++i) {
        m[i] = c[i] / c[0];
        if (!std::isfinite(m[i])) return -1;
        if (i && std::fabs(m[i]) > bound) bound = std::fabs(m[i]);
    }
    std::complex<double> z[8];
    for (int i = 0; i < deg; ++i) z[i] = std::polar(1.0 + bound, 2.0 * 3.14159265358979323846 * i / deg + 0.4);
    int polished = 0;
    for (int it = 0; it < 500; ++it) {     // Aberth-Ehrlich (Ceres: eigenvalues of the companion matrix)
        double worst = 0.0;
        for (int i = 0; i < deg; ++i) {
            std::complex<double> pv = m[0], dv = 0.0;
            for (int k = 1; k <= deg; ++k) { dv = dv * z[i] + pv; pv = pv * z[i] + m[k]; }
            if (std::abs(pv) == 0.0) continue;
            std::complex<double> ratio = pv / dv, sum = 0.0;
            if (std::abs(dv) == 0.0) ratio = 1e-3 * (1.0 + std::abs(z[i]));
            for (int k = 0; k < deg; ++k)
                if (k != i) sum += 1.0 / (z[i] - z[k]);
            const std::complex<double> step = ratio / (1.0 - ratio * sum);
            z[i] -= step;
            worst = std::max(worst, std::abs(step) / (1.0 + std::abs(z[i])));
        }
        if (polished) break;                  /* cubic convergence: one sweep after the 1e-13 sweep reaches rounding level */
        if (worst < 1e-13) polished = 1;
    }
    for (int i = 0; i < deg; ++i) {
        if (!std::isfinite(z[i].real())) return -1;
        re[i] = z[i].real();
    }
    return deg;
}

// FindInterpolatingPolynomial (full-pivot elimination) + MinimizeInterpolatingPolynomial
inline double ls_minimize(const LsSample *smp, int ns, double x_min, double x_max) {
    int nc = 0;
    for (int i = 0; i < ns; ++i) nc += (smp[i].value_ok ? 1 : 0) + (smp[i].gradient_ok ? 1 : 0);
    const int deg = nc - 1;
    double A[6][7];
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (smp[i].value_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = std::pow(smp[i].x, deg - j);
            A[row][nc] = smp[i].value;
            ++row;
        }
        if (smp[i].gradient_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = j < deg ? (deg - j) * std::pow(smp[i].x, deg - j - 1) : 0.0;
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
                if (std::fabs(A[i][j]) > best) { best = std::fabs(A[i][j]); pr = i; pc = j; }
        if (!(best > 0.0)) { for (int i = k; i < nc; ++i) A[i][nc] = 0.0; break; }
        for (int j = 0; j <= nc; ++j) std::swap(A[k][j], A[pr][j]);
        for (int i = 0; i < nc; ++i) std::swap(A[i][k], A[i][pc]);
        std::swap(perm[k], perm[pc]);
        for (int i = k + 1; i < nc; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= nc; ++j) A[i][j] -= f * A[k][j];
        }
    }
    double y[6], coef[6];
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
        double der[6], roots[8];
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
    double dir_max_norm = 0.0, optimal_step = 1.0;
    int num_iterations = 0;
    bool done = false, success = false;

    void begin(double initial_cost, double initial_gradient, double dmax) {
        *this = Armijo();
        initial.x = 0.0; initial.value = initial_cost; initial.gradient = initial_gradient;
        initial.value_ok = initial.gradient_ok = true;
        current.x = 1.0;        // step_size_estimate
        dir_max_norm = dmax;
    }
    // feed the evaluation at current.x; afterwards either done, or current.x is the next step to evaluate
    void feed(double value, double gradient) {
        const double sufficient_decrease = 1e-4, max_step_contraction = 1e-3, min_step_contraction = 0.6, min_step_size = 1e-9;
        const int max_num_iterations = 20;
        current.value = value; current.gradient = gradient;
        current.value_ok = std::isfinite(value);
        current.gradient_ok = current.value_ok && std::isfinite(gradient);
        if (current.value_ok && !(current.value > initial.value + sufficient_decrease * initial.gradient * current.x)) {
            optimal_step = current.x; success = true; done = true;
            return;
        }
        if (++num_iterations >= max_num_iterations) { done = true; return; }
        const double lo = max_step_contraction * current.x, hi = min_step_contraction * current.x;
        double step;
        if (!current.value_ok) {
            step = std::min(std::max(current.x * 0.5, lo), hi);
        } else {
            LsSample smp[3];
            int ns = 0;
            smp[ns++] = initial;
            smp[ns++] = current;
            if (previous.value_ok) smp[ns++] = previous;
            step = ls_minimize(smp, ns, lo, hi);
        }
        if (step * dir_max_norm < min_step_size) { done = true; return; }
        previous = current;
        current = LsSample();
        current.x = step;
    }
};

}  // namespace ssba

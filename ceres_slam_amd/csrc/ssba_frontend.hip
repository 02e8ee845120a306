// Front end: the VO initial guess the reference computes before every solve
// (src/ceres_slam/dataset_problem.cpp:179-270 compute_initial_guess; point_cloud_aligner.cpp:64-136).  For
// every pair of consecutive states the reference runs a 400-iteration 3-point RANSAC on the CPU: per
// iteration an SVD alignment of 3 matched points and a stereo-reprojection inlier test over all matches.
// The pairs are independent, so here ALL (pair, iteration) hypotheses are scored in one launch: one
// workgroup per hypothesis (lane 0 aligns the 3 sample points, 256 lanes stride the pair's matches), then one
// workgroup per pair picks the first maximum (the reference keeps a hypothesis only if it has strictly more
// inliers) and writes its transformation and inlier flags.  Integer counts, fixed order: bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>
#include <map>
#include <mutex>

#include "../../include/ssba.h"
#include "ssba_pool.h"

namespace {

struct Cam { double fu, fv, cu, cv, b; };

__device__ __forceinline__ double dot3d(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3d(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations, eigenvalues descending, V columns
__device__ void jacobi_eig3(const double A[9], double w[3], double V[9]) {
    double a[9];
    for (int i = 0; i < 9; ++i) { a[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
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
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - s * akq; a[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
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
    const double dg[3] = {a[0], a[4], a[8]};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (dg[ord[j]] > dg[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[9];
    for (int c = 0; c < 3; ++c) { w[c] = dg[ord[c]]; for (int r = 0; r < 3; ++r) Vs[3 * r + c] = V[3 * r + ord[c]]; }
    for (int i = 0; i < 9; ++i) V[i] = Vs[i];
}

// PointCloudAligner::compute_transformation (point_cloud_aligner.cpp:12-62) for 3 points: W has rank <= 2, so
// C_1_0 = U diag(1, 1, det U det V) V^T = u1 v1^T + u2 v2^T + (u1 x u2)(v1 x v2)^T.  T = [t | R row-major].
__device__ void align3(const double s0[9], const double s1[9], double T[12]) {
    double c0[3], c1[3];
    for (int c = 0; c < 3; ++c) { c0[c] = (s0[c] + s0[3 + c] + s0[6 + c]) / 3.0; c1[c] = (s1[c] + s1[3 + c] + s1[6 + c]) / 3.0; }
    double W[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int i = 0; i < 3; ++i) v += (s1[3 * i + r] - c1[r]) * (s0[3 * i + c] - c0[c]);
            W[3 * r + c] = v / 3.0;
        }
    double WtW[9], w[3], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += W[3 * k + r] * W[3 * k + c];
            WtW[3 * r + c] = v;
        }
    jacobi_eig3(WtW, w, V);
    const double v1[3] = {V[0], V[3], V[6]}, v2[3] = {V[1], V[4], V[7]};
    double v3[3], u1[3], u2[3], u3[3];
    for (int r = 0; r < 3; ++r) {
        u1[r] = W[3 * r] * v1[0] + W[3 * r + 1] * v1[1] + W[3 * r + 2] * v1[2];
        u2[r] = W[3 * r] * v2[0] + W[3 * r + 1] * v2[1] + W[3 * r + 2] * v2[2];
    }
    // W = U S V^T: u_i = W v_i / s_i.  A degenerate sample (collinear or coincident points: rank W < 2) has s_i = 0;
    // an SVD still returns orthonormal U columns there -- complete the basis instead of dividing by zero
    // (point_cloud_aligner.cpp:49-55 uses JacobiSVD, which does the same up to the choice of the null-space basis)
    const double n1 = sqrt(dot3d(u1, u1));
    if (n1 > 1e-300) { for (int r = 0; r < 3; ++r) u1[r] /= n1; }
    else { u1[0] = 1.0; u1[1] = 0.0; u1[2] = 0.0; }
    const double d12 = dot3d(u1, u2);
    for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
    const double n2 = sqrt(dot3d(u2, u2));
    if (n2 > 1e-12 * (n1 > 1e-300 ? n1 : 1.0)) { for (int r = 0; r < 3; ++r) u2[r] /= n2; }
    else {      // any unit vector orthogonal to u1
        const int m = fabs(u1[0]) <= fabs(u1[1]) ? (fabs(u1[0]) <= fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
        double e[3] = {0.0, 0.0, 0.0};
        e[m] = 1.0;
        const double de = dot3d(u1, e);
        for (int r = 0; r < 3; ++r) u2[r] = e[r] - de * u1[r];
        const double ne = sqrt(dot3d(u2, u2));
        for (int r = 0; r < 3; ++r) u2[r] /= ne;
    }
    cross3d(v1, v2, v3);
    cross3d(u1, u2, u3);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[3 + 3 * r + c] = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
    for (int r = 0; r < 3; ++r) T[r] = c1[r] - (T[3 + 3 * r] * c0[0] + T[4 + 3 * r] * c0[1] + T[5 + 3 * r] * c0[2]);
}

// (camera->project(pts_1[i]) - camera->project(T_1_0 * pts_0[i])).squaredNorm() < thresh   (:117-124)
__device__ __forceinline__ bool is_inlier(const Cam &cam, const double T[12], const double *p0, const double *p1, double thresh) {
    const double q0 = T[3] * p0[0] + T[4] * p0[1] + T[5] * p0[2] + T[0];
    const double q1 = T[6] * p0[0] + T[7] * p0[1] + T[8] * p0[2] + T[1];
    const double q2 = T[9] * p0[0] + T[10] * p0[1] + T[11] * p0[2] + T[2];
    const double du = (cam.fu * p1[0] / p1[2] + cam.cu) - (cam.fu * q0 / q2 + cam.cu);
    const double dv = (cam.fv * p1[1] / p1[2] + cam.cv) - (cam.fv * q1 / q2 + cam.cv);
    const double dd = cam.fu * cam.b / p1[2] - cam.fu * cam.b / q2;
    return du * du + dv * dv + dd * dd < thresh;
}

__device__ void hypothesis(const double *pts0, const double *pts1, const uint32_t *smp, uint32_t base, double T[12]) {
    double s0[9], s1[9];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            s0[3 * k + c] = pts0[3 * (size_t)(base + smp[k]) + c];
            s1[3 * k + c] = pts1[3 * (size_t)(base + smp[k]) + c];
        }
    align3(s0, s1, T);
}

// grid (num_iters, num_pairs): inlier count of one hypothesis
__global__ __launch_bounds__(256) void k_fe_score(Cam cam, const uint32_t *offset, const double *pts0, const double *pts1,
                                                  const uint32_t *samples, uint32_t num_iters, double thresh, uint32_t *counts) {
    __shared__ double sT[12];
    __shared__ uint32_t sc[4];
    const uint32_t pair = blockIdx.x / num_iters, it = blockIdx.x - pair * num_iters, base = offset[pair], n = offset[pair + 1] - base;
    if (threadIdx.x == 0) {
        double T[12];
        hypothesis(pts0, pts1, samples + 3 * ((size_t)pair * num_iters + it), base, T);
        for (int i = 0; i < 12; ++i) sT[i] = T[i];
    }
    __syncthreads();
    double T[12];
    for (int i = 0; i < 12; ++i) T[i] = sT[i];
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        cnt += is_inlier(cam, T, pts0 + 3 * (size_t)(base + i), pts1 + 3 * (size_t)(base + i), thresh) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[(size_t)pair * num_iters + it] = sc[0] + sc[1] + sc[2] + sc[3];
}

// one block per pair: first maximum of the counts (a hypothesis replaces the best only with MORE inliers, :127-130),
// its transformation and inlier flags; identity and no inliers when every hypothesis scores 0
__global__ __launch_bounds__(256) void k_fe_select(Cam cam, const uint32_t *offset, const double *pts0, const double *pts1,
                                                   const uint32_t *samples, uint32_t num_iters, double thresh, const uint32_t *counts,
                                                   double *Tout, uint8_t *inlier, uint32_t *best_count) {
    __shared__ double sT[12];
    __shared__ uint32_t sBest, sIt;
    const uint32_t pair = blockIdx.x, base = offset[pair], n = offset[pair + 1] - base;
    if (threadIdx.x == 0) {
        uint32_t best = 0, bit = 0;
        for (uint32_t it = 0; it < num_iters; ++it) {
            const uint32_t c = counts[(size_t)pair * num_iters + it];
            if (c > best) { best = c; bit = it; }
        }
        double T[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (best > 0) hypothesis(pts0, pts1, samples + 3 * ((size_t)pair * num_iters + bit), base, T);
        for (int i = 0; i < 12; ++i) { sT[i] = T[i]; Tout[(size_t)pair * 12 + i] = T[i]; }
        sBest = best; sIt = bit;
        best_count[pair] = best;
    }
    __syncthreads();
    double T[12];
    for (int i = 0; i < 12; ++i) T[i] = sT[i];
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        inlier[base + i] = (sBest > 0 && is_inlier(cam, T, pts0 + 3 * (size_t)(base + i), pts1 + 3 * (size_t)(base + i), thresh)) ? 1 : 0;
}

// ---- the whole compute_initial_guess on the device (ssba_frontend_vo) -----------------------------------------------
// Reciprocal matches of states (q, q + 1) (dataset_problem.cpp:209-222): an observation of one state is kept when its
// landmark id occurs in the other state; both lists keep their state's order and are paired by position.  One workgroup
// per pair: membership by binary search in the other state's SORTED ids, order-preserving compaction by block scans,
// StereoCamera::triangulate (stereo_camera.hpp:112-120) of the kept observations.
__device__ __forceinline__ bool fe_contains(const uint32_t *sorted, uint32_t n, uint32_t id) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] < id) lo = mid + 1; else hi = mid;
    }
    return lo < n && sorted[lo] == id;
}
__global__ __launch_bounds__(256) void k_fe_match(Cam cam, const uint32_t *state_start, const uint32_t *point_id, const uint32_t *sorted_id,
                                                  const double *uvd, uint32_t *match_obs, double *pts0, double *pts1, uint32_t *match_count) {
    __shared__ uint32_t sscan[256], srun;
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    uint32_t cnt[2];
    for (int side = 0; side < 2; ++side) {       // 0: state q against q + 1 -> pts0 ; 1: state q + 1 against q -> pts1
        const uint32_t b0 = state_start[q + side], n0 = state_start[q + side + 1] - b0;
        const uint32_t bo = state_start[q + 1 - side], no = state_start[q + 2 - side] - bo;
        double *pts = side ? pts1 : pts0;
        if (t == 0) srun = 0;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n0; c0 += 256) {
            const uint32_t i = c0 + t;
            const bool keep = i < n0 && fe_contains(sorted_id + bo, no, point_id[b0 + i]);
            sscan[t] = keep ? 1u : 0u;
            __syncthreads();
            for (uint32_t o = 1; o < 256; o <<= 1) {      // inclusive scan
                const uint32_t v = t >= o ? sscan[t - o] : 0u;
                __syncthreads();
                sscan[t] += v;
                __syncthreads();
            }
            const uint32_t pos = srun + sscan[t] - (keep ? 1u : 0u);
            if (keep) {
                const uint32_t dst = state_start[q] + pos;       // pair q owns the slots of state q (matches <= its observations)
                if (pos < state_start[q + 1] - state_start[q]) {
                    const double *z = uvd + 3 * (size_t)(b0 + i);
                    const double b_over_d = cam.b / z[2];
                    pts[3 * (size_t)dst] = (z[0] - cam.cu) * b_over_d;
                    pts[3 * (size_t)dst + 1] = (z[1] - cam.cv) * b_over_d * (cam.fu / cam.fv);
                    pts[3 * (size_t)dst + 2] = cam.fu * b_over_d;
                    if (side == 0) match_obs[dst] = b0 + i;
                }
            }
            __syncthreads();
            if (t == 255) srun += sscan[255];
            __syncthreads();
        }
        cnt[side] = srun;
        __syncthreads();
    }
    // a landmark seen twice in one state makes the two lists differ in length: reported as "no matches" (the reference
    // would pair them by position anyway and align garbage)
    if (t == 0) match_count[q] = cnt[0] == cnt[1] ? cnt[0] : 0u;
}

// the reference's draws for every pair (point_cloud_aligner.cpp:69-91): std::mt19937 re-seeded with 42 per call, so
// every pair walks the same raw stream (`raw`, generated once on the host) through uniform_int_distribution(0, n - 1)
__device__ __forceinline__ uint32_t fe_uniform(const uint32_t *raw, uint32_t nraw, uint32_t &pos, uint32_t n, int variant, bool &ok) {
    auto next = [&]() -> uint32_t { if (pos >= nraw) { ok = false; return 0u; } return raw[pos++]; };
    if (variant == 1) {
        uint64_t product = (uint64_t)next() * (uint64_t)n;
        uint32_t low = (uint32_t)product;
        if (low < n) {
            const uint32_t threshold = (uint32_t)(0u - n) % n;
            while (low < threshold && ok) { product = (uint64_t)next() * (uint64_t)n; low = (uint32_t)product; }
        }
        return (uint32_t)(product >> 32);
    }
    const uint64_t scaling = 0xFFFFFFFFull / n, past = (uint64_t)n * scaling;
    uint64_t ret;
    do ret = next(); while (ret >= past && ok);
    return (uint32_t)(ret / scaling);
}
__global__ __launch_bounds__(64) void k_fe_samples(const uint32_t *raw, uint32_t nraw, const uint32_t *match_count, uint32_t num_pairs,
                                                   uint32_t num_iters, int variant, uint32_t *samples, int *status) {
    const uint32_t q = blockIdx.x * 64 + threadIdx.x;
    if (q >= num_pairs) return;
    const uint32_t n = match_count[q];
    uint32_t *out = samples + (size_t)q * 3 * num_iters;
    if (n < 3) { for (uint32_t i = 0; i < 3 * num_iters; ++i) out[i] = 0; atomicMax(status, 1); return; }
    uint32_t pos = 0;
    bool ok = true;
    for (uint32_t it = 0; it < num_iters && ok; ++it) {
        const uint32_t a = fe_uniform(raw, nraw, pos, n, variant, ok);
        uint32_t b = fe_uniform(raw, nraw, pos, n, variant, ok);
        while (b == a && ok) b = fe_uniform(raw, nraw, pos, n, variant, ok);
        uint32_t c = fe_uniform(raw, nraw, pos, n, variant, ok);
        while ((c == a || c == b) && ok) c = fe_uniform(raw, nraw, pos, n, variant, ok);
        out[3 * it] = a; out[3 * it + 1] = b; out[3 * it + 2] = c;
    }
    if (!ok) atomicMax(status, 2);
}

// one LANE per hypothesis: the 3-point alignment (the first version had lane 0 of a 256-lane workgroup do it)
__global__ __launch_bounds__(256) void k_fe_align(const uint32_t *state_start, const double *pts0, const double *pts1, const uint32_t *samples,
                                                  const uint32_t *match_count, uint32_t num_pairs, uint32_t num_iters, double *Thyp) {
    const size_t h = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (h >= (size_t)num_pairs * num_iters) return;
    const uint32_t q = (uint32_t)(h / num_iters);
    double T[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (match_count[q] >= 3) hypothesis(pts0, pts1, samples + 3 * h, state_start[q], T);
    for (int i = 0; i < 12; ++i) Thyp[12 * h + i] = T[i];
}
__global__ __launch_bounds__(256) void k_fe_score_T(Cam cam, const uint32_t *state_start, const double *pts0, const double *pts1, const double *Thyp,
                                                    const uint32_t *match_count, uint32_t num_iters, double thresh, uint32_t *counts) {
    __shared__ uint32_t sc[4];
    const uint32_t q = blockIdx.x / num_iters, base = state_start[q], n = match_count[q];
    double T[12];
    for (int i = 0; i < 12; ++i) T[i] = Thyp[12 * (size_t)blockIdx.x + i];
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        cnt += is_inlier(cam, T, pts0 + 3 * (size_t)(base + i), pts1 + 3 * (size_t)(base + i), thresh) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
// first maximum (a hypothesis replaces the best only with MORE inliers, :127-130), its T and the inlier flags
__global__ __launch_bounds__(256) void k_fe_select_T(Cam cam, const uint32_t *state_start, const double *pts0, const double *pts1, const double *Thyp,
                                                     const uint32_t *match_count, uint32_t num_iters, double thresh, const uint32_t *counts,
                                                     double *Tpair, uint8_t *inlier, uint32_t *best_count) {
    __shared__ uint32_t sBest[256], sIt[256];
    const uint32_t q = blockIdx.x, base = state_start[q], n = match_count[q], t = threadIdx.x;
    uint32_t best = 0, bit = 0;
    for (uint32_t it = t; it < num_iters; it += 256) {
        const uint32_t c = counts[(size_t)q * num_iters + it];
        if (c > best) { best = c; bit = it; }
    }
    sBest[t] = best; sIt[t] = bit;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {      // maximum count, smallest iteration among equals
        if (t < o) {
            const uint32_t b2 = sBest[t + o], i2 = sIt[t + o];
            if (b2 > sBest[t] || (b2 == sBest[t] && b2 > 0 && i2 < sIt[t])) { sBest[t] = b2; sIt[t] = i2; }
        }
        __syncthreads();
    }
    best = sBest[0]; bit = sIt[0];
    double T[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (best > 0)
        for (int i = 0; i < 12; ++i) T[i] = Thyp[12 * ((size_t)q * num_iters + bit) + i];
    if (t < 12) Tpair[(size_t)q * 12 + t] = T[t];
    if (t == 0) best_count[q] = best;
    for (uint32_t i = t; i < n; i += 256)
        inlier[base + i] = (best > 0 && is_inlier(cam, T, pts0 + 3 * (size_t)(base + i), pts1 + 3 * (size_t)(base + i), thresh)) ? 1 : 0;
}
// poses[k] = T_k_km1 * poses[k - 1] (dataset_problem.cpp:255): an inherently serial chain of 3x4 products; one wave walks
// it with the pair transforms read 64 at a time
__global__ __launch_bounds__(64) void k_fe_chain(const double *Tpair, uint32_t num_pairs, double *poses) {
    __shared__ double sT[64 * 12];
    double P[12];
    for (int i = 0; i < 12; ++i) P[i] = poses[i];
    for (uint32_t c0 = 0; c0 < num_pairs; c0 += 64) {
        const uint32_t m = num_pairs - c0 < 64 ? num_pairs - c0 : 64;
        for (uint32_t i = threadIdx.x; i < m * 12; i += 64) sT[i] = Tpair[(size_t)c0 * 12 + i];
        __syncthreads();
        for (uint32_t k = 0; k < m; ++k) {
            const double *Tk = sT + 12 * k;
            double N[12];
            for (int i = 0; i < 3; ++i) {
                N[i] = Tk[3 + 3 * i] * P[0] + Tk[4 + 3 * i] * P[1] + Tk[5 + 3 * i] * P[2] + Tk[i];
                for (int j = 0; j < 3; ++j) N[3 + 3 * i + j] = Tk[3 + 3 * i] * P[3 + j] + Tk[4 + 3 * i] * P[6 + j] + Tk[5 + 3 * i] * P[9 + j];
            }
            for (int i = 0; i < 12; ++i) P[i] = N[i];
            if (threadIdx.x < 12) poses[(size_t)(c0 + k + 1) * 12 + threadIdx.x] = P[threadIdx.x];
        }
        __syncthreads();
    }
}
// map initialisation (:259-269): a landmark takes its position from the FIRST pair in which it is an inlier match and
// not initialised yet -- pass 1 finds that pair (atomicMin), pass 2 writes poses[k-1]^-1 * pts_km1
__global__ __launch_bounds__(256) void k_fe_map_first(const uint32_t *state_start, const uint32_t *point_id, const uint32_t *match_obs,
                                                      const uint32_t *match_count, const uint8_t *inlier, const uint8_t *initialized,
                                                      uint32_t num_points, uint32_t *first_pair) {
    const uint32_t q = blockIdx.x, base = state_start[q], n = match_count[q];
    for (uint32_t m = threadIdx.x; m < n; m += 256) {
        if (!inlier[base + m]) continue;
        const uint32_t j = point_id[match_obs[base + m]];
        if (j < num_points && !initialized[j]) atomicMin(&first_pair[j], q);
    }
}
__global__ __launch_bounds__(256) void k_fe_map_write(const uint32_t *state_start, const uint32_t *point_id, const uint32_t *match_obs,
                                                      const uint32_t *match_count, const uint8_t *inlier, uint8_t *initialized, uint32_t num_points,
                                                      const uint32_t *first_pair, const double *pts0, const double *poses, double *map_points) {
    const uint32_t q = blockIdx.x, base = state_start[q], n = match_count[q];
    const double *T = poses + (size_t)q * 12;          // poses[k - 1] of pair q
    for (uint32_t m = threadIdx.x; m < n; m += 256) {
        if (!inlier[base + m]) continue;
        const uint32_t j = point_id[match_obs[base + m]];
        if (j >= num_points || first_pair[j] != q) continue;
        const double *p = pts0 + 3 * (size_t)(base + m);
        // SE3::inverse() * p (se3group.hpp:152-158, 191-193): R^T p + (-R^T t)
        for (int c = 0; c < 3; ++c) {
            const double ti = -(T[3 + c] * T[0] + T[6 + c] * T[1] + T[9 + c] * T[2]);
            map_points[3 * (size_t)j + c] = T[3 + c] * p[0] + T[6 + c] * p[1] + T[9 + c] * p[2] + ti;
        }
        initialized[j] = 1;
    }
}

// ---- host: the reference's sampling sequence --------------------------------------------------------
struct Mt19937 {       // std::mt19937
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};
// std::uniform_int_distribution<unsigned>(0, n-1) as libstdc++ implements it: variant 1 = GCC >= 11 (Lemire's
// method on the 32-bit generator), 0 = GCC <= 10 (scaling + rejection)
uint32_t uniform_uint(Mt19937 &g, uint32_t n, int variant) {
    if (variant == 1) {
        uint64_t product = (uint64_t)g.next() * (uint64_t)n;
        uint32_t low = (uint32_t)product;
        if (low < n) {
            const uint32_t threshold = (uint32_t)(0u - n) % n;
            while (low < threshold) { product = (uint64_t)g.next() * (uint64_t)n; low = (uint32_t)product; }
        }
        return (uint32_t)(product >> 32);
    }
    const uint64_t scaling = 0xFFFFFFFFull / n, past = (uint64_t)n * scaling;
    uint64_t ret;
    do ret = g.next(); while (ret >= past);
    return (uint32_t)(ret / scaling);
}

thread_local std::string g_fe_error;

}  // namespace

extern "C" {

int ssba_ransac_samples(uint32_t n, uint32_t num_iters, int libstdcxx_variant, uint32_t *idx3) {
    if (!idx3 || n < 3 || (libstdcxx_variant != 0 && libstdcxx_variant != 1)) return SSBA_ERR_INVALID_ARGUMENT;
    Mt19937 g(42u);                                         // rng.seed(42) in every call (:73)
    for (uint32_t it = 0; it < num_iters; ++it) {           // :82-91
        uint32_t a = uniform_uint(g, n, libstdcxx_variant), b, c;
        b = uniform_uint(g, n, libstdcxx_variant);
        while (b == a) b = uniform_uint(g, n, libstdcxx_variant);
        c = uniform_uint(g, n, libstdcxx_variant);
        while (c == a || c == b) c = uniform_uint(g, n, libstdcxx_variant);
        idx3[3 * it] = a; idx3[3 * it + 1] = b; idx3[3 * it + 2] = c;
    }
    return SSBA_OK;
}

int ssba_frontend_ransac(const ssba_camera *camera, int device, uint32_t num_pairs, const uint32_t *offset, const double *pts0,
                         const double *pts1, const uint32_t *samples, uint32_t num_iters, double thresh, double *T,
                         uint8_t *inlier, uint32_t *count, double *device_time_s) {
    if (!camera || !offset || !pts0 || !pts1 || !samples || !T || num_iters == 0) return SSBA_ERR_INVALID_ARGUMENT;
    if (num_pairs == 0) return SSBA_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SSBA_ERR_NO_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return SSBA_ERR_NO_DEVICE;
    if (device >= ndev || hipSetDevice(device) != hipSuccess) return SSBA_ERR_INVALID_ARGUMENT;
    const size_t npts = offset[num_pairs];
    for (uint32_t p = 0; p < num_pairs; ++p) {
        if (offset[p + 1] < offset[p]) return SSBA_ERR_INVALID_ARGUMENT;
        const uint32_t n = offset[p + 1] - offset[p];
        for (uint32_t k = 0; k < 3 * num_iters; ++k)
            if (n == 0 || samples[(size_t)p * 3 * num_iters + k] >= n) return SSBA_ERR_INVALID_ARGUMENT;
    }
    uint32_t *d_off = nullptr, *d_smp = nullptr, *d_cnt = nullptr, *d_best = nullptr;
    double *d_p0 = nullptr, *d_p1 = nullptr, *d_T = nullptr;
    uint8_t *d_in = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SSBA_OK;
#define FE_TRY(x) do { if ((x) != hipSuccess) { rc = SSBA_ERR_HIP; goto done; } } while (0)
    FE_TRY(ssba::pool_malloc((void **)&d_off, (num_pairs + 1) * sizeof(uint32_t)));
    FE_TRY(ssba::pool_malloc((void **)&d_smp, (size_t)num_pairs * num_iters * 3 * sizeof(uint32_t)));
    FE_TRY(ssba::pool_malloc((void **)&d_cnt, (size_t)num_pairs * num_iters * sizeof(uint32_t)));
    FE_TRY(ssba::pool_malloc((void **)&d_best, num_pairs * sizeof(uint32_t)));
    FE_TRY(ssba::pool_malloc((void **)&d_p0, (npts ? npts : 1) * 3 * sizeof(double)));
    FE_TRY(ssba::pool_malloc((void **)&d_p1, (npts ? npts : 1) * 3 * sizeof(double)));
    FE_TRY(ssba::pool_malloc((void **)&d_T, (size_t)num_pairs * 12 * sizeof(double)));
    FE_TRY(ssba::pool_malloc((void **)&d_in, npts ? npts : 1));
    FE_TRY(hipMemcpy(d_off, offset, (num_pairs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    FE_TRY(hipMemcpy(d_smp, samples, (size_t)num_pairs * num_iters * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
    FE_TRY(hipMemcpy(d_p0, pts0, npts * 3 * sizeof(double), hipMemcpyHostToDevice));
    FE_TRY(hipMemcpy(d_p1, pts1, npts * 3 * sizeof(double), hipMemcpyHostToDevice));
    {
        const Cam cam = {camera->fu, camera->fv, camera->cu, camera->cv, camera->b};
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_fe_score, dim3(num_iters * num_pairs), dim3(256), 0, nullptr, cam, d_off, d_p0, d_p1, d_smp, num_iters, thresh, d_cnt);
        hipLaunchKernelGGL(k_fe_select, dim3(num_pairs), dim3(256), 0, nullptr, cam, d_off, d_p0, d_p1, d_smp, num_iters, thresh, d_cnt,
                           d_T, d_in, d_best);
        hipEventRecord(e1, nullptr);
        FE_TRY(hipDeviceSynchronize());
        FE_TRY(hipGetLastError());
        float ms = 0.f;
        if (device_time_s && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *device_time_s = 1e-3 * ms;
    }
    FE_TRY(hipMemcpy(T, d_T, (size_t)num_pairs * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (inlier && npts) FE_TRY(hipMemcpy(inlier, d_in, npts, hipMemcpyDeviceToHost));
    if (count) FE_TRY(hipMemcpy(count, d_best, num_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
done:
#undef FE_TRY
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    (void)hipDeviceSynchronize();      // the buffers go back to the process-wide cache (ssba_pool.h)
    ssba::pool_free(d_off); ssba::pool_free(d_smp); ssba::pool_free(d_cnt); ssba::pool_free(d_best); ssba::pool_free(d_p0); ssba::pool_free(d_p1); ssba::pool_free(d_T); ssba::pool_free(d_in);
    return rc;
}

int ssba_frontend_vo(const ssba_camera *camera, int device, uint32_t num_states, const uint32_t *state_start, const uint32_t *point_id,
                     const double *uvd, uint32_t num_points, uint32_t num_iters, double thresh, int libstdcxx_variant, double *poses,
                     double *map_points, uint8_t *initialized, uint32_t *match_count, uint32_t *inlier_count, double *device_time_s) {
    if (!camera || !state_start || !point_id || !uvd || !poses || !map_points || !initialized || num_iters == 0 ||
        (libstdcxx_variant != 0 && libstdcxx_variant != 1))
        return SSBA_ERR_INVALID_ARGUMENT;
    if (num_states < 2) return SSBA_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SSBA_ERR_NO_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return SSBA_ERR_NO_DEVICE;
    if (device >= ndev || hipSetDevice(device) != hipSuccess) return SSBA_ERR_INVALID_ARGUMENT;
    const uint32_t num_pairs = num_states - 1;
    const size_t N = state_start[num_states];
    for (uint32_t k = 0; k < num_states; ++k)
        if (state_start[k + 1] < state_start[k]) return SSBA_ERR_INVALID_ARGUMENT;
    // ids of every state in ascending order for the membership searches (the dataset format lists them that way already)
    std::vector<uint32_t> sorted(point_id, point_id + N);
    for (uint32_t k = 0; k < num_states; ++k) {
        uint32_t *b = sorted.data() + state_start[k], *e = sorted.data() + state_start[k + 1];
        if (!std::is_sorted(b, e)) std::sort(b, e);
    }
    // raw std::mt19937(42) stream: 3 draws per iteration + rejections + duplicate redraws; 16 x that is far beyond need.
    // The stream is the same for every call with the same iteration count: generated and uploaded once per process and device.
    const uint32_t nraw = 48 * num_iters + 1024;
    static std::mutex raw_mu;
    static std::map<std::pair<int, uint32_t>, uint32_t *> raw_cache;

    // ONE device buffer for everything, laid out so that the inputs go up in one transfer and the results come back in one
    // (a window of two states is a few kilobytes: eight blocking uploads and six downloads were most of this call's 0.9 ms):
    //   [ start | id | sorted | uvd | in | first || status | mcnt | best | poses | map | init ] + scratch
    //     <------------------------- upload ------------------------------------------------->
    //                                              <----------------- download -------------->
    const size_t Nn = N ? N : 1, NP = num_points ? num_points : 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_start = take((num_states + 1) * sizeof(uint32_t)), o_id = take(Nn * sizeof(uint32_t)), o_sorted = take(Nn * sizeof(uint32_t)),
                 o_uvd = take(Nn * 3 * sizeof(double)), o_in = take(Nn), o_first = take(NP * sizeof(uint32_t));
    const size_t o_status = take(sizeof(int)), o_mcnt = take(num_pairs * sizeof(uint32_t)), o_best = take(num_pairs * sizeof(uint32_t)),
                 o_poses = take((size_t)num_states * 12 * sizeof(double)), o_map = take(NP * 3 * sizeof(double)), o_init = take(NP);
    const size_t up_bytes = off;
    const size_t o_mobs = take(Nn * sizeof(uint32_t)), o_smp = take((size_t)num_pairs * num_iters * 3 * sizeof(uint32_t)),
                 o_cnt = take((size_t)num_pairs * num_iters * sizeof(uint32_t)), o_p0 = take(Nn * 3 * sizeof(double)),
                 o_p1 = take(Nn * 3 * sizeof(double)), o_Th = take((size_t)num_pairs * num_iters * 12 * sizeof(double)),
                 o_Tp = take((size_t)num_pairs * 12 * sizeof(double));
    const size_t total_bytes = off;
    char *arena = nullptr;
    uint32_t *d_raw = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SSBA_OK, status = 0;
    std::vector<char> host(up_bytes, 0);
#define FE_TRY(x) do { if ((x) != hipSuccess) { rc = SSBA_ERR_HIP; goto done; } } while (0)
    {
        std::lock_guard<std::mutex> lock(raw_mu);
        uint32_t *&slot = raw_cache[{device, nraw}];
        if (!slot) {
            std::vector<uint32_t> raw(nraw);
            { Mt19937 g(42u); for (uint32_t i = 0; i < nraw; ++i) raw[i] = g.next(); }
            FE_TRY(hipMalloc((void **)&slot, nraw * sizeof(uint32_t)));
            FE_TRY(hipMemcpy(slot, raw.data(), nraw * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        d_raw = slot;
    }
    FE_TRY(ssba::pool_malloc((void **)&arena, total_bytes));
    memcpy(host.data() + o_start, state_start, (num_states + 1) * sizeof(uint32_t));
    memcpy(host.data() + o_id, point_id, N * sizeof(uint32_t));
    memcpy(host.data() + o_sorted, sorted.data(), N * sizeof(uint32_t));
    memcpy(host.data() + o_uvd, uvd, N * 3 * sizeof(double));
    memset(host.data() + o_first, 0xFF, NP * sizeof(uint32_t));
    memcpy(host.data() + o_poses, poses, 12 * sizeof(double));
    memcpy(host.data() + o_map, map_points, (size_t)num_points * 3 * sizeof(double));
    memcpy(host.data() + o_init, initialized, num_points);
    FE_TRY(hipMemcpy(arena, host.data(), up_bytes, hipMemcpyHostToDevice));
    {
    uint32_t *d_start = (uint32_t *)(arena + o_start), *d_id = (uint32_t *)(arena + o_id), *d_sorted = (uint32_t *)(arena + o_sorted),
             *d_mobs = (uint32_t *)(arena + o_mobs), *d_mcnt = (uint32_t *)(arena + o_mcnt), *d_smp = (uint32_t *)(arena + o_smp),
             *d_cnt = (uint32_t *)(arena + o_cnt), *d_best = (uint32_t *)(arena + o_best), *d_first = (uint32_t *)(arena + o_first);
    double *d_uvd = (double *)(arena + o_uvd), *d_p0 = (double *)(arena + o_p0), *d_p1 = (double *)(arena + o_p1), *d_Th = (double *)(arena + o_Th),
           *d_Tp = (double *)(arena + o_Tp), *d_poses = (double *)(arena + o_poses), *d_map = (double *)(arena + o_map);
    uint8_t *d_in = (uint8_t *)(arena + o_in), *d_init = (uint8_t *)(arena + o_init);
    int *d_status = (int *)(arena + o_status);
    {
        const Cam cam = {camera->fu, camera->fv, camera->cu, camera->cv, camera->b};
        const size_t nh = (size_t)num_pairs * num_iters;
        if (device_time_s) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, nullptr); }
        hipLaunchKernelGGL(k_fe_match, dim3(num_pairs), dim3(256), 0, nullptr, cam, d_start, d_id, d_sorted, d_uvd, d_mobs, d_p0, d_p1, d_mcnt);
        hipLaunchKernelGGL(k_fe_samples, dim3((num_pairs + 63) / 64), dim3(64), 0, nullptr, d_raw, nraw, d_mcnt, num_pairs, num_iters, libstdcxx_variant, d_smp, d_status);
        hipLaunchKernelGGL(k_fe_align, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, nullptr, d_start, d_p0, d_p1, d_smp, d_mcnt, num_pairs, num_iters, d_Th);
        hipLaunchKernelGGL(k_fe_score_T, dim3((unsigned)nh), dim3(256), 0, nullptr, cam, d_start, d_p0, d_p1, d_Th, d_mcnt, num_iters, thresh, d_cnt);
        hipLaunchKernelGGL(k_fe_select_T, dim3(num_pairs), dim3(256), 0, nullptr, cam, d_start, d_p0, d_p1, d_Th, d_mcnt, num_iters, thresh, d_cnt, d_Tp, d_in, d_best);
        hipLaunchKernelGGL(k_fe_chain, dim3(1), dim3(64), 0, nullptr, d_Tp, num_pairs, d_poses);
        hipLaunchKernelGGL(k_fe_map_first, dim3(num_pairs), dim3(256), 0, nullptr, d_start, d_id, d_mobs, d_mcnt, d_in, d_init, num_points, d_first);
        hipLaunchKernelGGL(k_fe_map_write, dim3(num_pairs), dim3(256), 0, nullptr, d_start, d_id, d_mobs, d_mcnt, d_in, d_init, num_points, d_first, d_p0, d_poses, d_map);
        if (device_time_s) hipEventRecord(e1, nullptr);
        FE_TRY(hipDeviceSynchronize());
        FE_TRY(hipGetLastError());
        float ms = 0.f;
        if (device_time_s && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *device_time_s = 1e-3 * ms;
    }
    }
    {
        const size_t down = up_bytes - o_status;
        FE_TRY(hipMemcpy(host.data() + o_status, arena + o_status, down, hipMemcpyDeviceToHost));
        memcpy(&status, host.data() + o_status, sizeof(int));
        if (match_count) memcpy(match_count, host.data() + o_mcnt, num_pairs * sizeof(uint32_t));
        if (inlier_count) memcpy(inlier_count, host.data() + o_best, num_pairs * sizeof(uint32_t));
        if (status == 1) { g_fe_error = "a pair of consecutive states has fewer than three matches"; rc = SSBA_ERR_NUMERICAL_FAILURE; goto done; }
        if (status == 2) { g_fe_error = "random stream exhausted"; rc = SSBA_ERR_NUMERICAL_FAILURE; goto done; }
        memcpy(poses, host.data() + o_poses, (size_t)num_states * 12 * sizeof(double));
        memcpy(map_points, host.data() + o_map, (size_t)num_points * 3 * sizeof(double));
        memcpy(initialized, host.data() + o_init, num_points);
    }
done:
#undef FE_TRY
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (arena) { (void)hipDeviceSynchronize(); ssba::pool_free(arena); }
    return rc;
}

}  // extern "C"

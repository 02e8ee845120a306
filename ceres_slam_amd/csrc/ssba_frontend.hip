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

#include <string>
#include <vector>

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
    const double n1 = sqrt(dot3d(u1, u1));
    for (int r = 0; r < 3; ++r) u1[r] /= n1;
    const double d12 = dot3d(u1, u2);
    for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
    const double n2 = sqrt(dot3d(u2, u2));
    for (int r = 0; r < 3; ++r) u2[r] /= n2;
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
    const uint32_t it = blockIdx.x, pair = blockIdx.y, base = offset[pair], n = offset[pair + 1] - base;
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
        hipLaunchKernelGGL(k_fe_score, dim3(num_iters, num_pairs), dim3(256), 0, nullptr, cam, d_off, d_p0, d_p1, d_smp, num_iters, thresh, d_cnt);
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

}  // extern "C"

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
#include "ssba_phong_device.h"

namespace ssba {

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

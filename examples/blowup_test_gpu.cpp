// blowup_test_gpu -- the reference's pose-graph consistency experiment (/root/reference tests/blowup_test.cpp:30-130)
// written against include/ceres_slam_amd/ceres_shim.hpp: a chain of poses linked by RelativePoseErrorAutomatic blocks,
// solved two states at a time; the covariance of the second state becomes the PoseErrorAutomatic prior of the next
// window (here: the prior stiffness of the first window is given, the measurement is the true relative pose plus a
// fixed perturbation, so the result is deterministic and checked by tests/test_gpu_pose_factors.py).
//
// usage: blowup_test_gpu <num_poses> <meas_sigma>
// Output: one line per state "k tx ty tz trace(cov)".
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "ceres_slam_amd/ceres_shim.hpp"

static void compose(const double *A, const double *B, double *C) {      // C = A * B on [t | R row-major] blocks
    for (int i = 0; i < 3; ++i) {
        C[i] = A[3 + 3 * i] * B[0] + A[4 + 3 * i] * B[1] + A[5 + 3 * i] * B[2] + A[i];
        for (int j = 0; j < 3; ++j) C[3 + 3 * i + j] = A[3 + 3 * i] * B[3 + j] + A[4 + 3 * i] * B[6 + j] + A[5 + 3 * i] * B[9 + j];
    }
}

int main(int argc, char **argv) {
    const int num_poses = argc > 1 ? std::atoi(argv[1]) : 10;
    const double sigma = argc > 2 ? std::atof(argv[2]) : 0.1;
    // measurement: 1 m forward with a small yaw, T_2_1 (tests/blowup_test.cpp:34-40)
    const double yaw = 0.05;
    const double meas[12] = {0.0, 0.0, -1.0, std::cos(yaw), 0, std::sin(yaw), 0, 1, 0, -std::sin(yaw), 0, std::cos(yaw)};
    double meas_stiffness[36] = {0}, prior_stiffness[36] = {0};
    for (int c = 0; c < 6; ++c) { meas_stiffness[7 * c] = 1.0 / sigma; prior_stiffness[7 * c] = 1e6; }      // covars[0] = 1e-12 I (:50)
    std::vector<double> T(12 * (size_t)num_poses, 0.0);
    for (int k = 0; k < num_poses; ++k) T[12 * k + 3] = T[12 * k + 7] = T[12 * k + 11] = 1.0;
    std::cout.precision(17);
    std::cout << 0 << " " << T[0] << " " << T[1] << " " << T[2] << " " << 6e-12 << std::endl;
    ceres::LocalParameterization *se3 = nullptr;
    for (int k1 = 0; k1 + 1 < num_poses; ++k1) {
        double *T1 = &T[12 * k1], *T2 = &T[12 * (k1 + 1)];
        compose(meas, T1, T2);                                        // initial guess: chain the measurement (:60-62)
        T2[0] += 0.02; T2[2] -= 0.03;                                 // ... and disturb it, so that the solve has something to do
        ceres::Problem problem;
        se3 = ceres_slam::SE3Perturbation::Create();
        problem.AddResidualBlock(ceres_slam::RelativePoseErrorAutomatic::Create(meas, meas_stiffness), NULL, T1, T2);      // :70-76
        problem.AddResidualBlock(ceres_slam::PoseErrorAutomatic::Create(T1, prior_stiffness), NULL, T1);                   // :84-87
        problem.SetParameterization(T1, se3);
        problem.SetParameterization(T2, se3);
        ceres::Solver::Options options;
        options.max_num_iterations = 1000;
        ceres::Solver::Summary summary;
        ceres::Solve(options, &problem, &summary);
        if (!summary.IsSolutionUsable()) { std::cerr << summary.message << std::endl; return EXIT_FAILURE; }
        ceres::Covariance::Options copt;
        ceres::Covariance covariance(copt);
        std::vector<std::pair<const double *, const double *>> blocks(1, std::make_pair((const double *)T2, (const double *)T2));
        double cov[36];
        if (!covariance.Compute(blocks, &problem) || !covariance.GetCovarianceBlockInTangentSpace(T2, T2, cov)) {
            std::cerr << "covariance failed: " << covariance.message() << std::endl;
            return EXIT_FAILURE;
        }
        double tr = 0.0;
        for (int c = 0; c < 6; ++c) tr += cov[7 * c];
        std::cout << k1 + 1 << " " << T2[0] << " " << T2[1] << " " << T2[2] << " " << tr << std::endl;
        // next window's prior: inverse square root of this covariance -- diagonal here up to rounding, keep it simple
        for (int c = 0; c < 36; ++c) prior_stiffness[c] = 0.0;
        for (int c = 0; c < 6; ++c) prior_stiffness[7 * c] = 1.0 / std::sqrt(cov[7 * c]);
    }
    return EXIT_SUCCESS;
}

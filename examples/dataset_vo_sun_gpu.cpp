// dataset_vo_sun_gpu -- the reference's sun-aided sliding-window VO driver (/root/reference tests/dataset_vo_sun.cpp)
// written against include/ceres_slam_amd/ceres_shim.hpp and the DatasetProblemSun container of
// include/ceres_slam_amd/dataset_problem_sun.hpp: the same Ceres calls, executed by the MI355X back end.
//
// usage: dataset_vo_sun_gpu <track_file> <ref_sun_file> <obs_sun_file> [--window (2)] [--huber-param (0)]
//                           [--az-err-thresh (1000)] [--zen-err-thresh (1000)] [--sun-only]
//   track_file    DatasetProblemSun::read_csv (src/ceres_slam/dataset_problem_sun.cpp:16-116): row 1
//                 "num_states,num_points", row 2 "fu,fv,cu,cv,b", row 3 first pose (4x4 row-major), then
//                 "k,j,u,v,d,c00..c22" rows (observation + its 3x3 covariance)
//   ref_sun_file  "k,e,n,u": expected sun direction of state k in the global (ENU) frame (:118-141)
//   obs_sun_file  "k,x,y,z,c00,c01,c10,c11": observed sun direction in the camera frame + azimuth/zenith covariance (:143-170)
// Per window [k1, k2) (main, :267-311): initial guess by 3-point RANSAC between consecutive states
// (compute_initial_guess(k1, k2), dataset_problem_sun.cpp:250-354; the RANSAC of the window's pairs runs in one GPU
// batch), then solveWindow (:28-185): stereo blocks with the per-point stiffness, sun blocks (HuberLoss when
// --huber-param), a PoseErrorAutomatic prior on the first state from the previous window's covariance, no constant
// state, DOGLEG / SUBSPACE_DOGLEG, and ceres::Covariance of state k1+1 for the next window's prior.
// Two passes as in the reference: without sun blocks (written to <track>_poses.csv), then with them
// (<track>_<last '_' token of obs_sun_file>_poses.csv).  Poses are written at full precision.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "ceres_slam_amd/ceres_shim.hpp"
#include "ceres_slam_amd/dataset_problem_sun.hpp"

// SSBA_DRIVER_TIMING=1: wall time spent in the three stages of a window, printed at exit
static double g_time[3] = {0, 0, 0};
struct StageTimer {
    int stage;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(int s) : stage(s), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { g_time[stage] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// solveWindow (tests/dataset_vo_sun.cpp:28-185)
static void solveWindow(ceres_slam::DatasetProblemSun &dataset, ceres_slam::uint k1, ceres_slam::uint k2, bool use_sun, double huber_param,
                        double az_err_thresh, double zen_err_thresh) {
    std::cerr << "Working on interval [" << k1 << "," << k2 << ")/" << dataset.num_states << ": ";
    ceres::Problem problem;
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    for (ceres_slam::uint k = k1; k < k2; ++k) {
        for (ceres_slam::uint i : dataset.obs_indices_at_state(k)) {
            const ceres_slam::uint j = dataset.point_ids[i];
            if (!dataset.initialized_point[j]) continue;                        // :55
            double stereo_obs_stiffness[9];
            ceres_slam::inverse_sqrt_symmetric(3, dataset.stereo_obs_covars[j].data(), stereo_obs_stiffness);     // stereo_obs_covars[j]  (:58-60)
            ceres::CostFunction *stereo_cost =
                ceres_slam::StereoReprojectionErrorAutomatic::Create(dataset.camera, dataset.stereo_obs_list[i].data(), stereo_obs_stiffness);
            problem.AddResidualBlock(stereo_cost, NULL, dataset.poses[k].data(), dataset.map_points[j].data());
        }
        if (use_sun && dataset.state_has_sun_obs[k]) {                          // :77-106
            double sun_obs_stiffness[4];
            ceres_slam::inverse_sqrt_symmetric(2, dataset.sun_obs_covars[k].data(), sun_obs_stiffness);
            ceres::CostFunction *sun_cost = ceres_slam::SunSensorErrorAutomatic::Create(dataset.sun_obs_list[k].data(), dataset.sun_dir_g[k].data(),
                                                                                         sun_obs_stiffness, az_err_thresh, zen_err_thresh);
            problem.AddResidualBlock(sun_cost, huber_param > 0. ? new ceres::HuberLoss(huber_param) : NULL, dataset.poses[k].data());
        }
    }
    double pose_prior_stiffness[36];                                            // :111-124
    ceres_slam::inverse_sqrt_symmetric(6, dataset.pose_covars[k1].data(), pose_prior_stiffness);
    ceres::CostFunction *pose_prior_cost = ceres_slam::PoseErrorAutomatic::Create(dataset.poses[k1].data(), pose_prior_stiffness);
    problem.AddResidualBlock(pose_prior_cost, NULL, dataset.poses[k1].data());
    for (ceres_slam::uint k = k1; k < k2; ++k) problem.SetParameterization(dataset.poses[k].data(), se3_perturbation);

    ceres::Solver::Options solver_options;                                      // :135-145
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    solver_options.trust_region_strategy_type = ceres::DOGLEG;
    solver_options.dogleg_type = ceres::SUBSPACE_DOGLEG;
    ceres::Solver::Summary summary;
    {
        StageTimer timer(1);
        Solve(solver_options, &problem, &summary);
    }
    std::cout << summary.BriefReport() << std::endl;
    if (!summary.message.empty()) std::cerr << summary.message << std::endl;

    ceres::Covariance::Options covariance_options;                              // :159-183
    covariance_options.num_threads = solver_options.num_threads;
    covariance_options.sparse_linear_algebra_library_type = ceres::SUITE_SPARSE;
    covariance_options.algorithm_type = ceres::SPARSE_QR;
    ceres::Covariance covariance(covariance_options);
    std::vector<std::pair<const double *, const double *>> covar_blocks;
    covar_blocks.push_back(std::make_pair(dataset.poses[k1 + 1].data(), dataset.poses[k1 + 1].data()));
    StageTimer timer(2);
    if (!covariance.Compute(covar_blocks, &problem)) {
        std::cout << "WARNING: Covariance computation failed! Using previous state covariance." << std::endl;
        dataset.pose_covars[k1 + 1] = dataset.pose_covars[k1];
    } else {
        covariance.GetCovarianceBlockInTangentSpace(dataset.poses[k1 + 1].data(), dataset.poses[k1 + 1].data(), dataset.pose_covars[k1 + 1].data());
    }
}

// tests/dataset_vo_sun.cpp:187-324
int main(int argc, char **argv) {
    const std::string usage("usage: dataset_vo_sun_gpu <track_file> <ref_sun_file> <obs_sun_file> [--window (2)] [--huber-param (0)] "
                            "[--az-err-thresh (1000)] [--zen-err-thresh (1000)] [--sun-only]");
    if (argc < 4) { std::cerr << usage << std::endl; return EXIT_FAILURE; }
    ceres_slam::uint window_size = 2;
    bool sun_only = false;
    double huber_param = 0., az_err_thresh = 1000., zen_err_thresh = 1000.;
    const double pi = 3.14159265358979323846;
    for (int a = 4; a < argc; ++a) {
        const std::string flag(argv[a]);
        if (flag == "--window" && argc > a + 1) window_size = (ceres_slam::uint)std::stoi(argv[++a]);
        else if (flag == "--huber-param" && argc > a + 1) huber_param = std::stod(argv[++a]);
        else if (flag == "--az-err-thresh" && argc > a + 1) az_err_thresh = std::stod(argv[++a]) * pi / 180.;     // degrees on the command line
        else if (flag == "--zen-err-thresh" && argc > a + 1) zen_err_thresh = std::stod(argv[++a]) * pi / 180.;
        else if (flag == "--sun-only") sun_only = true;
        else { std::cerr << usage << std::endl; return EXIT_FAILURE; }
    }
    ceres_slam::DatasetProblemSun dataset;
    if (!dataset.read_csv(argv[1], argv[2], argv[3])) return EXIT_FAILURE;
    for (int a = 4; a < argc; ++a)          // --refprecision: the reference's four significant digits (utils/utils.hpp:34) instead of 17
        if (std::string(argv[a]) == "--refprecision") dataset.csv_precision = ceres_slam::DatasetProblemSun::kReferenceCsvPrecision;
    if (window_size == 0 || window_size > dataset.num_states) window_size = dataset.num_states;      // 0 = full batch (:259-262)

    auto run_pass = [&](bool use_sun) {                                             // :267-283 / :295-311
        for (ceres_slam::uint k1 = 0; k1 + window_size <= dataset.num_states; ++k1) {
            const ceres_slam::uint k2 = std::min(k1 + window_size, dataset.num_states);
            bool guess;
            {
                StageTimer timer(0);
                guess = dataset.compute_initial_guess(k1, k2);
            }
            if (guess) {
                solveWindow(dataset, k1, k2, use_sun, huber_param, az_err_thresh, zen_err_thresh);
            } else {
                std::cerr << "WARNING: Initial guess failed. Copying previous pose and covariance." << std::endl;
                dataset.poses[k2 - 1] = dataset.poses[k1];
                dataset.pose_covars[k2 - 1] = dataset.pose_covars[k1];
            }
            dataset.reset_points();
        }
    };
    std::string track(argv[1]), base = track.substr(0, track.find('.'));
    if (!sun_only) {
        std::cerr << "Computing VO without sun measurements" << std::endl;
        run_pass(false);
        if (!dataset.write_csv(base)) return EXIT_FAILURE;
    }
    std::cerr << "Computing VO with sun measurements" << std::endl;
    run_pass(true);
    std::string obs_sun(argv[3]);                                                   // :314-321
    obs_sun = obs_sun.substr(0, obs_sun.find('.'));
    const size_t us = obs_sun.find_last_of('_');
    const std::string tag = us == std::string::npos ? obs_sun : obs_sun.substr(us + 1);
    if (std::getenv("SSBA_DRIVER_TIMING"))
        std::cerr << "stage seconds: initial guess " << g_time[0] << ", Solve " << g_time[1] << ", Covariance " << g_time[2] << std::endl;
    return dataset.write_csv(base + "_" + tag) ? EXIT_SUCCESS : EXIT_FAILURE;
}

// dataset_vo_gpu -- the solve stage of the reference's stereo driver
// (/root/reference tests/dataset_vo.cpp:22-85, solveWindow) written against
// include/ceres_slam_amd/ceres_shim.hpp, i.e. the same calls the reference makes against
// Ceres, executed by the MI355X back end.
//
// usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N]
//        dataset_vo_gpu <dataset.csv> --frontend [--huber A] [--window N]
//   dataset.csv   reference format (src/ceres_slam/dataset_problem.cpp:16-83): row 1
//                 "num_states,num_points", row 2 intrinsics "fu,fv,cu,cv,b", row 3 variances,
//                 row 4 first pose (4x4 row-major), then "k,j,u,v,d" rows
//   init_*.csv    initial guess in the format the reference's write_csv emits
//                 (dataset_problem.cpp:144-159): header + 4x4 row-major poses; "id,x,y,z" points
// --frontend computes the initial guess itself window by window, as the reference's main does (tests/dataset_vo.cpp:121-127,
// DatasetProblem::compute_initial_guess, src/ceres_slam/dataset_problem.cpp:179-270): reciprocal matches of
// consecutive states, triangulation, the 400-iteration 3-point RANSAC of ALL state pairs in one GPU batch
// (ssba_frontend_ransac; the draw sequence of std::mt19937(42) + std::uniform_int_distribution is
// restated by ssba_ransac_samples), pose chaining and map initialisation from the inliers.  --window N runs the reference's sliding
// window loop (tests/dataset_vo.cpp:121-127): states [k1, k1+N) per solve, first state of the window constant, poses
// carried over; with --frontend every window recomputes its initial guess from the state the previous window left
// (compute_initial_guess(k1, k2), then reset_points(); the map file holds the points of the last window), with
// initial-guess files the points are reset to the supplied guess between windows.
// Output: <dataset>_poses.csv / <dataset>_map.csv at full precision + the brief report.
#include <cmath>
#include <iostream>

#include "ceres_slam_amd/ceres_shim.hpp"
#include "ceres_slam_amd/dataset_problem.hpp"

// tests/dataset_vo.cpp:22-85
static bool solveWindow(ceres_slam::DatasetProblem &dataset, ceres_slam::uint k1, ceres_slam::uint k2, double huber) {
    std::cerr << "Working on interval [" << k1 << "," << k2 << ")" << std::endl;
    ceres::Problem problem;
    // stiffness = Sigma^-1/2 of the (diagonal) observation covariance (:29-32)
    const double *var = dataset.stereo_obs_var.data();
    const double stereo_obs_stiffness[9] = {1.0 / std::sqrt(var[0]), 0, 0, 0, 1.0 / std::sqrt(var[1]), 0, 0, 0, 1.0 / std::sqrt(var[2])};
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    for (ceres_slam::uint k = k1; k < k2; ++k) {
        bool used = false;
        for (ceres_slam::uint i : dataset.obs_indices_at_state(k)) {
            const ceres_slam::uint j = dataset.point_ids[i];
            if (j >= dataset.num_points || !dataset.initialized_point[j]) continue;     // only initialised map points (:45)
            ceres::CostFunction *stereo_cost =
                ceres_slam::StereoReprojectionErrorAutomatic::Create(dataset.camera, dataset.stereo_obs_list[i].data(), stereo_obs_stiffness);
            problem.AddResidualBlock(stereo_cost, huber > 0 ? new ceres::HuberLoss(huber) : NULL, dataset.poses[k].data(),
                                     dataset.map_points[j].data());
            used = true;
        }
        if (used) problem.SetParameterization(dataset.poses[k].data(), se3_perturbation);
    }
    problem.SetParameterBlockConstant(dataset.poses[k1].data());                        // :62

    ceres::Solver::Options solver_options;
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.num_threads = 8;
    solver_options.num_linear_solver_threads = 8;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    ceres::Solver::Summary summary;
    ceres::Solve(solver_options, &problem, &summary);
    std::cout << summary.BriefReport() << std::endl << std::endl;
    if (!summary.message.empty()) std::cerr << summary.message << std::endl;
    return summary.IsSolutionUsable();
}

// tests/dataset_vo.cpp:87-138
int main(int argc, char **argv) {
    const bool use_frontend = argc >= 3 && std::string(argv[2]) == "--frontend";
    if (argc < 4 && !use_frontend) {
        std::cerr << "usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N]\n"
                     "       dataset_vo_gpu <dataset.csv> --frontend [--huber A] [--window N]" << std::endl;
        return EXIT_FAILURE;
    }
    double huber = 0.0;
    ceres_slam::uint window_size = 0;
    for (int a = use_frontend ? 3 : 4; a + 1 < argc; ++a) {
        if (std::string(argv[a]) == "--huber") huber = std::atof(argv[a + 1]);
        if (std::string(argv[a]) == "--window") window_size = (ceres_slam::uint)std::atoi(argv[a + 1]);
    }
    const std::string filename(argv[1]);
    ceres_slam::DatasetProblem dataset;
    if (!dataset.read_csv(filename)) return EXIT_FAILURE;
    if (!use_frontend && !dataset.read_initial_guess(argv[2], argv[3])) return EXIT_FAILURE;
    if (window_size == 0 || window_size > dataset.num_states) window_size = dataset.num_states;     // 0 = full batch (:117-119)
    const std::vector<ceres_slam::Point> supplied_points(dataset.map_points);
    bool usable = true;
    for (ceres_slam::uint k1 = 0; k1 + window_size <= dataset.num_states; ++k1) {                   // :121-127
        const ceres_slam::uint k2 = k1 + window_size;
        if (use_frontend) {
            if (!dataset.compute_initial_guess(k1, k2)) return EXIT_FAILURE;
        } else if (k1 > 0) {
            dataset.map_points = supplied_points;          // the supplied guess again: the role reset_points() + VO play in the reference
        }
        usable = solveWindow(dataset, k1, k2, huber);
        if (!usable) break;
        // reset_points() (:126).  The reference also resets after the LAST window and so writes an empty map file; the
        // points of the last window are kept here
        if (use_frontend && k1 + window_size < dataset.num_states) dataset.reset_points();
    }
    dataset.write_csv(filename);
    return usable ? EXIT_SUCCESS : EXIT_FAILURE;
}

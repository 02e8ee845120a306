// dataset_vo_gpu -- the solve stage of the reference's stereo driver
// (/root/reference tests/dataset_vo.cpp:22-85, solveWindow) written against
// include/ceres_slam_amd/ceres_shim.hpp, i.e. the same calls the reference makes against
// Ceres, executed by the MI355X back end.
//
// usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N]
//   dataset.csv   reference format (src/ceres_slam/dataset_problem.cpp:16-83): row 1
//                 "num_states,num_points", row 2 intrinsics "fu,fv,cu,cv,b", row 3 variances,
//                 row 4 first pose (4x4 row-major), then "k,j,u,v,d" rows
//   init_*.csv    initial guess in the format the reference's write_csv emits
//                 (dataset_problem.cpp:144-159): header + 4x4 row-major poses; "id,x,y,z" points
// The front end that produces the initial guess (compute_initial_guess: matching + RANSAC)
// is SURVEY.md section 8(f) row N2 and not part of this path.  --window N runs the reference's sliding
// window loop (tests/dataset_vo.cpp:121-127): states [k1, k1+N) per solve, first state of the window
// constant, poses carried over, points reset between windows (reset_points) to the supplied guess
// (the reference re-triangulates them in compute_initial_guess(k1, k2)).
// Output: <dataset>_poses.csv / <dataset>_map.csv at full precision + the brief report.
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ceres_slam_amd/ceres_shim.hpp"

static std::vector<double> parse_row(const std::string &line) {
    std::vector<double> v;
    std::stringstream ss(line);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        try { v.push_back(std::stod(tok)); } catch (...) { v.clear(); return v; }
    }
    return v;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N]" << std::endl;
        return EXIT_FAILURE;
    }
    double huber = 0.0;
    size_t window_size = 0;
    for (int a = 4; a + 1 < argc; ++a) {
        if (std::string(argv[a]) == "--huber") huber = std::atof(argv[a + 1]);
        if (std::string(argv[a]) == "--window") window_size = (size_t)std::atoi(argv[a + 1]);
    }
    std::ifstream f(argv[1]);
    if (!f.is_open()) { std::cerr << "Error: couldn't open " << argv[1] << std::endl; return EXIT_FAILURE; }
    std::string line;
    std::getline(f, line); auto meta = parse_row(line);
    std::getline(f, line); auto intr = parse_row(line);
    std::getline(f, line); auto var = parse_row(line);
    std::getline(f, line);   // first ground-truth pose: the initial guess file carries it
    if (meta.size() < 2 || intr.size() < 5 || var.size() < 3) { std::cerr << "malformed header" << std::endl; return EXIT_FAILURE; }
    const size_t num_states = (size_t)meta[0], num_points = (size_t)meta[1];
    std::vector<unsigned> state_ids, point_ids;
    std::vector<double> obs;
    while (std::getline(f, line)) {
        auto r = parse_row(line);
        if (r.size() < 5) continue;
        state_ids.push_back((unsigned)r[0]); point_ids.push_back((unsigned)r[1]);
        obs.insert(obs.end(), r.begin() + 2, r.begin() + 5);
    }
    // poses: 12 doubles [t | R row-major] per state (geometry/se3group.hpp:425-429)
    std::vector<double> poses(num_states * 12, 0.0), points(num_points * 3, 0.0);
    std::vector<bool> initialized(num_points, false);
    {
        std::ifstream pf(argv[2]);
        if (!pf.is_open()) { std::cerr << "Error: couldn't open " << argv[2] << std::endl; return EXIT_FAILURE; }
        size_t k = 0;
        while (std::getline(pf, line) && k < num_states) {
            auto r = parse_row(line);
            if (r.size() < 16) continue;   // header
            double *T = &poses[12 * k++];
            T[0] = r[3]; T[1] = r[7]; T[2] = r[11];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 + 3 * i + j] = r[4 * i + j];
        }
        std::ifstream mf(argv[3]);
        if (!mf.is_open()) { std::cerr << "Error: couldn't open " << argv[3] << std::endl; return EXIT_FAILURE; }
        while (std::getline(mf, line)) {
            auto r = parse_row(line);
            if (r.size() < 4 || (size_t)r[0] >= num_points) continue;
            const size_t j = (size_t)r[0];
            for (int c = 0; c < 3; ++c) points[3 * j + c] = r[1 + c];
            initialized[j] = true;
        }
    }

    if (window_size == 0 || window_size > num_states) window_size = num_states;     // 0 = full batch (:117-119)
    const std::vector<double> points_init(points);
    ceres::Solver::Summary summary;
    for (size_t k1 = 0; k1 + window_size <= num_states; ++k1) {                     // :121-127
    const size_t k2 = k1 + window_size;
    if (k1 > 0) points = points_init;                                               // reset_points()
    // ---- solveWindow (tests/dataset_vo.cpp:22-85) ------------------------------------------
    ceres::Problem problem;
    const double stiffness[9] = {1.0 / std::sqrt(var[0]), 0, 0, 0, 1.0 / std::sqrt(var[1]), 0, 0, 0, 1.0 / std::sqrt(var[2])};
    auto camera = std::make_shared<const ceres_slam::StereoCamera>(intr[0], intr[1], intr[2], intr[3], intr[4]);
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    std::vector<bool> pose_used(num_states, false);
    for (size_t i = 0; i < state_ids.size(); ++i) {
        const unsigned k = state_ids[i], j = point_ids[i];
        if (k < k1 || k >= k2 || j >= num_points || !initialized[j]) continue;    // states of the window, initialised map points
        ceres::CostFunction *stereo_cost = ceres_slam::StereoReprojectionErrorAutomatic::Create(camera, &obs[3 * i], stiffness);
        problem.AddResidualBlock(stereo_cost, huber > 0 ? new ceres::HuberLoss(huber) : NULL, &poses[12 * k], &points[3 * j]);
        pose_used[k] = true;
    }
    for (size_t k = 0; k < num_states; ++k)
        if (pose_used[k]) problem.SetParameterization(&poses[12 * k], se3_perturbation);
    problem.SetParameterBlockConstant(&poses[12 * k1]);                            // :62

    ceres::Solver::Options solver_options;
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.num_threads = 8;
    solver_options.num_linear_solver_threads = 8;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    ceres::Solve(solver_options, &problem, &summary);
    std::cout << summary.BriefReport() << std::endl << std::endl;
    if (!summary.message.empty()) std::cerr << summary.message << std::endl;
    if (!summary.IsSolutionUsable()) break;
    }   // windows

    // ---- write_csv (dataset_problem.cpp:121-165), full precision ----------------------------
    std::string base(argv[1]);
    base = base.substr(0, base.find_last_of('.'));
    std::ofstream po(base + "_poses.csv"), mo(base + "_map.csv");
    po.precision(17); mo.precision(17);
    po << "T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33" << std::endl;
    for (size_t k = 0; k < num_states; ++k) {
        const double *T = &poses[12 * k];
        for (int i = 0; i < 3; ++i) po << T[3 + 3 * i] << "," << T[4 + 3 * i] << "," << T[5 + 3 * i] << "," << T[i] << ",";
        po << "0,0,0,1" << std::endl;
    }
    mo << "point_id, x, y, z" << std::endl;
    for (size_t j = 0; j < num_points; ++j)
        if (initialized[j]) mo << j << "," << points[3 * j] << "," << points[3 * j + 1] << "," << points[3 * j + 2] << std::endl;
    return summary.IsSolutionUsable() ? EXIT_SUCCESS : EXIT_FAILURE;
}

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
#include <fstream>
#include <iostream>
#include <map>
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
    const bool use_frontend = argc >= 3 && std::string(argv[2]) == "--frontend";
    if (argc < 4 && !use_frontend) {
        std::cerr << "usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N]\n"
                     "       dataset_vo_gpu <dataset.csv> --frontend [--huber A] [--window N]" << std::endl;
        return EXIT_FAILURE;
    }
    double huber = 0.0;
    size_t window_size = 0;
    for (int a = use_frontend ? 3 : 4; a + 1 < argc; ++a) {
        if (std::string(argv[a]) == "--huber") huber = std::atof(argv[a + 1]);
        if (std::string(argv[a]) == "--window") window_size = (size_t)std::atoi(argv[a + 1]);
    }
    std::ifstream f(argv[1]);
    if (!f.is_open()) { std::cerr << "Error: couldn't open " << argv[1] << std::endl; return EXIT_FAILURE; }
    std::string line;
    std::getline(f, line); auto meta = parse_row(line);
    std::getline(f, line); auto intr = parse_row(line);
    std::getline(f, line); auto var = parse_row(line);
    std::getline(f, line); auto first_pose = parse_row(line);   // first ground-truth pose (4x4 row-major)
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
    // ---- DatasetProblem::compute_initial_guess(k1, k2) (src/ceres_slam/dataset_problem.cpp:179-270) ----------
    std::vector<std::vector<unsigned>> idx_of(num_states);
    for (size_t i = 0; i < state_ids.size(); ++i)
        if (state_ids[i] < num_states) idx_of[state_ids[i]].push_back((unsigned)i);
    auto triangulate = [&](unsigned i, double *p) {                      // stereo_camera.hpp:112-120
        const double b_over_d = intr[4] / obs[3 * i + 2];
        p[0] = (obs[3 * i] - intr[2]) * b_over_d;
        p[1] = (obs[3 * i + 1] - intr[3]) * b_over_d * (intr[0] / intr[1]);
        p[2] = intr[0] * b_over_d;
    };
    auto compute_initial_guess = [&](size_t k1, size_t k2) -> bool {
        if (k2 <= k1 + 1) return true;
        std::vector<uint32_t> offset(1, 0), samples;
        std::vector<double> pts0, pts1;
        std::vector<unsigned> match_km1;                                     // observation index in state k-1 of every match
        const uint32_t num_iters = 400;
        for (size_t k = k1 + 1; k < k2; ++k) {                               // :189-243
            std::vector<unsigned> a, b;
            std::map<unsigned, unsigned> in_k;
            for (unsigned i : idx_of[k]) in_k[point_ids[i]] = i;
            std::map<unsigned, int> kept;
            for (unsigned i : idx_of[k - 1]) if (in_k.count(point_ids[i])) { a.push_back(i); kept[point_ids[i]] = 1; }
            for (unsigned i : idx_of[k]) if (kept.count(point_ids[i])) b.push_back(i);
            if (a.size() < 3 || a.size() != b.size()) { std::cerr << "state " << k << ": fewer than 3 matches" << std::endl; return false; }
            for (size_t m = 0; m < a.size(); ++m) {
                double p[3];
                triangulate(a[m], p); pts0.insert(pts0.end(), p, p + 3);
                triangulate(b[m], p); pts1.insert(pts1.end(), p, p + 3);
                match_km1.push_back(a[m]);
            }
            offset.push_back((uint32_t)(pts0.size() / 3));
            std::vector<uint32_t> smp(3 * num_iters);
            if (ssba_ransac_samples((uint32_t)a.size(), num_iters, __GNUC__ >= 11 ? 1 : 0, smp.data())) return false;
            samples.insert(samples.end(), smp.begin(), smp.end());
        }
        const uint32_t num_pairs = (uint32_t)(k2 - k1 - 1);
        std::vector<double> T((size_t)num_pairs * 12);
        std::vector<uint8_t> inlier(pts0.size() / 3);
        ssba_camera cam = {intr[0], intr[1], intr[2], intr[3], intr[4]};
        int rc = ssba_frontend_ransac(&cam, -1, num_pairs, offset.data(), pts0.data(), pts1.data(), samples.data(), num_iters, 4.0,
                                      T.data(), inlier.data(), nullptr, nullptr);                     // :246-249
        if (rc) { std::cerr << "ssba_frontend_ransac: " << ssba_status_string(rc) << std::endl; return false; }
        for (size_t k = k1 + 1; k < k2; ++k) {
            const size_t q = k - k1 - 1;
            const double *Tk = &T[12 * q], *Tp = &poses[12 * (k - 1)];
            double *Tn = &poses[12 * k];
            for (int i = 0; i < 3; ++i) {                                    // poses[k] = T_k_km1 * poses[k-1]  (:256)
                Tn[i] = Tk[3 + 3 * i] * Tp[0] + Tk[4 + 3 * i] * Tp[1] + Tk[5 + 3 * i] * Tp[2] + Tk[i];
                for (int j = 0; j < 3; ++j) Tn[3 + 3 * i + j] = Tk[3 + 3 * i] * Tp[3 + j] + Tk[4 + 3 * i] * Tp[6 + j] + Tk[5 + 3 * i] * Tp[9 + j];
            }
            for (uint32_t m = offset[q]; m < offset[q + 1]; ++m) {           // :260-269
                const unsigned j = point_ids[match_km1[m]];
                if (!inlier[m] || j >= num_points || initialized[j]) continue;
                const double d[3] = {pts0[3 * m] - Tp[0], pts0[3 * m + 1] - Tp[1], pts0[3 * m + 2] - Tp[2]};
                for (int c = 0; c < 3; ++c) points[3 * j + c] = Tp[3 + c] * d[0] + Tp[6 + c] * d[1] + Tp[9 + c] * d[2];   // poses[k-1]^-1 * p
                initialized[j] = true;
            }
        }
        return true;
    };
    if (use_frontend) {
        if (first_pose.size() < 16) { std::cerr << "malformed first pose" << std::endl; return EXIT_FAILURE; }
        poses[0] = first_pose[3]; poses[1] = first_pose[7]; poses[2] = first_pose[11];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) poses[3 + 3 * i + j] = first_pose[4 * i + j];
    } else {
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
    if (use_frontend) {                                                             // main loop of tests/dataset_vo.cpp:121-127
        if (!compute_initial_guess(k1, k2)) return EXIT_FAILURE;                    //   compute_initial_guess(k1, k2)
    } else if (k1 > 0) {
        points = points_init;                                                       // reset_points() to the supplied guess
    }
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
    if (use_frontend && k1 + window_size < num_states) initialized.assign(num_points, false);      // reset_points() (:126)
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

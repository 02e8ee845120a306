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
// --gpus N (full batch only): N processes, one per GPU, each holding the landmarks j with j * N / num_points == rank and
// all their observations; the reduced camera systems of the shards are summed by ncclAllReduce (RCCL over xGMI) inside
// libssba.so at every iteration, no host code in the loop (include/ssba.h: ssba_set_distributed, ssba_set_rccl).  The
// program forks its N - 1 peers itself before anything touches a GPU (rank r sees GPU r through HIP_VISIBLE_DEVICES),
// rank 0 hands out the RCCL id and collects the map points of the other ranks through pipes, and writes the output.
// --gpus 1 runs the same path with a communicator of one rank.
// Output: <dataset>_poses.csv / <dataset>_map.csv at full precision + the brief report.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <sstream>
#include <cstdint>
#include <iostream>

#include "ceres_slam_amd/ceres_shim.hpp"
#include "ceres_slam_amd/dataset_problem.hpp"

// landmark sharding of --gpus N
static int g_world = 0, g_rank = 0;
static unsigned char g_rccl_id[SSBA_RCCL_UNIQUE_ID_BYTES];
static bool owned(ceres_slam::uint j, ceres_slam::uint num_points) {
    return g_world <= 1 || (int)((uint64_t)j * (uint64_t)g_world / num_points) == g_rank;
}

// tests/dataset_vo.cpp:22-85
static bool solveWindow(ceres_slam::DatasetProblem &dataset, ceres_slam::uint k1, ceres_slam::uint k2, double huber) {
    if (g_rank == 0) std::cerr << "Working on interval [" << k1 << "," << k2 << ")" << std::endl;
    ceres::Problem problem;
    if (g_world > 0) problem.SetDistributed(g_world, g_rank, g_rccl_id);
    // stiffness = Sigma^-1/2 of the (diagonal) observation covariance (:29-32)
    const double *var = dataset.stereo_obs_var.data();
    const double stereo_obs_stiffness[9] = {1.0 / std::sqrt(var[0]), 0, 0, 0, 1.0 / std::sqrt(var[1]), 0, 0, 0, 1.0 / std::sqrt(var[2])};
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    for (ceres_slam::uint k = k1; k < k2; ++k) {
        bool used = false;
        if (g_world > 0) {      // a shard: the pose blocks of all states with observations, on every rank, in the same order
            for (ceres_slam::uint i : dataset.obs_indices_at_state(k)) {
                const ceres_slam::uint j = dataset.point_ids[i];
                if (j < dataset.num_points && dataset.initialized_point[j]) { used = true; break; }
            }
            if (used) problem.AddParameterBlock(dataset.poses[k].data(), 12);
        }
        for (ceres_slam::uint i : dataset.obs_indices_at_state(k)) {
            const ceres_slam::uint j = dataset.point_ids[i];
            if (j >= dataset.num_points || !dataset.initialized_point[j]) continue;     // only initialised map points (:45)
            if (!owned(j, dataset.num_points)) continue;
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
    if (g_rank == 0) std::cout << summary.BriefReport() << std::endl << std::endl;
    if (!summary.message.empty()) std::cerr << summary.message << std::endl;
    return summary.IsSolutionUsable();
}

// tests/dataset_vo.cpp:87-138
int main(int argc, char **argv) {
    const bool use_frontend = argc >= 3 && std::string(argv[2]) == "--frontend";
    if (argc < 4 && !use_frontend) {
        std::cerr << "usage: dataset_vo_gpu <dataset.csv> <init_poses.csv> <init_map.csv> [--huber A] [--window N] [--gpus N]\n"
                     "       dataset_vo_gpu <dataset.csv> --frontend [--huber A] [--window N] [--gpus N]" << std::endl;
        return EXIT_FAILURE;
    }
    double huber = 0.0;
    ceres_slam::uint window_size = 0;
    int gpus = 0;
    for (int a = use_frontend ? 3 : 4; a + 1 < argc; ++a) {
        if (std::string(argv[a]) == "--huber") huber = std::atof(argv[a + 1]);
        if (std::string(argv[a]) == "--window") window_size = (ceres_slam::uint)std::atoi(argv[a + 1]);
        if (std::string(argv[a]) == "--gpus") gpus = std::atoi(argv[a + 1]);
    }
    // --gpus N: fork the peers before anything touches a GPU; pipes: id (rank 0 -> r), map points (r -> rank 0)
    std::vector<int> id_w, pts_r;
    std::vector<pid_t> peers;
    int my_id_r = -1, my_pts_w = -1;
    if (gpus > 0) {
        if (window_size != 0) { std::cerr << "--gpus: full batch only (one communicator per run)" << std::endl; return EXIT_FAILURE; }
        g_world = gpus;
        for (int r = 1; r < gpus; ++r) {
            int a[2], b[2];
            if (pipe(a) || pipe(b)) { perror("pipe"); return EXIT_FAILURE; }
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return EXIT_FAILURE; }
            if (pid == 0) {     // peer r
                g_rank = r;
                close(a[1]); close(b[0]);
                for (int fd : id_w) close(fd);
                for (int fd : pts_r) close(fd);
                id_w.clear(); pts_r.clear(); peers.clear();
                my_id_r = a[0]; my_pts_w = b[1];
                break;
            }
            close(a[0]); close(b[1]);
            id_w.push_back(a[1]); pts_r.push_back(b[0]); peers.push_back(pid);
        }
        if (gpus > 1) {     // rank r takes the r-th entry of a device list the caller set, or device r
            std::string pick = std::to_string(g_rank);
            if (const char *vis = getenv("HIP_VISIBLE_DEVICES")) {
                std::vector<std::string> ids;
                std::stringstream ss(vis);
                for (std::string t; std::getline(ss, t, ',');) if (!t.empty()) ids.push_back(t);
                if ((int)ids.size() < gpus) { std::cerr << "HIP_VISIBLE_DEVICES lists fewer than --gpus devices" << std::endl; return EXIT_FAILURE; }
                pick = ids[g_rank];
            }
            setenv("HIP_VISIBLE_DEVICES", pick.c_str(), 1);
        }
        if (g_rank == 0) {
            if (ssba_rccl_unique_id(g_rccl_id, sizeof g_rccl_id)) { std::cerr << "ssba_rccl_unique_id: " << ssba_last_error() << std::endl; return EXIT_FAILURE; }
            for (int fd : id_w)
                if (write(fd, g_rccl_id, sizeof g_rccl_id) != (ssize_t)sizeof g_rccl_id) { perror("write"); return EXIT_FAILURE; }
        } else if (read(my_id_r, g_rccl_id, sizeof g_rccl_id) != (ssize_t)sizeof g_rccl_id) {
            perror("read");
            return EXIT_FAILURE;
        }
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
    if (g_world > 1) {      // the map points live on their ranks: collect them on rank 0
        struct Rec { uint32_t j; double p[3]; };
        if (g_rank > 0) {
            std::vector<Rec> out;
            for (ceres_slam::uint j = 0; j < dataset.num_points; ++j)
                if (dataset.initialized_point[j] && owned(j, dataset.num_points))
                    out.push_back(Rec{(uint32_t)j, {dataset.map_points[j].data()[0], dataset.map_points[j].data()[1], dataset.map_points[j].data()[2]}});
            const uint64_t n = out.size();
            bool ok = write(my_pts_w, &n, sizeof n) == (ssize_t)sizeof n;
            const char *bytes = reinterpret_cast<const char *>(out.data());
            for (size_t off = 0; ok && off < n * sizeof(Rec);) {
                const ssize_t w = write(my_pts_w, bytes + off, n * sizeof(Rec) - off);
                if (w <= 0) ok = false; else off += (size_t)w;
            }
            return usable && ok ? EXIT_SUCCESS : EXIT_FAILURE;
        }
        for (size_t r = 0; r < pts_r.size(); ++r) {
            uint64_t n = 0;
            bool ok = read(pts_r[r], &n, sizeof n) == (ssize_t)sizeof n;
            std::vector<Rec> in(ok ? n : 0);
            char *bytes = reinterpret_cast<char *>(in.data());
            for (size_t off = 0; ok && off < n * sizeof(Rec);) {
                const ssize_t g = read(pts_r[r], bytes + off, n * sizeof(Rec) - off);
                if (g <= 0) ok = false; else off += (size_t)g;
            }
            for (const Rec &q : in)
                for (int c = 0; c < 3; ++c) dataset.map_points[q.j].data()[c] = q.p[c];
            int status = 0;
            waitpid(peers[r], &status, 0);
            if (!ok || !WIFEXITED(status) || WEXITSTATUS(status) != 0) usable = false;
        }
    }
    for (int a = 1; a < argc; ++a)          // --refprecision: the reference's four significant digits (utils/utils.hpp:34) instead of 17
        if (std::string(argv[a]) == "--refprecision") dataset.csv_precision = ceres_slam::DatasetProblem::kReferenceCsvPrecision;
    dataset.write_csv(filename);
    return usable ? EXIT_SUCCESS : EXIT_FAILURE;
}

// dataset_vo_sun_gpu -- the reference's sun-aided sliding-window VO driver (/root/reference tests/dataset_vo_sun.cpp)
// written against include/ceres_slam_amd/ceres_shim.hpp: the same Ceres calls, executed by the MI355X back end.
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

// Eigen::SelfAdjointEigenSolver<M>(A).operatorInverseSqrt() for n <= 6: cyclic Jacobi, V diag(1/sqrt(w)) V^T
static void inverse_sqrt_symmetric(int n, const double *A, double *out) {
    double a[36], v[36];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { a[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]); v[i * n + j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? diag : off) += a[i * n + j] * a[i * n + j];
        if (off <= 1e-32 * diag) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                if (a[p * n + q] == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * a[p * n + q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {      // A <- A J
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {      // A <- J^T A
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += v[i * n + k] * v[j * n + k] / std::sqrt(a[k * n + k]);
            out[i * n + j] = s;
        }
}

// SSBA_DRIVER_TIMING=1: wall time spent in the three stages of a window, printed at exit
static double g_time[3] = {0, 0, 0};
struct StageTimer {
    int stage;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(int s) : stage(s), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { g_time[stage] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Dataset {
    size_t num_states = 0, num_points = 0;
    double intr[5];
    std::vector<unsigned> state_ids, point_ids;
    std::vector<double> obs, obs_covars;                 // 3 / 9 per observation
    std::vector<double> poses, pose_covars, points;      // 12 / 36 per state, 3 per point
    std::vector<bool> initialized;
    std::vector<double> sun_dir_g, sun_obs, sun_covars;  // 3 / 3 / 4 per state
    std::vector<bool> has_sun;
    std::vector<std::vector<unsigned>> idx_of;
};

static void triangulate(const Dataset &D, unsigned i, double *p) {     // stereo_camera.hpp:112-120
    const double b_over_d = D.intr[4] / D.obs[3 * i + 2];
    p[0] = (D.obs[3 * i] - D.intr[2]) * b_over_d;
    p[1] = (D.obs[3 * i + 1] - D.intr[3]) * b_over_d * (D.intr[0] / D.intr[1]);
    p[2] = D.intr[0] * b_over_d;
}

// DatasetProblemSun::compute_initial_guess(k1, k2) (dataset_problem_sun.cpp:250-354)
static bool compute_initial_guess(Dataset &D, size_t k1, size_t k2) {
    StageTimer timer(0);
    if (k2 <= k1 + 1) return true;
    const uint32_t num_iters = 400;
    std::vector<uint32_t> offset(1, 0), samples;
    std::vector<double> pts0, pts1;
    std::vector<unsigned> match_km1;
    for (size_t k = k1 + 1; k < k2; ++k) {
        std::vector<unsigned> a, b;
        std::map<unsigned, unsigned> in_k;
        for (unsigned i : D.idx_of[k]) in_k[D.point_ids[i]] = i;
        std::map<unsigned, int> kept;
        for (unsigned i : D.idx_of[k - 1]) if (in_k.count(D.point_ids[i])) { a.push_back(i); kept[D.point_ids[i]] = 1; }
        for (unsigned i : D.idx_of[k]) if (kept.count(D.point_ids[i])) b.push_back(i);
        if (a.size() < 3 || a.size() != b.size()) { std::cout << "WARNING: Fewer than 3 inliers found." << std::endl; return false; }
        for (size_t m = 0; m < a.size(); ++m) {
            double p[3];
            triangulate(D, a[m], p); pts0.insert(pts0.end(), p, p + 3);
            triangulate(D, b[m], p); pts1.insert(pts1.end(), p, p + 3);
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
    std::vector<uint32_t> count(num_pairs);
    ssba_camera cam = {D.intr[0], D.intr[1], D.intr[2], D.intr[3], D.intr[4]};
    const int rc = ssba_frontend_ransac(&cam, -1, num_pairs, offset.data(), pts0.data(), pts1.data(), samples.data(), num_iters, 4.0,
                                        T.data(), inlier.data(), count.data(), nullptr);              // :315-318
    if (rc) { std::cerr << "ssba_frontend_ransac: " << ssba_status_string(rc) << " (" << ssba_last_error() << ")" << std::endl; return false; }
    for (size_t k = k1 + 1; k < k2; ++k) {
        const size_t q = k - k1 - 1;
        if (count[q] < 3) { std::cout << "WARNING: Fewer than 3 inliers found." << std::endl; return false; }   // :325-328
        const double *Tk = &T[12 * q], *Tp = &D.poses[12 * (k - 1)];
        double *Tn = &D.poses[12 * k];
        for (int i = 0; i < 3; ++i) {                                    // poses[k] = T_k_km1 * poses[k-1]  (:331)
            Tn[i] = Tk[3 + 3 * i] * Tp[0] + Tk[4 + 3 * i] * Tp[1] + Tk[5 + 3 * i] * Tp[2] + Tk[i];
            for (int j = 0; j < 3; ++j) Tn[3 + 3 * i + j] = Tk[3 + 3 * i] * Tp[3 + j] + Tk[4 + 3 * i] * Tp[6 + j] + Tk[5 + 3 * i] * Tp[9 + j];
        }
        for (uint32_t m = offset[q]; m < offset[q + 1]; ++m) {           // :335-350
            const unsigned j = D.point_ids[match_km1[m]];
            if (!inlier[m] || j >= D.num_points || D.initialized[j]) continue;
            const double d[3] = {pts0[3 * m] - Tp[0], pts0[3 * m + 1] - Tp[1], pts0[3 * m + 2] - Tp[2]};
            for (int c = 0; c < 3; ++c) D.points[3 * j + c] = Tp[3 + c] * d[0] + Tp[6 + c] * d[1] + Tp[9 + c] * d[2];
            D.initialized[j] = true;
        }
    }
    return true;
}

// solveWindow (tests/dataset_vo_sun.cpp:28-185)
static void solve_window(Dataset &D, size_t k1, size_t k2, bool use_sun, double huber_param, double az_err_thresh, double zen_err_thresh) {
    std::cerr << "Working on interval [" << k1 << "," << k2 << ")/" << D.num_states << ": ";
    ceres::Problem problem;
    auto camera = std::make_shared<const ceres_slam::StereoCamera>(D.intr[0], D.intr[1], D.intr[2], D.intr[3], D.intr[4]);
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    for (size_t k = k1; k < k2; ++k) {
        for (unsigned i : D.idx_of[k]) {
            const unsigned j = D.point_ids[i];
            if (!D.initialized[j]) continue;                                    // :55
            double stereo_obs_stiffness[9];
            inverse_sqrt_symmetric(3, &D.obs_covars[9 * (size_t)j], stereo_obs_stiffness);      // stereo_obs_covars[j]  (:58-60)
            ceres::CostFunction *stereo_cost = ceres_slam::StereoReprojectionErrorAutomatic::Create(camera, &D.obs[3 * i], stereo_obs_stiffness);
            problem.AddResidualBlock(stereo_cost, NULL, &D.poses[12 * k], &D.points[3 * j]);
        }
        if (use_sun && D.has_sun[k]) {                                          // :77-106
            double sun_obs_stiffness[4];
            inverse_sqrt_symmetric(2, &D.sun_covars[4 * k], sun_obs_stiffness);
            ceres::CostFunction *sun_cost = ceres_slam::SunSensorErrorAutomatic::Create(&D.sun_obs[3 * k], &D.sun_dir_g[3 * k], sun_obs_stiffness,
                                                                                         az_err_thresh, zen_err_thresh);
            problem.AddResidualBlock(sun_cost, huber_param > 0. ? new ceres::HuberLoss(huber_param) : NULL, &D.poses[12 * k]);
        }
    }
    double pose_prior_stiffness[36];                                            // :111-124
    inverse_sqrt_symmetric(6, &D.pose_covars[36 * k1], pose_prior_stiffness);
    ceres::CostFunction *pose_prior_cost = ceres_slam::PoseErrorAutomatic::Create(&D.poses[12 * k1], pose_prior_stiffness);
    problem.AddResidualBlock(pose_prior_cost, NULL, &D.poses[12 * k1]);
    for (size_t k = k1; k < k2; ++k) problem.SetParameterization(&D.poses[12 * k], se3_perturbation);

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
    covar_blocks.push_back(std::make_pair(&D.poses[12 * (k1 + 1)], &D.poses[12 * (k1 + 1)]));
    StageTimer timer(2);
    if (!covariance.Compute(covar_blocks, &problem)) {
        std::cout << "WARNING: Covariance computation failed! Using previous state covariance." << std::endl;
        std::copy(&D.pose_covars[36 * k1], &D.pose_covars[36 * k1] + 36, &D.pose_covars[36 * (k1 + 1)]);
    } else {
        covariance.GetCovarianceBlockInTangentSpace(&D.poses[12 * (k1 + 1)], &D.poses[12 * (k1 + 1)], &D.pose_covars[36 * (k1 + 1)]);
    }
}

static bool write_poses(const Dataset &D, const std::string &base) {          // DatasetProblemSun::write_csv (:172-232)
    std::cout << "Outputting to file:\n\t" << base + "_poses.csv" << std::endl;
    std::ofstream po(base + "_poses.csv");
    if (!po.is_open()) return false;
    po.precision(17);
    po << "T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33" << std::endl;
    for (size_t k = 0; k < D.num_states; ++k) {
        const double *T = &D.poses[12 * k];
        for (int i = 0; i < 3; ++i) po << T[3 + 3 * i] << "," << T[4 + 3 * i] << "," << T[5 + 3 * i] << "," << T[i] << ",";
        po << "0,0,0,1" << std::endl;
    }
    return true;
}

int main(int argc, char **argv) {
    const std::string usage("usage: dataset_vo_sun_gpu <track_file> <ref_sun_file> <obs_sun_file> [--window (2)] [--huber-param (0)] "
                            "[--az-err-thresh (1000)] [--zen-err-thresh (1000)] [--sun-only]");
    if (argc < 4) { std::cerr << usage << std::endl; return EXIT_FAILURE; }
    size_t window_size = 2;
    bool sun_only = false;
    double huber_param = 0., az_err_thresh = 1000., zen_err_thresh = 1000.;
    const double pi = 3.14159265358979323846;
    for (int a = 4; a < argc; ++a) {
        const std::string flag(argv[a]);
        if (flag == "--window" && argc > a + 1) window_size = (size_t)std::stoi(argv[++a]);
        else if (flag == "--huber-param" && argc > a + 1) huber_param = std::stod(argv[++a]);
        else if (flag == "--az-err-thresh" && argc > a + 1) az_err_thresh = std::stod(argv[++a]) * pi / 180.;     // degrees on the command line
        else if (flag == "--zen-err-thresh" && argc > a + 1) zen_err_thresh = std::stod(argv[++a]) * pi / 180.;
        else if (flag == "--sun-only") sun_only = true;
        else { std::cerr << usage << std::endl; return EXIT_FAILURE; }
    }
    // ---- DatasetProblemSun::read_csv --------------------------------------------------------------
    Dataset D;
    std::ifstream f(argv[1]);
    if (!f.is_open()) { std::cerr << "Error: couldn't open " << argv[1] << std::endl; return EXIT_FAILURE; }
    std::string line;
    std::getline(f, line); auto meta = parse_row(line);
    std::getline(f, line); auto intr = parse_row(line);
    std::getline(f, line); auto first_pose = parse_row(line);
    if (meta.size() < 2 || intr.size() < 5 || first_pose.size() < 16) { std::cerr << "malformed header" << std::endl; return EXIT_FAILURE; }
    D.num_states = (size_t)meta[0]; D.num_points = (size_t)meta[1];
    std::copy(intr.begin(), intr.begin() + 5, D.intr);
    while (std::getline(f, line)) {
        auto r = parse_row(line);
        if (r.size() < 14) continue;
        D.state_ids.push_back((unsigned)r[0]); D.point_ids.push_back((unsigned)r[1]);
        D.obs.insert(D.obs.end(), r.begin() + 2, r.begin() + 5);
        D.obs_covars.insert(D.obs_covars.end(), r.begin() + 5, r.begin() + 14);
    }
    D.poses.assign(D.num_states * 12, 0.0);
    for (size_t k = 0; k < D.num_states; ++k) D.poses[12 * k + 3] = D.poses[12 * k + 7] = D.poses[12 * k + 11] = 1.0;   // SE3(): identity
    D.poses[0] = first_pose[3]; D.poses[1] = first_pose[7]; D.poses[2] = first_pose[11];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D.poses[3 + 3 * i + j] = first_pose[4 * i + j];
    D.pose_covars.assign(D.num_states * 36, 0.0);
    for (int c = 0; c < 6; ++c) D.pose_covars[7 * c] = 1e-12;                      // :81
    D.points.assign(D.num_points * 3, 0.0);
    D.initialized.assign(D.num_points, false);
    D.idx_of.resize(D.num_states);
    for (size_t i = 0; i < D.state_ids.size(); ++i) {
        if (D.state_ids[i] >= D.num_states || D.point_ids[i] >= D.num_points) { std::cerr << "observation out of range" << std::endl; return EXIT_FAILURE; }
        D.idx_of[D.state_ids[i]].push_back((unsigned)i);
    }
    for (unsigned j : D.point_ids)      // solveWindow indexes the per-observation covariance list by POINT id (:58)
        if (9 * (size_t)j + 9 > D.obs_covars.size()) { std::cerr << "point id beyond the covariance list" << std::endl; return EXIT_FAILURE; }
    D.sun_dir_g.assign(D.num_states * 3, 0.0); D.sun_obs.assign(D.num_states * 3, 0.0); D.sun_covars.assign(D.num_states * 4, 0.0);
    D.has_sun.assign(D.num_states, false);
    std::ifstream f2(argv[2]);
    if (!f2.is_open()) { std::cerr << "Error: couldn't open " << argv[2] << std::endl; return EXIT_FAILURE; }
    while (std::getline(f2, line)) {
        auto r = parse_row(line);
        if (r.size() < 4 || (size_t)r[0] >= D.num_states) continue;
        std::copy(r.begin() + 1, r.begin() + 4, &D.sun_dir_g[3 * (size_t)r[0]]);
    }
    std::ifstream f3(argv[3]);
    if (!f3.is_open()) { std::cerr << "Error: couldn't open " << argv[3] << std::endl; return EXIT_FAILURE; }
    while (std::getline(f3, line)) {
        auto r = parse_row(line);
        if (r.size() < 8 || (size_t)r[0] >= D.num_states) continue;
        const size_t k = (size_t)r[0];
        std::copy(r.begin() + 1, r.begin() + 4, &D.sun_obs[3 * k]);
        std::copy(r.begin() + 4, r.begin() + 8, &D.sun_covars[4 * k]);
        D.has_sun[k] = true;
    }
    if (window_size == 0 || window_size > D.num_states) window_size = D.num_states;      // 0 = full batch (:259-262)

    auto run_pass = [&](bool use_sun) {                                             // :267-283 / :295-311
        for (size_t k1 = 0; k1 + window_size <= D.num_states; ++k1) {
            const size_t k2 = std::min(k1 + window_size, D.num_states);
            if (compute_initial_guess(D, k1, k2)) {
                solve_window(D, k1, k2, use_sun, huber_param, az_err_thresh, zen_err_thresh);
            } else {
                std::cerr << "WARNING: Initial guess failed. Copying previous pose and covariance." << std::endl;
                std::copy(&D.poses[12 * k1], &D.poses[12 * k1] + 12, &D.poses[12 * (k2 - 1)]);
                std::copy(&D.pose_covars[36 * k1], &D.pose_covars[36 * k1] + 36, &D.pose_covars[36 * (k2 - 1)]);
            }
            D.initialized.assign(D.num_points, false);                              // reset_points()
        }
    };
    std::string track(argv[1]), base = track.substr(0, track.find('.'));
    if (!sun_only) {
        std::cerr << "Computing VO without sun measurements" << std::endl;
        run_pass(false);
        if (!write_poses(D, base)) return EXIT_FAILURE;
    }
    std::cerr << "Computing VO with sun measurements" << std::endl;
    run_pass(true);
    std::string obs_sun(argv[3]);                                                   // :314-321
    obs_sun = obs_sun.substr(0, obs_sun.find('.'));
    const size_t us = obs_sun.find_last_of('_');
    const std::string tag = us == std::string::npos ? obs_sun : obs_sun.substr(us + 1);
    if (std::getenv("SSBA_DRIVER_TIMING"))
        std::cerr << "stage seconds: initial guess " << g_time[0] << ", Solve " << g_time[1] << ", Covariance " << g_time[2] << std::endl;
    return write_poses(D, base + "_" + tag) ? EXIT_SUCCESS : EXIT_FAILURE;
}

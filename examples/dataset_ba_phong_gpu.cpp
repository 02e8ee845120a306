// dataset_ba_phong_gpu -- the solve stage of the reference's Phong driver
// (/root/reference tests/dataset_ba_phong.cpp:26-255, solveWindow, single-stage) written against
// include/ceres_slam_amd/ceres_shim.hpp: the same calls the reference makes against Ceres, executed by the
// MI355X back end.
//
// usage: dataset_ba_phong_gpu <dataset.csv> <init_poses.csv> <init_map.csv> <init_lights.csv> [--nolight | --dirlight] [--multistage] [--window N]
//        dataset_ba_phong_gpu <dataset.csv> --frontend [--nolight | --dirlight] [--multistage] [--window N]
//   dataset.csv     reference format (src/ceres_slam/dataset_problem_phong.cpp:16-117): rows
//                   "num_states,num_vertices,num_materials" | "fu,fv,cu,cv,b" |
//                   "stereo var (3), normal var (3), intensity var" | light position or direction |
//                   first pose (4x4 row-major), then "t,j,material,u,v,d,I,nx,ny,nz" rows
//   init_*.csv      initial guess in the formats the reference's write_csv emits (:177-232): 4x4 poses;
//                   "point_id,x,y,z,nx,ny,nz,ka,ks,exponent,kd"; light "x,y,z"
// --frontend computes the initial guess itself as the reference's main does (tests/dataset_ba_phong.cpp:303-311;
// DatasetProblemPhong::compute_initial_guess, src/ceres_slam/dataset_problem_phong.cpp:250-391): materials at
// (ka, ks, exponent) = (0, 0, 1), textures at the median intensity of the material, reciprocal matches of consecutive
// states, the 400-hypothesis RANSAC of all pairs in one GPU batch (threshold 9), pose chaining, and for inlier vertices
// position and normal through poses[k-1]^-1 -- including the reference's `material_ids[i]` indexing of the vertex
// material (i = position in the pair's list, :369-370); light from the dataset header.  --window N slides
// solveWindow over the states (:317-327; for k1 > 0 the reference's compute_initial_guess(k2-1, k2) touches no vertex).  --multistage runs the reference's three stages
// (:96-100 poses and points without lighting terms, :210-246 lighting with every pose and position block
// constant, :249-252 everything jointly).
// Output: <dataset>_poses.csv / _map.csv / _lights.csv at full precision + the brief report.
#include <algorithm>
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
    if (argc < 5 && !use_frontend) {
        std::cerr << "usage: dataset_ba_phong_gpu <dataset.csv> <init_poses.csv> <init_map.csv> <init_lights.csv> [--nolight | --dirlight] [--multistage] [--window N]\n"
                     "       dataset_ba_phong_gpu <dataset.csv> --frontend [--nolight | --dirlight] [--multistage] [--window N]" << std::endl;
        return EXIT_FAILURE;
    }
    bool use_light = true, directional_light = false, multi_stage = false;
    size_t window_size = 0;
    for (int a = use_frontend ? 3 : 5; a < argc; ++a) {
        if (std::string(argv[a]) == "--nolight") use_light = false;
        if (std::string(argv[a]) == "--dirlight") directional_light = true;
        if (std::string(argv[a]) == "--multistage") multi_stage = true;
        if (std::string(argv[a]) == "--window" && a + 1 < argc) window_size = (size_t)std::atoi(argv[++a]);
    }
    std::ifstream f(argv[1]);
    if (!f.is_open()) { std::cerr << "Error: couldn't open " << argv[1] << std::endl; return EXIT_FAILURE; }
    std::string line;
    std::getline(f, line); auto meta = parse_row(line);
    std::getline(f, line); auto intr = parse_row(line);
    std::getline(f, line); auto var = parse_row(line);
    std::getline(f, line); auto light_row = parse_row(line);    // light position / direction (the initial-guess file overrides it)
    std::getline(f, line); auto first_pose = parse_row(line);   // first ground-truth pose (4x4 row-major)
    if (meta.size() < 3 || intr.size() < 5 || var.size() < 7) { std::cerr << "malformed header" << std::endl; return EXIT_FAILURE; }
    const size_t num_states = (size_t)meta[0], num_vertices = (size_t)meta[1], num_materials = (size_t)meta[2];
    std::vector<unsigned> vertex_ids, material_ids, state_of;
    std::vector<double> t, stereo_obs, int_list, normal_obs_list;
    while (std::getline(f, line)) {
        auto r = parse_row(line);
        if (r.size() < 10) continue;
        if (!t.empty() && r[0] != t.back()) state_of.push_back(state_of.back() + 1);   // a new timestamp = next state (:120-133)
        else state_of.push_back(t.empty() ? 0 : state_of.back());
        t.push_back(r[0]);
        vertex_ids.push_back((unsigned)r[1]); material_ids.push_back((unsigned)r[2]);
        stereo_obs.insert(stereo_obs.end(), r.begin() + 3, r.begin() + 6);
        int_list.push_back(r[6]);
        normal_obs_list.insert(normal_obs_list.end(), r.begin() + 7, r.begin() + 10);
    }
    // parameter blocks: poses 12 doubles [t | R row-major]; per vertex position / normal; per material
    // Phong parameters (ka, ks, exponent) and texture (kd); one light
    std::vector<double> poses(num_states * 12, 0.0), positions(num_vertices * 3, 0.0), normals(num_vertices * 3, 0.0);
    std::vector<double> phong(num_materials * 3, 0.0), texture(num_materials, 0.0), light(3, 0.0);
    std::vector<bool> initialized(num_vertices, false);
    std::vector<unsigned> material_of_vertex(num_vertices, 0);      // map_vertices[j].material(): what solveWindow and write_csv use
    for (size_t i = 0; i < vertex_ids.size(); ++i) material_of_vertex[vertex_ids[i]] = material_ids[i];
    if (use_frontend) {
        // ---- DatasetProblemPhong::compute_initial_guess(0, num_states) ---------------------------------------
        if (first_pose.size() < 16 || light_row.size() < 3) { std::cerr << "malformed header" << std::endl; return EXIT_FAILURE; }
        for (int c = 0; c < 3; ++c) light[c] = light_row[c];
        for (int i = 0; i < 3; ++i) { poses[i] = first_pose[4 * i + 3]; for (int j = 0; j < 3; ++j) poses[3 + 3 * i + j] = first_pose[4 * i + j]; }
        for (size_t m = 0; m < num_materials; ++m) {                         // :264-277
            phong[3 * m] = 0.0; phong[3 * m + 1] = 0.0; phong[3 * m + 2] = 1.0;
            std::vector<double> ints;
            for (size_t i = 0; i < material_ids.size(); ++i) if (material_ids[i] == m) ints.push_back(int_list[i]);
            if (ints.empty()) continue;
            std::nth_element(ints.begin(), ints.begin() + ints.size() / 2, ints.end());
            texture[m] = ints[ints.size() / 2];
        }
        std::vector<std::vector<unsigned>> idx_of(num_states);
        for (size_t i = 0; i < state_of.size(); ++i) idx_of[state_of[i]].push_back((unsigned)i);
        auto triangulate = [&](unsigned i, double *p) {                      // stereo_camera.hpp:112-120
            const double b_over_d = intr[4] / stereo_obs[3 * i + 2];
            p[0] = (stereo_obs[3 * i] - intr[2]) * b_over_d;
            p[1] = (stereo_obs[3 * i + 1] - intr[3]) * b_over_d * (intr[0] / intr[1]);
            p[2] = intr[0] * b_over_d;
        };
        std::vector<uint32_t> offset(1, 0), samples;
        std::vector<double> pts0, pts1;
        std::vector<unsigned> match_km1;
        const uint32_t num_iters = 400;
        for (size_t k = 1; k < num_states; ++k) {                            // :279-331
            std::vector<unsigned> a, b;
            std::map<unsigned, unsigned> in_k;
            for (unsigned i : idx_of[k]) in_k[vertex_ids[i]] = i;
            std::map<unsigned, int> kept;
            for (unsigned i : idx_of[k - 1]) if (in_k.count(vertex_ids[i])) { a.push_back(i); kept[vertex_ids[i]] = 1; }
            for (unsigned i : idx_of[k]) if (kept.count(vertex_ids[i])) b.push_back(i);
            if (a.size() < 3 || a.size() != b.size()) { std::cerr << "state " << k << ": fewer than 3 matches" << std::endl; return EXIT_FAILURE; }
            for (size_t m = 0; m < a.size(); ++m) {
                double p[3];
                triangulate(a[m], p); pts0.insert(pts0.end(), p, p + 3);
                triangulate(b[m], p); pts1.insert(pts1.end(), p, p + 3);
                match_km1.push_back(a[m]);
            }
            offset.push_back((uint32_t)(pts0.size() / 3));
            std::vector<uint32_t> smp(3 * num_iters);
            if (ssba_ransac_samples((uint32_t)a.size(), num_iters, __GNUC__ >= 11 ? 1 : 0, smp.data())) return EXIT_FAILURE;
            samples.insert(samples.end(), smp.begin(), smp.end());
        }
        const uint32_t num_pairs = (uint32_t)num_states - 1;
        std::vector<double> T((size_t)num_pairs * 12);
        std::vector<uint8_t> inlier(pts0.size() / 3);
        ssba_camera cam = {intr[0], intr[1], intr[2], intr[3], intr[4]};
        if (num_pairs) {
            const int rc = ssba_frontend_ransac(&cam, -1, num_pairs, offset.data(), pts0.data(), pts1.data(), samples.data(), num_iters, 9.0,
                                                T.data(), inlier.data(), nullptr, nullptr);           // :340-343
            if (rc) { std::cerr << "ssba_frontend_ransac: " << ssba_status_string(rc) << std::endl; return EXIT_FAILURE; }
        }
        for (size_t k = 1; k < num_states; ++k) {
            const double *Tk = &T[12 * (k - 1)], *Tp = &poses[12 * (k - 1)];
            double *Tn = &poses[12 * k];
            for (int i = 0; i < 3; ++i) {                                    // poses[k] = T_k_km1 * poses[k-1]  (:352)
                Tn[i] = Tk[3 + 3 * i] * Tp[0] + Tk[4 + 3 * i] * Tp[1] + Tk[5 + 3 * i] * Tp[2] + Tk[i];
                for (int j = 0; j < 3; ++j) Tn[3 + 3 * i + j] = Tk[3 + 3 * i] * Tp[3 + j] + Tk[4 + 3 * i] * Tp[6 + j] + Tk[5 + 3 * i] * Tp[9 + j];
            }
            for (uint32_t m = offset[k - 1]; m < offset[k]; ++m) {           // :356-380
                const unsigned j = vertex_ids[match_km1[m]];
                if (!inlier[m] || j >= num_vertices || initialized[j]) continue;
                const double d[3] = {pts0[3 * m] - Tp[0], pts0[3 * m + 1] - Tp[1], pts0[3 * m + 2] - Tp[2]};
                const double *n = &normal_obs_list[3 * match_km1[m]];
                for (int c = 0; c < 3; ++c) {
                    positions[3 * j + c] = Tp[3 + c] * d[0] + Tp[6 + c] * d[1] + Tp[9 + c] * d[2];      // poses[k-1]^-1 * p
                    normals[3 * j + c] = Tp[3 + c] * n[0] + Tp[6 + c] * n[1] + Tp[9 + c] * n[2];        // rotation only
                }
                material_of_vertex[j] = material_ids[m - offset[k - 1]];    // the reference indexes material_ids by the position in the pair's list
                initialized[j] = true;
            }
        }
    } else {
        std::ifstream pf(argv[2]);
        if (!pf.is_open()) { std::cerr << "Error: couldn't open " << argv[2] << std::endl; return EXIT_FAILURE; }
        std::getline(pf, line);   // header
        for (size_t k = 0; k < num_states && std::getline(pf, line); ++k) {
            auto r = parse_row(line);
            if (r.size() < 16) { std::cerr << "malformed pose row" << std::endl; return EXIT_FAILURE; }
            for (int i = 0; i < 3; ++i) {
                poses[12 * k + i] = r[4 * i + 3];
                for (int j = 0; j < 3; ++j) poses[12 * k + 3 + 3 * i + j] = r[4 * i + j];
            }
        }
        std::ifstream mf(argv[3]);
        if (!mf.is_open()) { std::cerr << "Error: couldn't open " << argv[3] << std::endl; return EXIT_FAILURE; }
        std::getline(mf, line);
        while (std::getline(mf, line)) {
            auto r = parse_row(line);
            if (r.size() < 11) continue;
            const size_t j = (size_t)r[0];
            if (j >= num_vertices) continue;
            for (int c = 0; c < 3; ++c) { positions[3 * j + c] = r[1 + c]; normals[3 * j + c] = r[4 + c]; }
            const unsigned m = material_of_vertex[j];
            for (int c = 0; c < 3; ++c) phong[3 * m + c] = r[7 + c];
            texture[m] = r[10];
            initialized[j] = true;
        }
        std::ifstream lf(argv[4]);
        if (!lf.is_open()) { std::cerr << "Error: couldn't open " << argv[4] << std::endl; return EXIT_FAILURE; }
        std::getline(lf, line);
        std::getline(lf, line);
        auto r = parse_row(line);
        if (r.size() < 3) { std::cerr << "malformed light row" << std::endl; return EXIT_FAILURE; }
        for (int c = 0; c < 3; ++c) light[c] = r[c];
    }

    if (window_size == 0 || window_size > num_states) window_size = num_states;       // :313-315
    ceres::Solver::Summary summary;
    for (size_t k1 = 0; k1 + window_size <= num_states; ++k1) {                        // :317-327
    const size_t k2 = k1 + window_size;
    // ---- solveWindow (tests/dataset_ba_phong.cpp:26-255) ----
    std::cerr << "Working on interval [" << k1 << "," << k2 << ")" << std::endl;
    ceres::Problem problem;
    double stereo_stiffness[9] = {0}, normal_stiffness[9] = {0};
    for (int c = 0; c < 3; ++c) {
        stereo_stiffness[4 * c] = 1.0 / std::sqrt(var[c]);          // :34-37
        normal_stiffness[4 * c] = 1.0 / std::sqrt(var[3 + c]);      // :39-42
    }
    const double int_stiffness = 1.0 / std::sqrt(var[6]);           // :44
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    ceres::LocalParameterization *unit_vector_perturbation = ceres_slam::UnitVectorPerturbation::Create();
    auto camera = std::make_shared<const ceres_slam::StereoCamera>(intr[0], intr[1], intr[2], intr[3], intr[4]);

    for (size_t i = 0; i < vertex_ids.size(); ++i) {                // stereo terms (:53-73)
        const unsigned k = state_of[i], j = vertex_ids[i];
        if (k < k1 || k >= k2 || !initialized[j]) continue;
        ceres::CostFunction *stereo_cost = ceres_slam::StereoReprojectionErrorAutomatic::Create(camera, &stereo_obs[3 * i], stereo_stiffness);
        problem.AddResidualBlock(stereo_cost, NULL, &poses[12 * k], &positions[3 * j]);
        problem.SetParameterization(&poses[12 * k], se3_perturbation);
    }
    problem.SetParameterBlockConstant(&poses[12 * k1]);             // :76

    ceres::Solver::Options solver_options;                          // :79-87
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.num_threads = 8;
    solver_options.num_linear_solver_threads = 8;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    solver_options.trust_region_strategy_type = ceres::DOGLEG;
    solver_options.dogleg_type = ceres::SUBSPACE_DOGLEG;
    solver_options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;

    if (multi_stage) {                                              // stage 1 (:96-100): poses and points only, no lighting
        std::cerr << "Solving stage 1: poses and points" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
    }

    if (use_light) {                                                // lighting terms (:102-207)
        for (size_t i = 0; i < vertex_ids.size(); ++i) {
            const unsigned k = state_of[i], j = vertex_ids[i];
            if (k < k1 || k >= k2 || !initialized[j]) continue;
            const unsigned m = material_of_vertex[j];                  // map_vertices[j].material() (:117-121)
            ceres::CostFunction *intensity_cost =
                directional_light ? ceres_slam::IntensityErrorDirectionalLightAutomatic::Create(int_list[i], int_stiffness)
                                  : ceres_slam::IntensityErrorPointLightAutomatic::Create(int_list[i], int_stiffness);
            problem.AddResidualBlock(intensity_cost, NULL, &poses[12 * k], &positions[3 * j], &normals[3 * j], &phong[3 * m],
                                     &texture[m], light.data());
            problem.SetParameterLowerBound(&phong[3 * m], 0, 0.);   // :143-165
            problem.SetParameterUpperBound(&phong[3 * m], 0, 1.);
            problem.SetParameterLowerBound(&phong[3 * m], 1, 0.);
            problem.SetParameterUpperBound(&phong[3 * m], 1, 1.);
            problem.SetParameterLowerBound(&phong[3 * m], 2, 1.);
            problem.SetParameterLowerBound(&texture[m], 0, 0.);     // :175-178
            problem.SetParameterUpperBound(&texture[m], 0, 1.);
            ceres::CostFunction *normal_cost = ceres_slam::NormalErrorAutomatic::Create(&normal_obs_list[3 * i], normal_stiffness);
            problem.AddResidualBlock(normal_cost, NULL, &poses[12 * k], &normals[3 * j]);
            problem.SetParameterization(&normals[3 * j], unit_vector_perturbation);
        }
        if (directional_light) problem.SetParameterization(light.data(), unit_vector_perturbation);   // :201-204
    }

    if (multi_stage) {                                              // stage 2 (:210-246): lighting only
        for (size_t i = 0; i < vertex_ids.size(); ++i) {
            const unsigned k = state_of[i], j = vertex_ids[i];
            if (k < k1 || k >= k2) continue;
            if (initialized[j]) problem.SetParameterBlockConstant(&positions[3 * j]);
            problem.SetParameterBlockConstant(&poses[12 * k]);
        }
        std::cerr << "Solving stage 2: lighting" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
        if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;
        for (size_t i = 0; i < vertex_ids.size(); ++i) {
            const unsigned k = state_of[i], j = vertex_ids[i];
            if (k < k1 || k >= k2) continue;
            if (initialized[j]) problem.SetParameterBlockVariable(&positions[3 * j]);
            if (k > k1) problem.SetParameterBlockVariable(&poses[12 * k]);
        }
    }

    std::cerr << "Solving SLAM and lighting jointly" << std::endl; // :249-252
    ceres::Solve(solver_options, &problem, &summary);
    std::cout << summary.BriefReport() << std::endl << std::endl;
    if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;
    if (summary.termination_type == ceres::FAILURE) break;
    }   // windows

    // ---- write_csv (src/ceres_slam/dataset_problem_phong.cpp:177-232), full precision ----
    std::string base = argv[1];
    base = base.substr(0, base.rfind('.'));
    std::ofstream pose_file(base + "_poses.csv"), map_file(base + "_map.csv"), light_file(base + "_lights.csv");
    pose_file.precision(17); map_file.precision(17); light_file.precision(17);
    pose_file << "T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33" << std::endl;
    for (size_t k = 0; k < num_states; ++k) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) pose_file << poses[12 * k + 3 + 3 * i + j] << ",";
            pose_file << poses[12 * k + i] << ",";
        }
        pose_file << "0,0,0,1" << std::endl;
    }
    map_file << "point_id, x, y, z, nx, ny, nz, ka, ks, exponent, kd" << std::endl;
    for (size_t j = 0; j < num_vertices; ++j)
        if (initialized[j]) {
            const unsigned m = material_of_vertex[j];
            map_file << j << "," << positions[3 * j] << "," << positions[3 * j + 1] << "," << positions[3 * j + 2] << "," << normals[3 * j]
                     << "," << normals[3 * j + 1] << "," << normals[3 * j + 2] << "," << phong[3 * m] << "," << phong[3 * m + 1] << ","
                     << phong[3 * m + 2] << "," << texture[m] << std::endl;
        }
    light_file << (directional_light ? "i, j, k" : "x, y, z") << std::endl;
    light_file << light[0] << "," << light[1] << "," << light[2] << std::endl;
    return summary.termination_type == ceres::FAILURE ? EXIT_FAILURE : EXIT_SUCCESS;
}

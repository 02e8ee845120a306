// dataset_ba_phong_gpu -- the solve stage of the reference's Phong driver
// (/root/reference tests/dataset_ba_phong.cpp:26-255, solveWindow, single-stage) written against
// include/ceres_slam_amd/ceres_shim.hpp: the same calls the reference makes against Ceres, executed by the
// MI355X back end.
//
// usage: dataset_ba_phong_gpu <dataset.csv> <init_poses.csv> <init_map.csv> <init_lights.csv> [--nolight | --dirlight] [--multistage]
//   dataset.csv     reference format (src/ceres_slam/dataset_problem_phong.cpp:16-117): rows
//                   "num_states,num_vertices,num_materials" | "fu,fv,cu,cv,b" |
//                   "stereo var (3), normal var (3), intensity var" | light position or direction |
//                   first pose (4x4 row-major), then "t,j,material,u,v,d,I,nx,ny,nz" rows
//   init_*.csv      initial guess in the formats the reference's write_csv emits (:177-232): 4x4 poses;
//                   "point_id,x,y,z,nx,ny,nz,ka,ks,exponent,kd"; light "x,y,z"
// The front end that produces the initial guess (compute_initial_guess: matching + RANSAC, :250-400) is
// SURVEY.md section 8(f) row N2 and not part of this path.  --multistage runs the reference's three stages
// (:96-100 poses and points without lighting terms, :210-246 lighting with every pose and position block
// constant, :249-252 everything jointly).
// Output: <dataset>_poses.csv / _map.csv / _lights.csv at full precision + the brief report.
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
    if (argc < 5) {
        std::cerr << "usage: dataset_ba_phong_gpu <dataset.csv> <init_poses.csv> <init_map.csv> <init_lights.csv> [--nolight | --dirlight] [--multistage]" << std::endl;
        return EXIT_FAILURE;
    }
    bool use_light = true, directional_light = false, multi_stage = false;
    for (int a = 5; a < argc; ++a) {
        if (std::string(argv[a]) == "--nolight") use_light = false;
        if (std::string(argv[a]) == "--dirlight") directional_light = true;
        if (std::string(argv[a]) == "--multistage") multi_stage = true;
    }
    std::ifstream f(argv[1]);
    if (!f.is_open()) { std::cerr << "Error: couldn't open " << argv[1] << std::endl; return EXIT_FAILURE; }
    std::string line;
    std::getline(f, line); auto meta = parse_row(line);
    std::getline(f, line); auto intr = parse_row(line);
    std::getline(f, line); auto var = parse_row(line);
    std::getline(f, line);   // light: taken from the initial-guess file
    std::getline(f, line);   // first ground-truth pose: the initial guess carries it
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
    {
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
        std::vector<unsigned> material_of_vertex(num_vertices, 0);
        for (size_t i = 0; i < vertex_ids.size(); ++i) material_of_vertex[vertex_ids[i]] = material_ids[i];
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

    // ---- solveWindow (tests/dataset_ba_phong.cpp:26-255), k1 = 0, k2 = num_states ----
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
        if (!initialized[j]) continue;
        ceres::CostFunction *stereo_cost = ceres_slam::StereoReprojectionErrorAutomatic::Create(camera, &stereo_obs[3 * i], stereo_stiffness);
        problem.AddResidualBlock(stereo_cost, NULL, &poses[12 * k], &positions[3 * j]);
        problem.SetParameterization(&poses[12 * k], se3_perturbation);
    }
    problem.SetParameterBlockConstant(&poses[0]);                   // :76

    ceres::Solver::Options solver_options;                          // :79-87
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.num_threads = 8;
    solver_options.num_linear_solver_threads = 8;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    solver_options.trust_region_strategy_type = ceres::DOGLEG;
    solver_options.dogleg_type = ceres::SUBSPACE_DOGLEG;
    solver_options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
    ceres::Solver::Summary summary;

    if (multi_stage) {                                              // stage 1 (:96-100): poses and points only, no lighting
        std::cerr << "Solving stage 1: poses and points" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
    }

    if (use_light) {                                                // lighting terms (:102-207)
        for (size_t i = 0; i < vertex_ids.size(); ++i) {
            const unsigned k = state_of[i], j = vertex_ids[i], m = material_ids[i];
            if (!initialized[j]) continue;
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
            if (initialized[j]) problem.SetParameterBlockConstant(&positions[3 * j]);
            problem.SetParameterBlockConstant(&poses[12 * k]);
        }
        std::cerr << "Solving stage 2: lighting" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
        if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;
        for (size_t i = 0; i < vertex_ids.size(); ++i) {
            const unsigned k = state_of[i], j = vertex_ids[i];
            if (initialized[j]) problem.SetParameterBlockVariable(&positions[3 * j]);
            if (k > 0) problem.SetParameterBlockVariable(&poses[12 * k]);
        }
    }

    std::cerr << "Solving SLAM and lighting jointly" << std::endl; // :249-252
    ceres::Solve(solver_options, &problem, &summary);
    std::cout << summary.BriefReport() << std::endl << std::endl;
    if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;

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
    std::vector<unsigned> material_of_vertex(num_vertices, 0);
    for (size_t i = 0; i < vertex_ids.size(); ++i) material_of_vertex[vertex_ids[i]] = material_ids[i];
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

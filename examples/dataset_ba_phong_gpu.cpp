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
#include <cmath>
#include <iostream>

#include "ceres_slam_amd/ceres_shim.hpp"
#include "ceres_slam_amd/dataset_problem_phong.hpp"

using ceres_slam::uint;

// tests/dataset_ba_phong.cpp:26-255.  false: the joint solve failed.
static bool solveWindow(ceres_slam::DatasetProblemPhong &dataset, uint k1, uint k2, bool use_light, bool multi_stage) {
    std::cerr << "Working on interval [" << k1 << "," << k2 << ")" << std::endl;
    ceres::Problem problem;
    double stereo_stiffness[9] = {0}, normal_stiffness[9] = {0};
    for (int c = 0; c < 3; ++c) {
        stereo_stiffness[4 * c] = 1.0 / std::sqrt(dataset.stereo_obs_var.data()[c]);      // :34-37
        normal_stiffness[4 * c] = 1.0 / std::sqrt(dataset.normal_obs_var.data()[c]);      // :39-42
    }
    const double int_stiffness = 1.0 / std::sqrt(dataset.int_var);                        // :44
    ceres::LocalParameterization *se3_perturbation = ceres_slam::SE3Perturbation::Create();
    ceres::LocalParameterization *unit_vector_perturbation = ceres_slam::UnitVectorPerturbation::Create();
    ceres::Solver::Summary summary;

    for (uint k = k1; k < k2; ++k) {                                    // stereo terms (:53-73)
        for (uint i : dataset.obs_indices_at_state((int)k)) {
            const uint j = dataset.vertex_ids[i];
            if (!dataset.initialized_vertex[j]) continue;
            ceres::CostFunction *stereo_cost =
                ceres_slam::StereoReprojectionErrorAutomatic::Create(dataset.camera, dataset.stereo_obs_list[i].data(), stereo_stiffness);
            problem.AddResidualBlock(stereo_cost, NULL, dataset.poses[k].data(), dataset.map_vertices[j].position().data());
            problem.SetParameterization(dataset.poses[k].data(), se3_perturbation);
        }
    }
    problem.SetParameterBlockConstant(dataset.poses[k1].data());        // :76

    ceres::Solver::Options solver_options;                              // :79-87
    solver_options.minimizer_progress_to_stdout = false;
    solver_options.num_threads = 8;
    solver_options.num_linear_solver_threads = 8;
    solver_options.max_num_iterations = 1000;
    solver_options.use_nonmonotonic_steps = true;
    solver_options.trust_region_strategy_type = ceres::DOGLEG;
    solver_options.dogleg_type = ceres::SUBSPACE_DOGLEG;
    solver_options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;

    if (multi_stage) {                                                  // stage 1 (:96-100): poses and points only, no lighting
        std::cerr << "Solving stage 1: poses and points" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
    }

    if (use_light) {                                                    // lighting terms (:102-207)
        for (uint k = k1; k < k2; ++k) {
            for (uint i : dataset.obs_indices_at_state((int)k)) {
                const uint j = dataset.vertex_ids[i];
                if (!dataset.initialized_vertex[j]) continue;
                ceres_slam::Vertex &vertex = dataset.map_vertices[j];
                double *phong = vertex.material()->phong_params().data(), *texture = vertex.texture()->data();      // :117-121
                ceres::CostFunction *intensity_cost =
                    dataset.directional_light ? ceres_slam::IntensityErrorDirectionalLightAutomatic::Create(dataset.int_list[i], int_stiffness)
                                              : ceres_slam::IntensityErrorPointLightAutomatic::Create(dataset.int_list[i], int_stiffness);
                problem.AddResidualBlock(intensity_cost, NULL, dataset.poses[k].data(), vertex.position().data(), vertex.normal().data(), phong,
                                         texture, dataset.light_data());
                problem.SetParameterLowerBound(phong, 0, 0.);           // :143-165
                problem.SetParameterUpperBound(phong, 0, 1.);
                problem.SetParameterLowerBound(phong, 1, 0.);
                problem.SetParameterUpperBound(phong, 1, 1.);
                problem.SetParameterLowerBound(phong, 2, 1.);
                problem.SetParameterLowerBound(texture, 0, 0.);         // :175-178
                problem.SetParameterUpperBound(texture, 0, 1.);
                ceres::CostFunction *normal_cost = ceres_slam::NormalErrorAutomatic::Create(dataset.normal_obs_list[i].data(), normal_stiffness);
                problem.AddResidualBlock(normal_cost, NULL, dataset.poses[k].data(), vertex.normal().data());
                problem.SetParameterization(vertex.normal().data(), unit_vector_perturbation);
            }
        }
        if (dataset.directional_light) problem.SetParameterization(dataset.light_data(), unit_vector_perturbation);   // :201-204
    }

    if (multi_stage) {                                                  // stage 2 (:210-246): lighting only
        for (uint k = k1; k < k2; ++k) {
            for (uint i : dataset.obs_indices_at_state((int)k)) {
                const uint j = dataset.vertex_ids[i];
                if (dataset.initialized_vertex[j]) problem.SetParameterBlockConstant(dataset.map_vertices[j].position().data());
            }
            if (!dataset.obs_indices_at_state((int)k).empty()) problem.SetParameterBlockConstant(dataset.poses[k].data());
        }
        std::cerr << "Solving stage 2: lighting" << std::endl;
        ceres::Solve(solver_options, &problem, &summary);
        std::cout << summary.BriefReport() << std::endl << std::endl;
        if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;
        for (uint k = k1; k < k2; ++k) {
            for (uint i : dataset.obs_indices_at_state((int)k)) {
                const uint j = dataset.vertex_ids[i];
                if (dataset.initialized_vertex[j]) problem.SetParameterBlockVariable(dataset.map_vertices[j].position().data());
            }
            if (k > k1 && !dataset.obs_indices_at_state((int)k).empty()) problem.SetParameterBlockVariable(dataset.poses[k].data());
        }
    }

    std::cerr << "Solving SLAM and lighting jointly" << std::endl;      // :249-252
    ceres::Solve(solver_options, &problem, &summary);
    std::cout << summary.BriefReport() << std::endl << std::endl;
    if (summary.termination_type == ceres::FAILURE && !summary.message.empty()) std::cerr << summary.message << std::endl;
    return summary.termination_type != ceres::FAILURE;
}

// tests/dataset_ba_phong.cpp:257-334
int main(int argc, char **argv) {
    const bool use_frontend = argc >= 3 && std::string(argv[2]) == "--frontend";
    if (argc < 5 && !use_frontend) {
        std::cerr << "usage: dataset_ba_phong_gpu <dataset.csv> <init_poses.csv> <init_map.csv> <init_lights.csv> [--nolight | --dirlight] [--multistage] [--window N]\n"
                     "       dataset_ba_phong_gpu <dataset.csv> --frontend [--nolight | --dirlight] [--multistage] [--window N]" << std::endl;
        return EXIT_FAILURE;
    }
    bool use_light = true, directional_light = false, multi_stage = false;
    uint window_size = 0;
    for (int a = use_frontend ? 3 : 5; a < argc; ++a) {
        if (std::string(argv[a]) == "--nolight") use_light = false;
        if (std::string(argv[a]) == "--dirlight") directional_light = true;
        if (std::string(argv[a]) == "--multistage") multi_stage = true;
        if (std::string(argv[a]) == "--window" && a + 1 < argc) window_size = (uint)std::atoi(argv[++a]);
    }
    const std::string filename(argv[1]);
    ceres_slam::DatasetProblemPhong dataset(directional_light);
    if (!dataset.read_csv(filename)) return EXIT_FAILURE;
    if (use_frontend) {
        if (!dataset.compute_initial_guess()) return EXIT_FAILURE;                     // :303-311
    } else if (!dataset.read_initial_guess(argv[2], argv[3], argv[4])) {
        return EXIT_FAILURE;
    }
    if (window_size == 0 || window_size > dataset.num_states) window_size = dataset.num_states;      // :313-315
    bool ok = true;
    for (uint k1 = 0; k1 + window_size <= dataset.num_states && ok; ++k1)             // :317-327
        ok = solveWindow(dataset, k1, k1 + window_size, use_light, multi_stage);
    for (int a = 1; a < argc; ++a)          // --refprecision: the reference's four significant digits (utils/utils.hpp:34) instead of 17
        if (std::string(argv[a]) == "--refprecision") dataset.csv_precision = ceres_slam::DatasetProblemPhong::kReferenceCsvPrecision;
    dataset.write_csv(filename);
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}

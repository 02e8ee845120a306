// DatasetProblemPhong -- the host-side data model of the reference's Phong bundle-adjustment driver
// (/root/reference include/ceres_slam/dataset_problem_phong.hpp:20-118, src/ceres_slam/dataset_problem_phong.cpp), with
// the same public fields and method set, on the plain value types of dataset_problem.hpp.  It owns the parameter
// blocks the solve mutates in place, at the addresses the reference's solveWindow hands to Ceres
// (tests/dataset_ba_phong.cpp:53-139): poses[k].data() (12), map_vertices[j].position().data() (3),
// map_vertices[j].normal().data() (3), map_vertices[j].material()->phong_params().data() (3: ka, ks, exponent),
// map_vertices[j].texture()->data() (1: kd) -- materials and textures shared by all vertices of a material --,
// light_pos.data() / light_dir.data() (3).  Any number of materials (the file says how many, :266-278).
//
// compute_initial_guess follows dataset_problem_phong.cpp:251-391: materials at (0, 0, 1), textures at the median
// intensity of the material, stereo VO by RANSAC alignment of consecutive states (threshold 9 px^2, all state pairs
// of the call in one GPU batch), vertices initialised through poses[k-1]^-1 (normals: rotation only), including the
// reference's indexing of `material_ids` by the position in the pair's match list (:369-370).
#pragma once
#include <algorithm>

#include "ceres_slam_amd/dataset_problem.hpp"

namespace ceres_slam {

//! Phong reflectance parameters [ka, ks, exponent] (lighting/material.hpp:11-65)
struct Material {
    typedef std::shared_ptr<Material> Ptr;
    Point params;
    Point &phong_params() { return params; }
    const Point &phong_params() const { return params; }
};
//! Diffuse texture value kd (lighting/texture.hpp:10-34)
struct Texture {
    typedef std::shared_ptr<Texture> Ptr;
    double kd = 0.0;
    double *data() { return &kd; }
    const double *data() const { return &kd; }
};
//! Position + normal + shared material / texture (lighting/vertex3d.hpp:21-97)
struct Vertex {
    Point pos, nrm;
    Material::Ptr mat;
    Texture::Ptr tex;
    Point &position() { return pos; }
    const Point &position() const { return pos; }
    Vector &normal() { return nrm; }
    const Vector &normal() const { return nrm; }
    Material::Ptr &material() { return mat; }
    const Material::Ptr &material() const { return mat; }
    Texture::Ptr &texture() { return tex; }
    const Texture::Ptr &texture() const { return tex; }
};

//! Class for reading simulated datasets with lighting from file (dataset_problem_phong.hpp:20)
class DatasetProblemPhong {
 public:
    typedef StereoCamera Camera;
    typedef std::shared_ptr<const Camera> CameraPtr;
    typedef DatasetProblem::Observation Observation;

    explicit DatasetProblemPhong(bool dir_light = false)
        : num_states(0), num_vertices(0), num_materials(0), directional_light(dir_light), int_var(0.0) {}

    CameraPtr camera;
    //! Timestamps (measured), one per observation row
    std::vector<double> t;
    uint num_states, num_vertices, num_materials;
    //! Camera poses in base frame (to be estimated)
    std::vector<SE3> poses;
    //! Map vertices in base frame (to be estimated), vertex / material ID of every observation
    std::vector<Vertex> map_vertices;
    std::vector<uint> vertex_ids;
    std::vector<bool> initialized_vertex;
    std::vector<uint> material_ids;
    //! Light source: position (point light) or direction (directional light) in base frame (to be estimated)
    bool directional_light;
    Point light_pos;
    Vector light_dir;
    //! Materials and textures (to be estimated)
    std::vector<Material::Ptr> materials;
    std::vector<Texture::Ptr> textures;
    //! Observations and their variances
    std::vector<Observation> stereo_obs_list;
    Observation stereo_obs_var;
    std::vector<double> int_list;
    double int_var;
    std::vector<Vector> normal_obs_list;
    Vector normal_obs_var;

    //! The light block the driver optimises
    double *light_data() { return directional_light ? light_dir.data() : light_pos.data(); }

    //! Rows: "num_states,num_vertices,num_materials" | "fu,fv,cu,cv,b" | variances "stereo (3), normal (3), intensity" |
    //! light position or direction | first pose (4x4 row-major) | observations "t,j,material,u,v,d,I,nx,ny,nz"
    //! (dataset_problem_phong.cpp:16-173).  A new timestamp starts the next state (:120-133).
    bool read_csv(const std::string &filename) {
        std::ifstream f(filename);
        if (!f.is_open()) { std::cerr << "Error: Couldn't open file " << filename << std::endl; return false; }
        std::string line;
        std::getline(f, line); const std::vector<double> meta = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> intr = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> var = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> light = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> first = detail::parse_row(line);
        if (meta.size() < 3 || intr.size() < 5 || var.size() < 7 || light.size() < 3 || first.size() < 16) {
            std::cerr << "Error: malformed header in " << filename << std::endl;
            return false;
        }
        num_states = (uint)meta[0]; num_vertices = (uint)meta[1]; num_materials = (uint)meta[2];
        camera = std::make_shared<const Camera>(intr[0], intr[1], intr[2], intr[3], intr[4]);
        for (int c = 0; c < 3; ++c) { stereo_obs_var.v[c] = var[c]; normal_obs_var.v[c] = var[3 + c]; }
        int_var = var[6];
        light_pos = Point(light[0], light[1], light[2]);
        light_dir = light_pos;
        poses.assign(num_states, SE3());
        if (num_states) poses[0] = SE3::from_rows(first.data());
        materials.clear(); textures.clear();
        for (uint m = 0; m < num_materials; ++m) { materials.push_back(std::make_shared<Material>()); textures.push_back(std::make_shared<Texture>()); }
        map_vertices.assign(num_vertices, Vertex());
        initialized_vertex.assign(num_vertices, false);
        t.clear(); vertex_ids.clear(); material_ids.clear(); stereo_obs_list.clear(); int_list.clear(); normal_obs_list.clear();
        state_of_.clear();
        while (std::getline(f, line)) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 10) continue;
            if (!t.empty() && r[0] != t.back()) state_of_.push_back(state_of_.back() + 1);
            else state_of_.push_back(t.empty() ? 0 : state_of_.back());
            t.push_back(r[0]);
            vertex_ids.push_back((uint)r[1]);
            material_ids.push_back((uint)r[2]);
            stereo_obs_list.push_back(Observation{{r[3], r[4], r[5]}});
            int_list.push_back(r[6]);
            normal_obs_list.push_back(Vector(r[7], r[8], r[9]));
        }
        // every vertex points at the material / texture its observations name
        for (size_t i = 0; i < vertex_ids.size(); ++i)
            if (vertex_ids[i] < num_vertices && material_ids[i] < num_materials) set_material(vertex_ids[i], material_ids[i]);
        build_indices();
        return true;
    }

    //! The `_poses.csv` / `_map.csv` / `_lights.csv` triple write_csv emits, as the initial guess (not in the reference)
    bool read_initial_guess(const std::string &poses_file, const std::string &map_file, const std::string &lights_file) {
        std::ifstream pf(poses_file), mf(map_file), lf(lights_file);
        if (!pf.is_open() || !mf.is_open() || !lf.is_open()) { std::cerr << "Error: Couldn't open the initial-guess files" << std::endl; return false; }
        std::string line;
        std::getline(pf, line);   // header
        for (uint k = 0; k < num_states && std::getline(pf, line); ++k) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 16) { std::cerr << "malformed pose row" << std::endl; return false; }
            poses[k] = SE3::from_rows(r.data());
        }
        std::getline(mf, line);
        while (std::getline(mf, line)) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 11 || (uint)r[0] >= num_vertices) continue;
            Vertex &v = map_vertices[(uint)r[0]];
            v.pos = Point(r[1], r[2], r[3]);
            v.nrm = Vector(r[4], r[5], r[6]);
            if (v.mat) v.mat->params = Point(r[7], r[8], r[9]);
            if (v.tex) v.tex->kd = r[10];
            initialized_vertex[(uint)r[0]] = true;
        }
        std::getline(lf, line);
        std::getline(lf, line);
        const std::vector<double> r = detail::parse_row(line);
        if (r.size() < 3) { std::cerr << "malformed light row" << std::endl; return false; }
        light_pos = Point(r[0], r[1], r[2]);
        light_dir = light_pos;
        return true;
    }

    //! Significant digits write_csv prints: 17 round-trips a double, kReferenceCsvPrecision = 4 is the reference's
    //! Eigen::IOFormat(4, ...) (utils/utils.hpp:34) for byte-compatible output files.
    static constexpr int kReferenceCsvPrecision = 4;
    int csv_precision = 17;

    //! `<stem>_poses.csv`, `_map.csv`, `_lights.csv` (dataset_problem_phong.cpp:175-235)
    bool write_csv(const std::string &filename) const {
        const std::string stem = detail::stem(filename);
        std::ofstream pose_file(stem + "_poses.csv"), map_file(stem + "_map.csv"), light_file(stem + "_lights.csv");
        if (!pose_file.is_open() || !map_file.is_open() || !light_file.is_open()) return false;
        map_file.precision(csv_precision); light_file.precision(csv_precision);
        pose_file << "T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33" << std::endl;
        for (const SE3 &T : poses) pose_file << T.str(csv_precision) << std::endl;
        map_file << "point_id, x, y, z, nx, ny, nz, ka, ks, exponent, kd" << std::endl;
        for (uint j = 0; j < num_vertices; ++j) {
            if (!initialized_vertex[j]) continue;
            const Vertex &v = map_vertices[j];
            const Point ph = v.mat ? v.mat->params : Point();
            map_file << j << "," << v.pos(0) << "," << v.pos(1) << "," << v.pos(2) << "," << v.nrm(0) << "," << v.nrm(1) << "," << v.nrm(2) << ","
                     << ph(0) << "," << ph(1) << "," << ph(2) << "," << (v.tex ? v.tex->kd : 0.0) << std::endl;
        }
        const Point &l = directional_light ? light_dir : light_pos;
        light_file << (directional_light ? "i, j, k" : "x, y, z") << std::endl;
        light_file << l(0) << "," << l(1) << "," << l(2) << std::endl;
        return true;
    }

    const std::vector<uint> &obs_indices_at_state(int k) const { return state_indices_.at(k); }
    const std::vector<uint> &obs_indices_for_feature(int j) const { return feature_indices_.at(j); }
    const std::vector<uint> &obs_indices_for_material(int m) const { return material_indices_.at(m); }

    //! Initial guess (dataset_problem_phong.cpp:251-391); k2 = 0 means all states.  The material / texture
    //! initialisation (:264-277) runs when the call starts at the first state.
    bool compute_initial_guess(uint k1 = 0, uint k2 = 0) {
        if (k2 == 0) k2 = num_states;
        if (k1 == 0) {
            for (uint m = 0; m < num_materials; ++m) {
                materials[m]->params = Point(0.0, 0.0, 1.0);
                std::vector<double> ints;
                for (uint i : material_indices_[m]) ints.push_back(int_list[i]);
                if (ints.empty()) continue;
                std::nth_element(ints.begin(), ints.begin() + ints.size() / 2, ints.end());
                textures[m]->kd = ints[ints.size() / 2];
            }
        }
        if (k2 <= k1 + 1) return true;
        const Camera &cam = *camera;
        const uint32_t num_iters = 400;
        std::vector<uint32_t> offset(1, 0), samples;
        std::vector<double> pts0, pts1;
        std::vector<uint> match_km1;
        for (uint k = k1 + 1; k < k2; ++k) {                                 // :279-331
            std::map<uint, uint> in_k;
            for (uint i : state_indices_[k]) in_k[vertex_ids[i]] = i;
            std::vector<uint> a, b;
            std::map<uint, int> kept;
            for (uint i : state_indices_[k - 1])
                if (in_k.count(vertex_ids[i])) { a.push_back(i); kept[vertex_ids[i]] = 1; }
            for (uint i : state_indices_[k])
                if (kept.count(vertex_ids[i])) b.push_back(i);
            if (a.size() < 3 || a.size() != b.size()) { std::cerr << "state " << k << ": fewer than 3 matches" << std::endl; return false; }
            for (size_t m = 0; m < a.size(); ++m) {
                const Point p0 = detail::triangulate(cam, stereo_obs_list[a[m]].data()), p1 = detail::triangulate(cam, stereo_obs_list[b[m]].data());
                pts0.insert(pts0.end(), p0.v, p0.v + 3);
                pts1.insert(pts1.end(), p1.v, p1.v + 3);
                match_km1.push_back(a[m]);
            }
            offset.push_back((uint32_t)(pts0.size() / 3));
            std::vector<uint32_t> smp(3 * num_iters);
            if (ssba_ransac_samples((uint32_t)a.size(), num_iters, __GNUC__ >= 11 ? 1 : 0, smp.data())) return false;
            samples.insert(samples.end(), smp.begin(), smp.end());
        }
        const uint32_t num_pairs = k2 - k1 - 1;
        std::vector<double> T((size_t)num_pairs * 12);
        std::vector<uint8_t> inlier(pts0.size() / 3);
        ssba_camera c = {cam.fu, cam.fv, cam.cu, cam.cv, cam.b};
        const int rc = ssba_frontend_ransac(&c, -1, num_pairs, offset.data(), pts0.data(), pts1.data(), samples.data(), num_iters, 9.0,
                                            T.data(), inlier.data(), nullptr, nullptr);      // :340-345, threshold 9 px^2
        if (rc) { std::cerr << "ssba_frontend_ransac: " << ssba_status_string(rc) << std::endl; return false; }
        for (uint k = k1 + 1; k < k2; ++k) {
            const uint q = k - k1 - 1;
            SE3 T_k_km1;
            for (int i = 0; i < 12; ++i) T_k_km1.v[i] = T[12 * (size_t)q + i];
            const SE3 T_km1 = poses[k - 1];
            poses[k] = T_k_km1 * T_km1;                                      // :352
            const SE3 T_0_km1 = T_km1.inverse();
            for (uint32_t m = offset[q]; m < offset[q + 1]; ++m) {           // :356-389
                const uint j = vertex_ids[match_km1[m]];
                if (!inlier[m] || j >= num_vertices || initialized_vertex[j]) continue;
                Vertex &v = map_vertices[j];
                v.pos = T_0_km1 * Point(pts0[3 * (size_t)m], pts0[3 * (size_t)m + 1], pts0[3 * (size_t)m + 2]);
                const Vector &n = normal_obs_list[match_km1[m]];
                for (int cc = 0; cc < 3; ++cc) v.nrm.v[cc] = T_km1.v[3 + cc] * n(0) + T_km1.v[6 + cc] * n(1) + T_km1.v[9 + cc] * n(2);   // R^T n
                // the reference indexes material_ids with the position in the pair's match list, not the observation (:369-370)
                const uint mid = material_ids[m - offset[q]];
                if (mid < num_materials) set_material(j, mid);
                initialized_vertex[j] = true;
            }
        }
        return true;
    }

 private:
    void set_material(uint j, uint m) { map_vertices[j].mat = materials[m]; map_vertices[j].tex = textures[m]; }
    void build_indices() {
        state_indices_.assign(num_states, std::vector<uint>());
        feature_indices_.assign(num_vertices, std::vector<uint>());
        material_indices_.assign(num_materials, std::vector<uint>());
        for (uint i = 0; i < state_of_.size(); ++i) {
            if (state_of_[i] < num_states) state_indices_[state_of_[i]].push_back(i);
            if (vertex_ids[i] < num_vertices) feature_indices_[vertex_ids[i]].push_back(i);
            if (material_ids[i] < num_materials) material_indices_[material_ids[i]].push_back(i);
        }
    }
    std::vector<uint> state_of_;        // state index of every observation row
    std::vector<std::vector<uint>> state_indices_, feature_indices_, material_indices_;
};

}  // namespace ceres_slam

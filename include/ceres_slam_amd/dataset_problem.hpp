// DatasetProblem -- the host-side data model of the reference's stereo drivers
// (/root/reference include/ceres_slam/dataset_problem.hpp:14-82, src/ceres_slam/dataset_problem.cpp), with the same
// public fields and the same method set (read_csv / write_csv / obs_indices_at_state / obs_indices_for_feature /
// reset_points / compute_initial_guess), written against plain arrays instead of Eigen.  It owns the parameter memory the
// solve mutates in place: poses[k].data() is the 12-double [t | R row-major] block of SE3Group
// (geometry/se3group.hpp:425-429), map_points[j].data() the 3 doubles of Point3D -- exactly the pointers the reference's
// solveWindow hands to ceres::Problem::AddResidualBlock (tests/dataset_vo.cpp:41-56).
//
// compute_initial_guess does what dataset_problem.cpp:179-270 does (reciprocal matches of consecutive states,
// StereoCamera::triangulate, 3-point RANSAC alignment, pose chaining, map initialisation from the inliers), all of it
// on the device in one call (ssba_frontend_vo) for all state pairs of the range at once; the draw sequence of
// std::mt19937(42) + std::uniform_int_distribution (point_cloud_aligner.cpp:70-76) is reproduced there.
//
// Beyond the reference: read_initial_guess() loads the `_poses.csv` / `_map.csv` pair that write_csv emits (so that a
// saved state can be resumed), and write_csv prints full double precision (the reference's IOFormat(4) is lossy,
// utils.hpp:34).
#pragma once
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "ceres_slam_amd/ceres_shim.hpp"

namespace ceres_slam {

typedef unsigned int uint;

// ---- minimal value types exposing .data() (the reference's are Eigen subclasses) ------------------------------------
//! 3-vector (Point3D / Vector3D, geometry/point3d.hpp, vector3d.hpp)
struct Point {
    double v[3];
    Point() : v{0.0, 0.0, 0.0} {}
    Point(double x, double y, double z) : v{x, y, z} {}
    double *data() { return v; }
    const double *data() const { return v; }
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
typedef Point Vector;

//! SE(3) element stored as the reference stores it: [t (3) | R row-major (9)] (geometry/se3group.hpp:425-429)
struct SE3 {
    double v[12];
    SE3() : v{0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1} {}
    double *data() { return v; }
    const double *data() const { return v; }
    //! from a row-major 4x4 (the CSV form)
    static SE3 from_rows(const double *m16) {
        SE3 T;
        T.v[0] = m16[3]; T.v[1] = m16[7]; T.v[2] = m16[11];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T.v[3 + 3 * i + j] = m16[4 * i + j];
        return T;
    }
    //! composition (se3group.hpp:176-183): R <- R1 R2, t <- R1 t2 + t1
    SE3 operator*(const SE3 &o) const {
        SE3 T;
        for (int i = 0; i < 3; ++i) {
            T.v[i] = v[3 + 3 * i] * o.v[0] + v[4 + 3 * i] * o.v[1] + v[5 + 3 * i] * o.v[2] + v[i];
            for (int j = 0; j < 3; ++j) T.v[3 + 3 * i + j] = v[3 + 3 * i] * o.v[3 + j] + v[4 + 3 * i] * o.v[6 + j] + v[5 + 3 * i] * o.v[9 + j];
        }
        return T;
    }
    //! transform of a point (se3group.hpp:191-193): R p + t
    Point operator*(const Point &p) const {
        Point q;
        for (int i = 0; i < 3; ++i) q.v[i] = v[3 + 3 * i] * p.v[0] + v[4 + 3 * i] * p.v[1] + v[5 + 3 * i] * p.v[2] + v[i];
        return q;
    }
    //! inverse (se3group.hpp:152-158): (R^T, -R^T t)
    SE3 inverse() const {
        SE3 T;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T.v[3 + 3 * i + j] = v[3 + 3 * j + i];
            T.v[i] = -(v[3 + i] * v[0] + v[6 + i] * v[1] + v[9 + i] * v[2]);
        }
        return T;
    }
    //! one CSV row: the 4x4 matrix row-major
    std::string str(int precision = 17) const {
        std::ostringstream ss;
        ss.precision(precision);
        for (int i = 0; i < 3; ++i) ss << v[3 + 3 * i] << "," << v[4 + 3 * i] << "," << v[5 + 3 * i] << "," << v[i] << ",";
        ss << "0,0,0,1";
        return ss.str();
    }
};

namespace detail {
inline std::vector<double> parse_row(const std::string &line) {
    std::vector<double> r;
    std::stringstream ss(line);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        try { r.push_back(std::stod(tok)); } catch (...) { r.clear(); return r; }
    }
    return r;
}
inline std::string stem(const std::string &filename) { return filename.substr(0, filename.find_last_of('.')); }
//! StereoCamera::triangulate (stereo_camera.hpp:112-120)
inline Point triangulate(const StereoCamera &c, const double *uvd) {
    const double b_over_d = c.b / uvd[2];
    return Point((uvd[0] - c.cu) * b_over_d, (uvd[1] - c.cv) * b_over_d * (c.fu / c.fv), c.fu * b_over_d);
}
}  // namespace detail

//! Class for reading simulated datasets from file (dataset_problem.hpp:14)
class DatasetProblem {
 public:
    typedef StereoCamera Camera;
    typedef std::shared_ptr<const Camera> CameraPtr;
    //! (u, v, d) (stereo_camera.hpp:20-24)
    struct Observation {
        double v[3];
        double *data() { return v; }
        const double *data() const { return v; }
    };
    typedef Observation ObservationVariance;

    DatasetProblem() : num_states(0), num_points(0) {}

    //! Camera model
    CameraPtr camera;
    //! Pose ID of every observation
    std::vector<uint> state_ids;
    //! Number of states / map points to optimize
    uint num_states, num_points;
    //! Camera poses in base frame (to be estimated)
    std::vector<SE3> poses;
    //! Map points in base frame (to be estimated)
    std::vector<Point> map_points;
    //! Map point ID of every observation
    std::vector<uint> point_ids;
    //! True if map point j has been initialized
    std::vector<bool> initialized_point;
    //! List of stereo observations, their (shared) variance
    std::vector<Observation> stereo_obs_list;
    ObservationVariance stereo_obs_var;

    //! Read dataset from a CSV file: row 1 "num_states,num_points", row 2 intrinsics "fu,fv,cu,cv,b", row 3 observation
    //! variances, row 4 the first pose (4x4 row-major), then observation rows "k,j,u,v,d" (dataset_problem.cpp:16-121)
    bool read_csv(const std::string &filename) {
        std::ifstream f(filename);
        if (!f.is_open()) { std::cerr << "Error: Couldn't open file " << filename << std::endl; return false; }
        std::string line;
        std::getline(f, line); const std::vector<double> meta = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> intr = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> var = detail::parse_row(line);
        std::getline(f, line); const std::vector<double> first = detail::parse_row(line);
        if (meta.size() < 2 || intr.size() < 5 || var.size() < 3 || first.size() < 16) {
            std::cerr << "Error: malformed header in " << filename << std::endl;
            return false;
        }
        num_states = (uint)meta[0];
        num_points = (uint)meta[1];
        camera = std::make_shared<const Camera>(intr[0], intr[1], intr[2], intr[3], intr[4]);
        for (int c = 0; c < 3; ++c) stereo_obs_var.v[c] = var[c];
        poses.assign(num_states, SE3());
        if (num_states) poses[0] = SE3::from_rows(first.data());       // :60-66
        map_points.assign(num_points, Point());
        initialized_point.assign(num_points, false);
        state_ids.clear(); point_ids.clear(); stereo_obs_list.clear();
        while (std::getline(f, line)) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 5) continue;
            state_ids.push_back((uint)r[0]);
            point_ids.push_back((uint)r[1]);
            stereo_obs_list.push_back(Observation{{r[2], r[3], r[4]}});
        }
        build_indices();
        return true;
    }

    //! The `_poses.csv` / `_map.csv` pair write_csv emits, as the initial guess (not in the reference)
    bool read_initial_guess(const std::string &poses_file, const std::string &map_file) {
        std::ifstream pf(poses_file), mf(map_file);
        if (!pf.is_open()) { std::cerr << "Error: Couldn't open file " << poses_file << std::endl; return false; }
        if (!mf.is_open()) { std::cerr << "Error: Couldn't open file " << map_file << std::endl; return false; }
        std::string line;
        uint k = 0;
        while (std::getline(pf, line) && k < num_states) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 16) continue;      // header
            poses[k++] = SE3::from_rows(r.data());
        }
        while (std::getline(mf, line)) {
            const std::vector<double> r = detail::parse_row(line);
            if (r.size() < 4 || (uint)r[0] >= num_points) continue;
            const uint j = (uint)r[0];
            map_points[j] = Point(r[1], r[2], r[3]);
            initialized_point[j] = true;
        }
        return true;
    }

    //! Significant digits write_csv prints.  17 (the default) round-trips a double; kReferenceCsvPrecision = 4 is what the
    //! reference writes (Eigen::IOFormat(4, ...), utils/utils.hpp:34) -- set it for byte-compatible output files.
    static constexpr int kReferenceCsvPrecision = 4;
    int csv_precision = 17;

    //! Write result to `<stem>_poses.csv` and `<stem>_map.csv` (dataset_problem.cpp:121-165)
    bool write_csv(const std::string &filename) const {
        const std::string stem = detail::stem(filename);
        std::ofstream pose_file(stem + "_poses.csv"), map_file(stem + "_map.csv");
        if (!pose_file.is_open() || !map_file.is_open()) { std::cerr << "Error: Couldn't open output files for " << stem << std::endl; return false; }
        pose_file << "T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33" << std::endl;
        for (const SE3 &T : poses) pose_file << T.str(csv_precision) << std::endl;
        map_file.precision(csv_precision);
        map_file << "point_id, x, y, z" << std::endl;
        for (uint j = 0; j < map_points.size(); ++j)
            if (initialized_point[j]) map_file << j << "," << map_points[j](0) << "," << map_points[j](1) << "," << map_points[j](2) << std::endl;
        return true;
    }

    //! Observation indices of state k / of feature j (dataset_problem.cpp:167-173)
    const std::vector<uint> &obs_indices_at_state(uint k) const { return state_indices_.at(k); }
    const std::vector<uint> &obs_indices_for_feature(uint j) const { return feature_indices_.at(j); }

    //! Reset initialization flags for all points (dataset_problem.cpp:175-177)
    void reset_points() { initialized_point.assign(num_points, false); }

    //! Initial guess for the poses of states (k1, k2) and the map points they see, by stereo VO
    //! (dataset_problem.cpp:179-270); k2 = 0 means all states.  Returns false if a state pair has fewer than three
    //! matches or the device is not available (the reference would carry on with garbage).
    bool compute_initial_guess(uint k1 = 0, uint k2 = 0) {
        if (k1 >= k2) { k1 = 0; k2 = num_states; }                                // :184-187
        if (k2 <= k1 + 1) return true;
        // the observations of the states [k1, k2), grouped by state in file order: what obs_indices_at_state walks (:197-206)
        const uint ns = k2 - k1;
        std::vector<uint32_t> state_start(ns + 1, 0), ids;
        std::vector<double> uvd;
        for (uint k = k1; k < k2; ++k) {
            for (uint i : state_indices_[k]) {
                ids.push_back(point_ids[i]);
                uvd.insert(uvd.end(), stereo_obs_list[i].v, stereo_obs_list[i].v + 3);
            }
            state_start[k - k1 + 1] = (uint32_t)ids.size();
        }
        // everything else -- matching, triangulation, 400-hypothesis RANSAC per pair (threshold 4 px^2, :246-249), pose
        // chaining (:255) and map initialisation (:259-269) -- runs on the device in one call
        std::vector<double> P((size_t)ns * 12), M((size_t)num_points * 3);
        std::vector<uint8_t> init(num_points);
        for (int c = 0; c < 12; ++c) P[c] = poses[k1].v[c];
        for (uint j = 0; j < num_points; ++j) {
            init[j] = initialized_point[j] ? 1 : 0;
            for (int c = 0; c < 3; ++c) M[3 * (size_t)j + c] = map_points[j].v[c];
        }
        const Camera &cam = *camera;
        ssba_camera c = {cam.fu, cam.fv, cam.cu, cam.cv, cam.b};
        const int rc = ssba_frontend_vo(&c, -1, ns, state_start.data(), ids.data(), uvd.data(), num_points, 400, 4.0, __GNUC__ >= 11 ? 1 : 0,
                                        P.data(), M.data(), init.data(), nullptr, nullptr, nullptr);
        if (rc) { std::cerr << "ssba_frontend_vo: " << ssba_status_string(rc) << std::endl; return false; }
        for (uint k = k1 + 1; k < k2; ++k)
            for (int cc = 0; cc < 12; ++cc) poses[k].v[cc] = P[12 * (size_t)(k - k1) + cc];
        for (uint j = 0; j < num_points; ++j) {
            if (init[j] && !initialized_point[j]) {
                map_points[j] = Point(M[3 * (size_t)j], M[3 * (size_t)j + 1], M[3 * (size_t)j + 2]);
                initialized_point[j] = true;
            }
        }
        return true;
    }

 private:
    void build_indices() {
        state_indices_.assign(num_states, std::vector<uint>());
        feature_indices_.assign(num_points, std::vector<uint>());
        for (uint i = 0; i < state_ids.size(); ++i) {
            if (state_ids[i] < num_states) state_indices_[state_ids[i]].push_back(i);
            if (point_ids[i] < num_points) feature_indices_[point_ids[i]].push_back(i);
        }
    }
    //! Lists of observation indices per state / per feature (one pass; the reference scans every pair)
    std::vector<std::vector<uint>> state_indices_, feature_indices_;
};

}  // namespace ceres_slam

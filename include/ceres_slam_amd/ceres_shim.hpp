// ceres_shim.hpp -- header-only C++11 shim that reproduces, on top of the C ABI in
// include/ssba.h, the call shapes the ceres-slam stereo drivers use against Ceres
// (/root/reference tests/dataset_vo.cpp:22-85):
//
//     ceres::Problem problem;
//     ceres::LocalParameterization *se3 = ceres_slam::SE3Perturbation::Create();
//     ceres::CostFunction *cost = ceres_slam::StereoReprojectionErrorAutomatic::Create(camera, obs, stiffness);
//     problem.AddResidualBlock(cost, NULL, pose_k, point_j);
//     problem.SetParameterization(pose_k, se3);
//     problem.SetParameterBlockConstant(pose_0);
//     ceres::Solver::Options options;  options.max_num_iterations = 1000; ...
//     ceres::Solver::Summary summary;  ceres::Solve(options, &problem, &summary);
//     std::cout << summary.BriefReport();
//
// and of the Phong driver (tests/dataset_ba_phong.cpp:26-255):
//
//     problem.AddResidualBlock(IntensityErrorPointLightAutomatic::Create(I, int_stiffness), NULL,
//                              pose_k, position_j, normal_j, phong_m, texture_m, light);
//     problem.AddResidualBlock(NormalErrorAutomatic::Create(n_obs, normal_stiffness), NULL, pose_k, normal_j);
//     problem.SetParameterization(normal_j, unit_vector_perturbation);
//     problem.SetParameterLowerBound(phong_m, 0, 0.);  problem.SetParameterUpperBound(phong_m, 0, 1.); ...
//     options.trust_region_strategy_type = ceres::DOGLEG;  options.dogleg_type = ceres::SUBSPACE_DOGLEG;
//
// and of the sun-aided VO driver (tests/dataset_vo_sun.cpp:28-185):
//
//     problem.AddResidualBlock(StereoReprojectionErrorAutomatic::Create(camera, obs, stiffness_of_point_j), NULL, pose_k, point_j);
//     problem.AddResidualBlock(SunSensorErrorAutomatic::Create(sun_obs_c, sun_dir_g, stiffness2x2, az_thresh, zen_thresh),
//                              new ceres::HuberLoss(huber_param), pose_k);
//     problem.AddResidualBlock(PoseErrorAutomatic::Create(T_ref, stiffness6x6), NULL, pose_k1);
//     ceres::Covariance covariance(covariance_options);
//     covariance.Compute(covar_blocks, &problem);
//     covariance.GetCovarianceBlockInTangentSpace(pose, pose, out36);
//
// The shim recognises the typed cost functions of this path and lowers them to observation
// tables; it does NOT run arbitrary user functors on the GPU -- any other CostFunction is
// rejected at AddResidualBlock with std::invalid_argument (Ceres would accept it: that is the
// documented limit of the drop-in).  Parameter blocks stay caller-owned; identity = address.
// Eigen is not needed: observations and stiffness are plain arrays.
#ifndef CERES_SLAM_AMD_CERES_SHIM_HPP_
#define CERES_SLAM_AMD_CERES_SHIM_HPP_

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../ssba.h"

namespace ceres {

class CostFunction { public: virtual ~CostFunction() {} };
class LocalParameterization { public: virtual ~LocalParameterization() {} };
class LossFunction { public: virtual ~LossFunction() {} };
class HuberLoss : public LossFunction {
 public:
    explicit HuberLoss(double a) : a_(a) {}
    double a() const { return a_; }
 private:
    double a_;
};

enum TerminationType { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };
enum LinearSolverType { SPARSE_NORMAL_CHOLESKY, SPARSE_SCHUR, DENSE_SCHUR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
enum CovarianceAlgorithmType { DENSE_SVD, SPARSE_QR };
enum SparseLinearAlgebraLibraryType { SUITE_SPARSE, CX_SPARSE, EIGEN_SPARSE, NO_SPARSE };

}  // namespace ceres

namespace ceres_slam {

// include/ceres_slam/stereo_camera.hpp:159-163
struct StereoCamera {
    double fu, fv, cu, cv, b;
    StereoCamera(double fu_, double fv_, double cu_, double cv_, double b_) : fu(fu_), fv(fv_), cu(cu_), cv(cv_), b(b_) {}
};

// include/ceres_slam/stereo_reprojection_error.hpp:59-69
class StereoReprojectionErrorAutomatic : public ceres::CostFunction {
 public:
    static ceres::CostFunction *Create(const std::shared_ptr<const StereoCamera> &camera, const double observation[3],
                                       const double stiffness[9]) {
        StereoReprojectionErrorAutomatic *c = new StereoReprojectionErrorAutomatic;
        c->camera = camera;
        std::memcpy(c->observation, observation, sizeof c->observation);
        std::memcpy(c->stiffness, stiffness, sizeof c->stiffness);
        return c;
    }
    std::shared_ptr<const StereoCamera> camera;
    double observation[3];
    double stiffness[9];
};

// include/ceres_slam/pose_error.hpp:59-66: prior on a pose block, r = S log(T_ref T^-1); T_ref is the 12-double
// [t | R row-major] block (SE3Group::data()), stiffness 6x6 row-major
class PoseErrorAutomatic : public ceres::CostFunction {
 public:
    static ceres::CostFunction *Create(const double T_k_0_ref[12], const double stiffness[36]) {
        PoseErrorAutomatic *c = new PoseErrorAutomatic;
        std::memcpy(c->T_ref, T_k_0_ref, sizeof c->T_ref);
        std::memcpy(c->stiffness, stiffness, sizeof c->stiffness);
        return c;
    }
    double T_ref[12], stiffness[36];
};

// include/ceres_slam/relative_pose_error.hpp:46-57: measurement T_2_1_ref between two pose blocks, r = S log(T_ref T_1 T_2^-1)
class RelativePoseErrorAutomatic : public ceres::CostFunction {
 public:
    static ceres::CostFunction *Create(const double T_2_1_ref[12], const double stiffness[36]) {
        RelativePoseErrorAutomatic *c = new RelativePoseErrorAutomatic;
        std::memcpy(c->T_ref, T_2_1_ref, sizeof c->T_ref);
        std::memcpy(c->stiffness, stiffness, sizeof c->stiffness);
        return c;
    }
    double T_ref[12], stiffness[36];
};

// include/ceres_slam/sun_sensor_error.hpp:108-120: azimuth / zenith error of the expected sun direction, stiffness 2x2
class SunSensorErrorAutomatic : public ceres::CostFunction {
 public:
    static ceres::CostFunction *Create(const double observed_sun_dir_c[3], const double expected_sun_dir_g[3], const double stiffness[4],
                                       double az_err_thresh, double zen_err_thresh) {
        SunSensorErrorAutomatic *c = new SunSensorErrorAutomatic;
        std::memcpy(c->observed, observed_sun_dir_c, sizeof c->observed);
        std::memcpy(c->expected, expected_sun_dir_g, sizeof c->expected);
        std::memcpy(c->stiffness, stiffness, sizeof c->stiffness);
        c->az_thresh = az_err_thresh;
        c->zen_thresh = zen_err_thresh;
        return c;
    }
    double observed[3], expected[3], stiffness[4], az_thresh, zen_thresh;
};

// include/ceres_slam/perturbations.hpp:69-75
class SE3Perturbation : public ceres::LocalParameterization {
 public:
    static ceres::LocalParameterization *Create() { return new SE3Perturbation; }
};

// include/ceres_slam/perturbations.hpp:87-113
class UnitVectorPerturbation : public ceres::LocalParameterization {
 public:
    static ceres::LocalParameterization *Create() { return new UnitVectorPerturbation; }
};

// include/ceres_slam/intensity_error_point_light.hpp:98-112 / intensity_error_directional_light.hpp:98-113:
// blocks (pose 12, position 3, normal 3, Phong parameters 3, texture 1, light 3) -> 1 residual
class IntensityErrorAutomaticBase : public ceres::CostFunction {
 public:
    double colour = 0.0, stiffness = 0.0;
    int light_type = 0;      // 0 point light, 1 directional light
};
class IntensityErrorPointLightAutomatic : public IntensityErrorAutomaticBase {
 public:
    static ceres::CostFunction *Create(double colour, double stiffness) {
        IntensityErrorPointLightAutomatic *c = new IntensityErrorPointLightAutomatic;
        c->colour = colour; c->stiffness = stiffness; c->light_type = 0;
        return c;
    }
};
class IntensityErrorDirectionalLightAutomatic : public IntensityErrorAutomaticBase {
 public:
    static ceres::CostFunction *Create(double colour, double stiffness) {
        IntensityErrorDirectionalLightAutomatic *c = new IntensityErrorDirectionalLightAutomatic;
        c->colour = colour; c->stiffness = stiffness; c->light_type = 1;
        return c;
    }
};

// include/ceres_slam/normal_error.hpp:46-54: blocks (pose 12, normal 3) -> 3 residuals
class NormalErrorAutomatic : public ceres::CostFunction {
 public:
    static ceres::CostFunction *Create(const double obs[3], const double stiffness[9]) {
        NormalErrorAutomatic *c = new NormalErrorAutomatic;
        std::memcpy(c->obs, obs, sizeof c->obs);
        std::memcpy(c->stiffness, stiffness, sizeof c->stiffness);
        return c;
    }
    double obs[3];
    double stiffness[9];
};

}  // namespace ceres_slam

namespace ceres {

class Problem;

class Solver {
 public:
    // the Options fields the drivers set (tests/dataset_vo.cpp:65-74), Ceres 1.x defaults
    struct Options {
        bool minimizer_progress_to_stdout = false;
        int num_threads = 1;
        int num_linear_solver_threads = 1;
        int max_num_iterations = 50;
        bool use_nonmonotonic_steps = false;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        DoglegType dogleg_type = TRADITIONAL_DOGLEG;
        LinearSolverType linear_solver_type = SPARSE_SCHUR;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
        double initial_trust_region_radius = 1e4;
    };
    struct Summary {
        TerminationType termination_type = NO_CONVERGENCE;
        int num_successful_steps = 0, num_unsuccessful_steps = 0;
        int num_line_search_steps = 0;      // bounds: evaluations of the projected line search (device + host)
        double initial_cost = 0, final_cost = 0, total_time_in_seconds = 0;
        std::string message;
        bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE; }
        std::string BriefReport() const {
            ssba_summary s;
            std::memset(&s, 0, sizeof s);
            s.termination_type = termination_type;
            s.num_successful_steps = num_successful_steps;
            s.num_unsuccessful_steps = num_unsuccessful_steps;
            s.initial_cost = initial_cost;
            s.final_cost = final_cost;
            char buf[256];
            ssba_brief_report(&s, buf, sizeof buf);
            return buf;
        }
    };
};

class Problem {
 public:
    Problem() {}
    ~Problem() {   // Ceres default: the problem owns cost functions, losses, parameterisations
        for (auto c : owned_costs_) delete c;
        for (auto &kv : owned_losses_) delete kv.first;
        for (auto &kv : owned_params_) delete kv.first;
        if (h_) ssba_destroy(h_);
    }
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;

    // intensity residual block (tests/dataset_ba_phong.cpp:108-139)
    void AddResidualBlock(CostFunction *cost, LossFunction *loss, double *pose_block, double *position_block, double *normal_block,
                          double *phong_block, double *texture_block, double *light_block) {
        ++version_;
        auto *c = dynamic_cast<ceres_slam::IntensityErrorAutomaticBase *>(cost);
        if (!c) throw std::invalid_argument("ceres_shim: a six-block residual must be an IntensityError*Automatic");
        if (loss) throw std::invalid_argument("ceres_shim: lighting residual blocks take a NULL loss");
        if (!intensity_.empty() && (c->stiffness != intensity_[0].stiffness || c->light_type != intensity_[0].light_type ||
                                    light_block != light_))
            throw std::invalid_argument("ceres_shim: intensity residual blocks must share stiffness, light type and light block");
        light_ = light_block;
        IntensityBlock b;
        b.pose = pose_block; b.position = position_block; b.normal = normal_block; b.phong = phong_block; b.texture = texture_block;
        b.colour = c->colour; b.stiffness = c->stiffness; b.light_type = c->light_type;
        intensity_.push_back(b);
        owned_costs_.push_back(cost);
    }

    // unary pose residual blocks (tests/dataset_vo_sun.cpp:80-124): pose prior, sun sensor (optionally with HuberLoss)
    void AddResidualBlock(CostFunction *cost, LossFunction *loss, double *pose_block) {
        ++version_;
        PoseFactor f;
        std::memset(&f, 0, sizeof f);
        if (auto *c = dynamic_cast<ceres_slam::PoseErrorAutomatic *>(cost)) {
            f.type = 0;
            std::memcpy(f.data, c->T_ref, sizeof c->T_ref);
            std::memcpy(f.stiffness, c->stiffness, sizeof c->stiffness);
        } else if (auto *c2 = dynamic_cast<ceres_slam::SunSensorErrorAutomatic *>(cost)) {
            f.type = 1;
            std::memcpy(f.data, c2->observed, sizeof c2->observed);
            std::memcpy(f.data + 3, c2->expected, sizeof c2->expected);
            f.data[6] = c2->az_thresh; f.data[7] = c2->zen_thresh;
            std::memcpy(f.stiffness, c2->stiffness, sizeof c2->stiffness);
        } else {
            throw std::invalid_argument("ceres_shim: a one-block residual must be a PoseErrorAutomatic or SunSensorErrorAutomatic");
        }
        if (loss) {
            HuberLoss *h = dynamic_cast<HuberLoss *>(loss);
            if (!h) throw std::invalid_argument("ceres_shim: loss must be NULL or ceres::HuberLoss");
            owned_losses_[loss] = 1;
            f.huber = h->a();
        }
        f.pose = pose_block;
        block_index(pose_index_, pose_blocks_, pose_block);
        pose_factors_.push_back(f);
        owned_costs_.push_back(cost);
    }

    void AddResidualBlock(CostFunction *cost, LossFunction *loss, double *pose_block, double *point_block) {
        ++version_;
        if (auto *rp = dynamic_cast<ceres_slam::RelativePoseErrorAutomatic *>(cost)) {      // (pose 1, pose 2): blowup_test.cpp:70-76
            RelFactor f;
            std::memset(&f, 0, sizeof f);
            f.pose1 = pose_block; f.pose2 = point_block;
            std::memcpy(f.T_ref, rp->T_ref, sizeof f.T_ref);
            std::memcpy(f.stiffness, rp->stiffness, sizeof f.stiffness);
            if (loss) {
                HuberLoss *h = dynamic_cast<HuberLoss *>(loss);
                if (!h) throw std::invalid_argument("ceres_shim: loss must be NULL or ceres::HuberLoss");
                owned_losses_[loss] = 1;
                f.huber = h->a();
            }
            block_index(pose_index_, pose_blocks_, pose_block);
            block_index(pose_index_, pose_blocks_, point_block);
            rel_factors_.push_back(f);
            owned_costs_.push_back(cost);
            return;
        }
        if (auto *n = dynamic_cast<ceres_slam::NormalErrorAutomatic *>(cost)) {   // (pose, normal): dataset_ba_phong.cpp:181-188
            if (loss) throw std::invalid_argument("ceres_shim: lighting residual blocks take a NULL loss");
            NormalBlock b;
            b.pose = pose_block; b.normal = point_block;
            std::memcpy(b.obs, n->obs, sizeof b.obs);
            std::memcpy(b.stiffness, n->stiffness, sizeof b.stiffness);
            normals_.push_back(b);
            owned_costs_.push_back(cost);
            return;
        }
        auto *s = dynamic_cast<ceres_slam::StereoReprojectionErrorAutomatic *>(cost);
        if (!s) throw std::invalid_argument("ceres_shim: only StereoReprojectionErrorAutomatic residual blocks run on the GPU path");
        const HuberLoss *h = nullptr;
        if (loss) {
            h = dynamic_cast<HuberLoss *>(loss);
            if (!h) throw std::invalid_argument("ceres_shim: loss must be NULL or ceres::HuberLoss");
            owned_losses_[loss] = 1;
        }
        const double a = h ? h->a() : 0.0;
        if (obs_pose_.empty()) {
            camera_ = s->camera;
            huber_a_ = a;
        } else if (a != huber_a_ ||
                   (s->camera != camera_ && std::memcmp(s->camera.get(), camera_.get(), sizeof(ceres_slam::StereoCamera)) != 0)) {
            throw std::invalid_argument("ceres_shim: all stereo residual blocks must share camera and loss");
        }
        obs_stiffness_.insert(obs_stiffness_.end(), s->stiffness, s->stiffness + 9);    // may differ per block (dataset_vo_sun.cpp:56-65)
        obs_pose_.push_back(block_index(pose_index_, pose_blocks_, pose_block));
        obs_point_.push_back(block_index(point_index_, point_blocks_, point_block));
        obs_uvd_.insert(obs_uvd_.end(), s->observation, s->observation + 3);
        owned_costs_.push_back(cost);
    }
    // Ceres: Problem::AddParameterBlock(values, size, local_parameterization).  Registers a pose block (size 12) or a
    // point block (size 3) ahead of its residual blocks.  Sharded runs (SetDistributed) need it for the poses: every rank
    // must hold every pose, in the same order, also where it owns no observation of it.
    void AddParameterBlock(double *values, int size, LocalParameterization *lp = nullptr) {
        ++version_;
        if (size == 12) block_index(pose_index_, pose_blocks_, values);
        else if (size == 3) block_index(point_index_, point_blocks_, values);
        else throw std::invalid_argument("ceres_shim: AddParameterBlock takes pose (12) or point (3) blocks");
        if (lp) SetParameterization(values, lp);
    }
    // Extension (not in Ceres): this problem is one shard of a landmark-sharded problem -- rank `rank` of `world_size`
    // processes, one per GPU; the shards' reduced camera systems are summed by ncclAllReduce inside libssba.so
    // (ssba_set_distributed + ssba_set_rccl; the 128-byte id comes from ssba_rccl_unique_id on rank 0).  Every rank adds all
    // pose blocks (AddParameterBlock) and the residual blocks of its own landmarks; Solve is then a collective call.
    void SetDistributed(int world_size, int rank, const void *rccl_unique_id) {
        ++version_;
        dist_world_ = world_size; dist_rank_ = rank;
        std::memcpy(dist_id_, rccl_unique_id, sizeof dist_id_);
    }
    void SetParameterization(double *block, LocalParameterization *lp) {
        owned_params_[lp] = 1;
        if (dynamic_cast<ceres_slam::UnitVectorPerturbation *>(lp)) { unit_vector_[block] = 1; return; }   // normals, light direction
        if (!dynamic_cast<ceres_slam::SE3Perturbation *>(lp)) throw std::invalid_argument("ceres_shim: only SE3Perturbation / UnitVectorPerturbation are supported");
        if (!pose_index_.count(block)) throw std::invalid_argument("ceres_shim: parameter block not found");
        parameterized_[block] = 1;
    }
    // poses: per block; shared lighting blocks (light, Phong parameters, textures): checked at Solve, the
    // GPU path holds ALL blocks of one kind constant or none (what the driver's DEBUG lines do)
    // point (position) blocks: all or none (stage 2 of --multistage), lighting problems only
    void SetParameterBlockConstant(double *block) { if (!constant_.count(block)) { constant_[block] = 1; ++version_; } }
    void SetParameterBlockVariable(double *block) { if (constant_.erase(block)) ++version_; }
    void SetParameterLowerBound(double *block, int index, double v) { lower_[block][index] = v; ++version_; }
    void SetParameterUpperBound(double *block, int index, double v) { upper_[block][index] = v; ++version_; }
    int NumResidualBlocks() const { return (int)obs_pose_.size(); }

 private:
    friend void Solve(const Solver::Options &, Problem *, Solver::Summary *);
    friend const char *shim_prepare(Problem &, int *);
    friend class Covariance;
    // back-end handle of the current structure and the contiguous staging tables it points into
    ssba_problem *h_ = nullptr;
    unsigned long version_ = 0, h_version_ = ~0ul;
    std::vector<double> st_poses_, st_points_, st_normals_, st_phong_, st_texture_, st_intensity_, st_normal_obs_;
    std::vector<double *> normal_blocks_, phong_blocks_, texture_blocks_;
    std::vector<uint32_t> st_material_of_point_;
    double st_light_[3] = {0, 0, 0};
    // the stereo + unary-pose part of the lowering, shared by Solve and Covariance::Compute; returns the failing call or NULL
    const char *lower_core_(ssba_problem *h, std::vector<double> &poses, std::vector<double> &points, int *rc) {
        if ((*rc = ssba_add_pose_blocks(h, poses.data(), (uint32_t)pose_blocks_.size()))) return "ssba_add_pose_blocks";
        if ((*rc = ssba_add_point_blocks(h, points.data(), (uint32_t)point_blocks_.size()))) return "ssba_add_point_blocks";
        for (size_t b = 0; b < obs_pose_.size();) {      // runs of equal stiffness: one call each (a single run for the other drivers)
            size_t e = b + 1;
            while (e < obs_pose_.size() && std::memcmp(&obs_stiffness_[9 * e], &obs_stiffness_[9 * b], 9 * sizeof(double)) == 0) ++e;
            if ((*rc = ssba_add_stereo_observations(h, &obs_pose_[b], &obs_point_[b], &obs_uvd_[3 * b], e - b, &obs_stiffness_[9 * b])))
                return "ssba_add_stereo_observations";
            b = e;
        }
        for (auto &kv : constant_)
            if (pose_index_.count(kv.first) && (*rc = ssba_set_pose_constant(h, pose_index_[kv.first], 1))) return "ssba_set_pose_constant";
        size_t const_points = 0;        // position blocks: all constant (stage 2 of --multistage, dataset_ba_phong.cpp:213-220) or none
        for (double *b : point_blocks_) const_points += constant_.count(b);
        if (const_points != 0 && const_points != point_blocks_.size()) throw std::invalid_argument("ceres_shim: hold all point blocks constant or none");
        if (const_points && (*rc = ssba_set_point_blocks_constant(h, 1))) return "ssba_set_point_blocks_constant";
        for (auto &f : pose_factors_) {
            const uint32_t k = pose_index_[f.pose];
            if (f.type == 0) { if ((*rc = ssba_add_pose_prior(h, k, f.data, f.stiffness, f.huber))) return "ssba_add_pose_prior"; }
            else if ((*rc = ssba_add_sun_observation(h, k, f.data, f.data + 3, f.stiffness, f.data[6], f.data[7], f.huber))) return "ssba_add_sun_observation";
        }
        for (auto &f : rel_factors_)
            if ((*rc = ssba_add_relative_pose(h, pose_index_[f.pose1], pose_index_[f.pose2], f.T_ref, f.stiffness, f.huber))) return "ssba_add_relative_pose";
        if (huber_a_ > 0 && (*rc = ssba_set_huber_loss(h, huber_a_))) return "ssba_set_huber_loss";
        return nullptr;
    }
    static uint32_t block_index(std::map<double *, uint32_t> &idx, std::vector<double *> &blocks, double *b) {
        auto it = idx.find(b);
        if (it != idx.end()) return it->second;
        const uint32_t i = (uint32_t)blocks.size();
        idx[b] = i;
        blocks.push_back(b);
        return i;
    }
    std::shared_ptr<const ceres_slam::StereoCamera> camera_;
    std::vector<double> obs_stiffness_;          // 9 per stereo residual block
    double huber_a_ = 0.0;
    struct PoseFactor { int type; double *pose; double data[18], stiffness[36], huber; };
    std::vector<PoseFactor> pose_factors_;
    struct RelFactor { double *pose1, *pose2; double T_ref[12], stiffness[36], huber; };
    std::vector<RelFactor> rel_factors_;
    std::map<double *, uint32_t> pose_index_, point_index_;
    int dist_world_ = 0, dist_rank_ = 0;
    unsigned char dist_id_[SSBA_RCCL_UNIQUE_ID_BYTES];
    std::vector<double *> pose_blocks_, point_blocks_;
    std::vector<uint32_t> obs_pose_, obs_point_;
    std::vector<double> obs_uvd_;
    std::map<double *, int> parameterized_, constant_, unit_vector_;
    struct IntensityBlock { double *pose, *position, *normal, *phong, *texture; double colour, stiffness; int light_type; };
    struct NormalBlock { double *pose, *normal; double obs[3], stiffness[9]; };
    std::vector<IntensityBlock> intensity_;
    std::vector<NormalBlock> normals_;
    double *light_ = nullptr;
    std::map<double *, std::map<int, double>> lower_, upper_;
    std::vector<CostFunction *> owned_costs_;
    std::map<LossFunction *, int> owned_losses_;
    std::map<LocalParameterization *, int> owned_params_;
};

// Gathers the caller's blocks into the problem's contiguous staging tables (the C ABI takes block tables) and makes
// sure a finalized handle for the problem's CURRENT structure exists: built on first use, kept across Solve /
// Covariance::Compute calls, rebuilt after any AddResidualBlock / SetParameterBlock* / bound change.  Returns the
// failing call (status in *rc) or NULL.
inline const char *shim_prepare(Problem &P, int *rc_out) {
    int &rc = *rc_out;
    rc = 0;
    for (double *b : P.pose_blocks_)
        if (!P.parameterized_.count(b)) throw std::invalid_argument("ceres_shim: pose block without SE3Perturbation");
    std::vector<double> &poses = P.st_poses_, &points = P.st_points_;
    poses.resize(P.pose_blocks_.size() * 12);
    points.resize(P.point_blocks_.size() * 3);
    for (size_t i = 0; i < P.pose_blocks_.size(); ++i) std::memcpy(&poses[12 * i], P.pose_blocks_[i], 12 * sizeof(double));
    for (size_t i = 0; i < P.point_blocks_.size(); ++i) std::memcpy(&points[3 * i], P.point_blocks_[i], 3 * sizeof(double));
    std::vector<double> &normals = P.st_normals_, &phong = P.st_phong_, &texture = P.st_texture_, &intensity = P.st_intensity_,
                        &normal_obs = P.st_normal_obs_;
    std::vector<double *> &normal_blocks = P.normal_blocks_, &phong_blocks = P.phong_blocks_, &texture_blocks = P.texture_blocks_;
    std::vector<uint32_t> &material_of_point = P.st_material_of_point_;
    double (&light)[3] = P.st_light_;
    const bool lighting = !P.intensity_.empty();
    if (P.h_ && P.h_version_ == P.version_) {      // same structure: only the values may have moved
        if (lighting) {
            for (size_t j = 0; j < P.point_blocks_.size(); ++j) std::memcpy(&normals[3 * j], normal_blocks[j], 3 * sizeof(double));
            for (size_t m = 0; m < phong_blocks.size(); ++m) { std::memcpy(&phong[3 * m], phong_blocks[m], 3 * sizeof(double)); texture[m] = *texture_blocks[m]; }
            std::memcpy(light, P.light_, sizeof light);
        }
        return nullptr;
    }
    if (P.h_) { ssba_destroy(P.h_); P.h_ = nullptr; }
    ssba_camera cam = {1.0, 1.0, 0.0, 0.0, 1.0};      // pose-graph problems (tests/blowup_test.cpp) have no stereo block
    if (P.camera_) cam = ssba_camera{P.camera_->fu, P.camera_->fv, P.camera_->cu, P.camera_->cv, P.camera_->b};
    ssba_problem *&h = P.h_;
    rc = ssba_create(&cam, -1, &h);
    auto fail = [&](const char *where) -> const char * {
        if (h) { ssba_destroy(h); h = nullptr; }
        return where;
    };
    if (rc) return fail("ssba_create");
    if (P.dist_world_ > 0 && (rc = ssba_set_distributed(h, P.dist_world_, P.dist_rank_))) return fail("ssba_set_distributed");
    if (const char *where = P.lower_core_(h, poses, points, &rc)) return fail(where);
    // ---- lighting terms (tests/dataset_ba_phong.cpp:101-204) -> the config-3 tables of the C ABI ----
    normal_blocks.assign(P.point_blocks_.size(), nullptr);
    phong_blocks.clear();
    texture_blocks.clear();
    material_of_point.assign(P.point_blocks_.size(), 0);
    if (lighting) {
        const size_t N = P.obs_pose_.size();
        if (P.intensity_.size() != N || P.normals_.size() != N)
            throw std::invalid_argument("ceres_shim: every stereo observation needs one intensity and one normal residual block");
        std::map<std::pair<double *, double *>, size_t> obs_of;      // (pose, position) -> stereo observation
        for (size_t i = 0; i < N; ++i) obs_of[{P.pose_blocks_[P.obs_pose_[i]], P.point_blocks_[P.obs_point_[i]]}] = i;
        std::map<double *, uint32_t> material_index, point_of_normal;
        intensity.assign(N, 0.0);
        normal_obs.assign(3 * N, 0.0);
        for (auto &b : P.intensity_) {
            auto it = obs_of.find({b.pose, b.position});
            if (it == obs_of.end()) throw std::invalid_argument("ceres_shim: intensity residual without a stereo residual on the same (pose, point)");
            const uint32_t j = P.point_index_[b.position];
            if (normal_blocks[j] && normal_blocks[j] != b.normal) throw std::invalid_argument("ceres_shim: a vertex has two normal blocks");
            normal_blocks[j] = b.normal;
            point_of_normal[b.normal] = j;
            if (!material_index.count(b.phong)) {
                material_index[b.phong] = (uint32_t)phong_blocks.size();
                phong_blocks.push_back(b.phong);
                texture_blocks.push_back(b.texture);
            }
            const uint32_t m = material_index[b.phong];
            if (texture_blocks[m] != b.texture) throw std::invalid_argument("ceres_shim: Phong parameter and texture blocks must pair one-to-one");
            material_of_point[j] = m;
            intensity[it->second] = b.colour;
        }
        for (auto &b : P.normals_) {
            auto pj = point_of_normal.find(b.normal);
            if (pj == point_of_normal.end()) throw std::invalid_argument("ceres_shim: normal residual on an unknown normal block");
            auto it = obs_of.find({b.pose, P.point_blocks_[pj->second]});
            if (it == obs_of.end()) throw std::invalid_argument("ceres_shim: normal residual without a stereo residual on the same (pose, point)");
            std::memcpy(&normal_obs[3 * it->second], b.obs, 3 * sizeof(double));
            if (std::memcmp(b.stiffness, P.normals_[0].stiffness, sizeof b.stiffness) != 0)
                throw std::invalid_argument("ceres_shim: normal residual blocks must share their stiffness");
        }
        const size_t M = phong_blocks.size();
        normals.resize(3 * P.point_blocks_.size());
        for (size_t j = 0; j < P.point_blocks_.size(); ++j) {
            if (!normal_blocks[j]) throw std::invalid_argument("ceres_shim: a vertex without lighting terms");
            if (!P.unit_vector_.count(normal_blocks[j])) throw std::invalid_argument("ceres_shim: normal block without UnitVectorPerturbation");
            std::memcpy(&normals[3 * j], normal_blocks[j], 3 * sizeof(double));
        }
        phong.resize(3 * M); texture.resize(M);
        for (size_t m = 0; m < M; ++m) { std::memcpy(&phong[3 * m], phong_blocks[m], 3 * sizeof(double)); texture[m] = *texture_blocks[m]; }
        std::memcpy(light, P.light_, sizeof light);
        const int light_type = P.intensity_[0].light_type;
        if (light_type == 1 && !P.unit_vector_.count(P.light_)) throw std::invalid_argument("ceres_shim: light direction without UnitVectorPerturbation");
        if ((rc = ssba_add_normal_blocks(h, normals.data(), (uint32_t)P.point_blocks_.size()))) return fail("ssba_add_normal_blocks");
        if ((rc = ssba_add_material_blocks(h, phong.data(), texture.data(), (uint32_t)M, material_of_point.data(), (uint32_t)P.point_blocks_.size())))
            return fail("ssba_add_material_blocks");
        if ((rc = ssba_add_light_block(h, light, light_type))) return fail("ssba_add_light_block");
        if ((rc = ssba_add_lighting_observations(h, intensity.data(), P.intensity_[0].stiffness, normal_obs.data(), P.normals_[0].stiffness, N)))
            return fail("ssba_add_lighting_observations");
        // SetParameterBlockConstant / bounds on shared blocks: all blocks of a kind or none
        auto all_or_none = [&](const std::vector<double *> &blocks, const char *what) {
            size_t n = 0;
            for (double *b : blocks) n += P.constant_.count(b);
            if (n != 0 && n != blocks.size()) throw std::invalid_argument(std::string("ceres_shim: hold all ") + what + " blocks constant or none");
            return n != 0;
        };
        if ((rc = ssba_set_shared_block_constant(h, SSBA_BLOCK_LIGHT, P.constant_.count(P.light_) ? 1 : 0))) return fail("ssba_set_shared_block_constant");
        if ((rc = ssba_set_shared_block_constant(h, SSBA_BLOCK_PHONG, all_or_none(phong_blocks, "Phong parameter") ? 1 : 0))) return fail("ssba_set_shared_block_constant");
        if ((rc = ssba_set_shared_block_constant(h, SSBA_BLOCK_TEXTURE, all_or_none(texture_blocks, "texture") ? 1 : 0))) return fail("ssba_set_shared_block_constant");
        auto bounds = [&](const std::vector<double *> &blocks, int which, int size) -> int {
            for (int idx = 0; idx < size; ++idx) {
                auto get = [&](std::map<double *, std::map<int, double>> &tab, double *b, double none) {
                    auto it = tab.find(b);
                    if (it == tab.end() || !it->second.count(idx)) return none;
                    return it->second[idx];
                };
                const double inf = 1.0 / 0.0;
                const double lo = get(P.lower_, blocks[0], -inf), hi = get(P.upper_, blocks[0], inf);
                for (double *b : blocks)
                    if (get(P.lower_, b, -inf) != lo || get(P.upper_, b, inf) != hi)
                        throw std::invalid_argument("ceres_shim: bounds must be the same on all blocks of a kind");
                if (lo != -inf || hi != inf)
                    if (int r = ssba_set_shared_block_bounds(h, which, idx, lo, hi)) return r;
            }
            return 0;
        };
        if ((rc = bounds(phong_blocks, SSBA_BLOCK_PHONG, 3))) return fail("ssba_set_shared_block_bounds");
        if ((rc = bounds(texture_blocks, SSBA_BLOCK_TEXTURE, 1))) return fail("ssba_set_shared_block_bounds");
    }
    if ((rc = ssba_finalize(h))) return fail("ssba_finalize");
    if (P.dist_world_ > 0 && (rc = ssba_set_rccl(h, P.dist_id_, sizeof P.dist_id_))) return fail("ssba_set_rccl");
    P.h_version_ = P.version_;
    return nullptr;
}

// ceres::Solve(options, &problem, &summary) (tests/dataset_vo.cpp:81).  Never throws for
// solver outcomes: the result is in `summary` (as with Ceres); API misuse throws.
inline void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary) {
    Problem &P = *problem;
    *summary = Solver::Summary();
    if (P.obs_pose_.empty() && P.pose_factors_.empty() && P.rel_factors_.empty() && P.dist_world_ <= 1) {     // (a shard without blocks still joins the collectives)
        summary->termination_type = CONVERGENCE;
        summary->message = "no residual blocks";
        return;
    }
    int rc = 0;
    auto fail = [&](const char *where) {
        summary->termination_type = FAILURE;
        summary->message = std::string(where) + ": " + ssba_status_string(rc) + " (" + ssba_last_error() + ")";
    };
    if (const char *where = shim_prepare(P, &rc)) return fail(where);
    ssba_problem *h = P.h_;
    std::vector<double> &poses = P.st_poses_, &points = P.st_points_, &normals = P.st_normals_, &phong = P.st_phong_, &texture = P.st_texture_;
    std::vector<double *> &normal_blocks = P.normal_blocks_, &phong_blocks = P.phong_blocks_, &texture_blocks = P.texture_blocks_;
    double (&light)[3] = P.st_light_;
    const bool lighting = !P.intensity_.empty();
    ssba_options o;
    ssba_default_options(&o);
    o.max_num_iterations = options.max_num_iterations;
    o.use_nonmonotonic_steps = options.use_nonmonotonic_steps ? 1 : 0;
    o.minimizer_progress_to_stdout = options.minimizer_progress_to_stdout ? 1 : 0;
    o.num_threads = options.num_threads;
    o.num_linear_solver_threads = options.num_linear_solver_threads;
    o.function_tolerance = options.function_tolerance;
    o.gradient_tolerance = options.gradient_tolerance;
    o.parameter_tolerance = options.parameter_tolerance;
    o.initial_trust_region_radius = options.initial_trust_region_radius;
    o.trust_region_strategy_type = options.trust_region_strategy_type == DOGLEG ? 1 : 0;
    o.dogleg_type = options.dogleg_type == SUBSPACE_DOGLEG ? 1 : 0;
    ssba_summary s;
    rc = ssba_solve(h, &o, &s);
    if (rc && rc != SSBA_ERR_NUMERICAL_FAILURE) return fail("ssba_solve");
    summary->termination_type = (TerminationType)s.termination_type;
    summary->num_successful_steps = s.num_successful_steps;
    summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
    summary->num_line_search_steps = s.num_line_search_steps;
    summary->initial_cost = s.initial_cost;
    summary->final_cost = s.final_cost;
    summary->total_time_in_seconds = s.total_time_s;
    if (summary->IsSolutionUsable()) {
        for (size_t i = 0; i < P.pose_blocks_.size(); ++i) std::memcpy(P.pose_blocks_[i], &poses[12 * i], 12 * sizeof(double));
        for (size_t i = 0; i < P.point_blocks_.size(); ++i) std::memcpy(P.point_blocks_[i], &points[3 * i], 3 * sizeof(double));
        if (lighting) {   // every lighting block is written back; constant ones come back unchanged
            for (size_t j = 0; j < P.point_blocks_.size(); ++j) std::memcpy(normal_blocks[j], &normals[3 * j], 3 * sizeof(double));
            for (size_t m = 0; m < phong_blocks.size(); ++m) { std::memcpy(phong_blocks[m], &phong[3 * m], 3 * sizeof(double)); *texture_blocks[m] = texture[m]; }
            std::memcpy(P.light_, light, sizeof light);
        }
    }
}

// ceres::Covariance for pose blocks (tests/dataset_vo_sun.cpp:159-183): Compute evaluates the requested diagonal pose
// blocks of (J^T J)^-1 in the tangent space at the problem's current parameter values (loss-corrected Jacobian, as
// Ceres' default apply_loss_function = true); GetCovarianceBlockInTangentSpace copies one out (6x6 row-major).
class Covariance {
 public:
    struct Options {
        int num_threads = 1;
        CovarianceAlgorithmType algorithm_type = SPARSE_QR;
        SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type = SUITE_SPARSE;
        int null_space_rank = 0;
        bool apply_loss_function = true;
    };
    explicit Covariance(const Options &options) : options_(options) {}
    bool Compute(const std::vector<std::pair<const double *, const double *>> &blocks, Problem *problem) {
        Problem &P = *problem;
        blocks_.clear();
        message_.clear();
        if (!P.intensity_.empty() || !P.normals_.empty()) { message_ = "covariance is not available with lighting terms"; return false; }
        if (P.pose_blocks_.empty()) { message_ = "no pose blocks"; return false; }
        int rc = 0;
        const char *where = shim_prepare(P, &rc);      // the handle of the preceding Solve when nothing changed since
        for (size_t i = 0; !where && i < blocks.size(); ++i) {
            double *a = const_cast<double *>(blocks[i].first);
            if (blocks[i].first != blocks[i].second || !P.pose_index_.count(a)) {
                message_ = "only diagonal pose blocks are supported";
                rc = SSBA_ERR_UNSUPPORTED;
                where = "Covariance::Compute";
                break;
            }
            Block b;
            b.ptr = blocks[i].first;
            if ((rc = ssba_pose_covariance(P.h_, P.pose_index_[a], b.cov))) where = "ssba_pose_covariance";
            else blocks_.push_back(b);
        }
        if (where && message_.empty()) message_ = std::string(where) + ": " + ssba_status_string(rc) + " (" + ssba_last_error() + ")";
        if (where) blocks_.clear();
        return where == nullptr;     // like Ceres: false on a rank-deficient Jacobian (no gauge constraint)
    }
    bool GetCovarianceBlockInTangentSpace(const double *a, const double *b, double *out) const {
        if (a != b) return false;
        for (const Block &blk : blocks_)
            if (blk.ptr == a) { std::memcpy(out, blk.cov, sizeof blk.cov); return true; }
        return false;
    }
    const std::string &message() const { return message_; }

 private:
    struct Block { const double *ptr; double cov[36]; };
    Options options_;
    std::vector<Block> blocks_;
    std::string message_;
};

}  // namespace ceres
#endif

"""Host-side mirror of the reference's operator interface for the stereo-BA path.

The reference has no FFI: its plug-in point is the Ceres C++ API as used by
/root/reference tests/dataset_vo.cpp:22-85.  This module reproduces those call
shapes in Python -- same names, argument meaning and error behaviour -- on top of
the C ABI (include/ssba.h), so that the parity tests read like the reference's
driver:

    problem = Problem()
    se3 = SE3Perturbation.Create()
    cost = StereoReprojectionErrorAutomatic.Create(camera, obs, stiffness)
    problem.AddResidualBlock(cost, None, poses[k], points[j])
    problem.SetParameterization(poses[k], se3)
    problem.SetParameterBlockConstant(poses[k1])
    Solve(options, problem, summary); print(summary.BriefReport())

A parameter block is a numpy row view (12 doubles ``[t | R row-major]`` for a
pose, 3 for a point); block identity is the address of its first element, exactly
as Ceres keys blocks by ``double*``.  Blocks are updated in place by ``Solve``.

and the sun-aided driver's surface (tests/dataset_vo_sun.cpp:28-185): per-block stereo stiffness,
``PoseErrorAutomatic`` / ``SunSensorErrorAutomatic`` on one pose block, ``RelativePoseErrorAutomatic`` on two
(tests/blowup_test.cpp), ``Covariance.Compute`` / ``GetCovarianceBlockInTangentSpace`` for diagonal pose blocks.

The C++ twin of this file is include/ceres_slam_amd/ceres_shim.hpp.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
_TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}


@dataclass(frozen=True)
class StereoCamera:
    """include/ceres_slam/stereo_camera.hpp:159-163"""
    fu: float
    fv: float
    cu: float
    cv: float
    b: float

    def as_c(self) -> capi.Camera:
        return capi.Camera(self.fu, self.fv, self.cu, self.cv, self.b)


class CostFunction:
    pass


class StereoReprojectionErrorAutomatic(CostFunction):
    """include/ceres_slam/stereo_reprojection_error.hpp:12-81 (3 residuals; blocks 12, 3)."""

    def __init__(self, camera: StereoCamera, observation, stiffness):
        self.camera = camera
        self.observation = np.asarray(observation, dtype=np.float64).reshape(3)
        self.stiffness = np.asarray(stiffness, dtype=np.float64).reshape(3, 3)

    @staticmethod
    def Create(camera, observation, stiffness) -> "StereoReprojectionErrorAutomatic":
        return StereoReprojectionErrorAutomatic(camera, observation, stiffness)


class PoseErrorAutomatic(CostFunction):
    """include/ceres_slam/pose_error.hpp:11-74: prior r = S log(T_ref T^-1) on one pose block (6 residuals)."""

    def __init__(self, T_k_0_ref, stiffness):
        self.T_ref = np.asarray(T_k_0_ref, dtype=np.float64).reshape(12).copy()
        self.stiffness = np.asarray(stiffness, dtype=np.float64).reshape(6, 6).copy()

    @staticmethod
    def Create(T_k_0_ref, stiffness) -> "PoseErrorAutomatic":
        return PoseErrorAutomatic(T_k_0_ref, stiffness)


class SunSensorErrorAutomatic(CostFunction):
    """include/ceres_slam/sun_sensor_error.hpp:12-131: azimuth / zenith error of the expected sun direction (2 residuals)."""

    def __init__(self, observed_sun_dir_c, expected_sun_dir_g, stiffness, az_err_thresh, zen_err_thresh):
        self.observed = np.asarray(observed_sun_dir_c, dtype=np.float64).reshape(3).copy()
        self.expected = np.asarray(expected_sun_dir_g, dtype=np.float64).reshape(3).copy()
        self.stiffness = np.asarray(stiffness, dtype=np.float64).reshape(2, 2).copy()
        self.az_err_thresh, self.zen_err_thresh = float(az_err_thresh), float(zen_err_thresh)

    @staticmethod
    def Create(observed_sun_dir_c, expected_sun_dir_g, stiffness, az_err_thresh, zen_err_thresh) -> "SunSensorErrorAutomatic":
        return SunSensorErrorAutomatic(observed_sun_dir_c, expected_sun_dir_g, stiffness, az_err_thresh, zen_err_thresh)


class RelativePoseErrorAutomatic(CostFunction):
    """include/ceres_slam/relative_pose_error.hpp:11-67: r = S log(T_2_1_ref T_1 T_2^-1) on two pose blocks."""

    def __init__(self, T_2_1_ref, stiffness):
        self.T_ref = np.asarray(T_2_1_ref, dtype=np.float64).reshape(12).copy()
        self.stiffness = np.asarray(stiffness, dtype=np.float64).reshape(6, 6).copy()

    @staticmethod
    def Create(T_2_1_ref, stiffness) -> "RelativePoseErrorAutomatic":
        return RelativePoseErrorAutomatic(T_2_1_ref, stiffness)


class LocalParameterization:
    pass


class SE3Perturbation(LocalParameterization):
    """include/ceres_slam/perturbations.hpp:45-76: Plus(T, eps) = exp(eps) * T, 12 -> 6."""

    @staticmethod
    def Create() -> "SE3Perturbation":
        return SE3Perturbation()


class HuberLoss:
    """ceres::HuberLoss(a) (call-site shape: tests/dataset_vo_sun.cpp:89-95)."""

    def __init__(self, a: float):
        if not a > 0:
            raise ValueError("HuberLoss scale must be positive")
        self.a = float(a)


class SolverOptions:
    """The ceres::Solver::Options fields the drivers touch (tests/dataset_vo.cpp:65-74),
    with Ceres 1.x defaults."""

    def __init__(self):
        o = capi.default_options()
        for name, _ in capi.Options._fields_:
            setattr(self, name, getattr(o, name))

    def as_c(self) -> capi.Options:
        o = capi.Options()
        for name, _ in capi.Options._fields_:
            setattr(o, name, getattr(self, name))
        return o


class SolverSummary:
    def __init__(self):
        self.termination_type = NO_CONVERGENCE
        self.num_successful_steps = 0
        self.num_unsuccessful_steps = 0
        self.num_line_search_steps = 0
        self.initial_cost = 0.0
        self.final_cost = 0.0
        self.total_time_in_seconds = 0.0
        self.device_time_in_seconds = 0.0
        self.iterations = []

    def IsSolutionUsable(self) -> bool:
        return self.termination_type in (CONVERGENCE, NO_CONVERGENCE)

    def BriefReport(self) -> str:
        return ("Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s"
                % (self.num_successful_steps + self.num_unsuccessful_steps, self.initial_cost,
                   self.final_cost, _TERMINATION.get(self.termination_type, "UNKNOWN")))


def _addr(block: np.ndarray) -> int:
    if not isinstance(block, np.ndarray) or block.dtype != np.float64 or not block.flags.c_contiguous:
        raise TypeError("a parameter block must be a C-contiguous float64 numpy view")
    return block.ctypes.data


class Problem:
    """ceres::Problem as used by the stereo drivers (tests/dataset_vo.cpp:26-62)."""

    def __init__(self):
        self._pose_blocks = {}      # address -> (index, view)
        self._point_blocks = {}
        self._obs_pose, self._obs_point, self._obs_uvd = [], [], []
        self._bulk = []             # (pose_idx array, point_idx array, uvd array)
        self._camera = None
        self._stiffness = None
        self._loss = "unset"
        self._parameterized = set()
        self._constant = set()
        self._obs_stiffness = []    # per residual block (tests/dataset_vo_sun.cpp:56-65 gives every map point its own)
        self._pose_factors = []     # dicts as for solver.StereoBA(pose_factors=...): prior / sun / relative pose

    # -- graph building -----------------------------------------------------
    def _block(self, table, block, size):
        a = _addr(block)
        if block.size != size:
            raise ValueError(f"parameter block has {block.size} doubles, expected {size}")
        if a not in table:
            table[a] = (len(table), block)
        return table[a][0]

    def _register(self, cost, loss):
        if not isinstance(cost, StereoReprojectionErrorAutomatic):
            # the GPU path executes typed observation tables, not arbitrary functors
            raise TypeError(f"unsupported cost function {type(cost).__name__}: this back end accelerates "
                            "StereoReprojectionErrorAutomatic residual blocks")
        if self._camera is None:
            self._camera, self._stiffness = cost.camera, cost.stiffness
        elif cost.camera != self._camera:
            raise ValueError("all residual blocks must share one camera")
        if loss is not None and not isinstance(loss, HuberLoss):
            raise TypeError("loss must be None or HuberLoss")
        key = None if loss is None else loss.a
        if self._loss == "unset":
            self._loss = key
        elif self._loss != key:
            raise ValueError("all residual blocks must share the same loss function")

    def AddResidualBlock(self, cost, loss, pose_block, point_block=None):
        if loss is not None and not isinstance(loss, HuberLoss):
            raise TypeError("loss must be None or HuberLoss")
        huber = 0.0 if loss is None else loss.a
        if isinstance(cost, (PoseErrorAutomatic, SunSensorErrorAutomatic)):       # one pose block (dataset_vo_sun.cpp:80-119)
            if point_block is not None:
                raise TypeError(f"{type(cost).__name__} takes one parameter block")
            k = self._block(self._pose_blocks, pose_block, 12)
            if isinstance(cost, PoseErrorAutomatic):
                self._pose_factors.append(dict(pose=k, type=0, data=cost.T_ref, stiffness=cost.stiffness.ravel(), huber=huber))
            else:
                data = np.concatenate([cost.observed, cost.expected, [cost.az_err_thresh, cost.zen_err_thresh]])
                self._pose_factors.append(dict(pose=k, type=1, data=data, stiffness=cost.stiffness.ravel(), huber=huber))
            return
        if isinstance(cost, RelativePoseErrorAutomatic):                           # two pose blocks (blowup_test.cpp:70-76)
            k1, k2 = self._block(self._pose_blocks, pose_block, 12), self._block(self._pose_blocks, point_block, 12)
            self._pose_factors.append(dict(pose=k1, pose2=k2, type=2, data=cost.T_ref, stiffness=cost.stiffness.ravel(), huber=huber))
            return
        self._register(cost, loss)
        self._obs_pose.append(self._block(self._pose_blocks, pose_block, 12))
        self._obs_point.append(self._block(self._point_blocks, point_block, 3))
        self._obs_uvd.append(cost.observation)
        self._obs_stiffness.append(cost.stiffness)

    def AddStereoResidualBlocks(self, camera, stiffness, loss, poses, points, pose_index, point_index, uvd):
        """Vectorised form of the driver's double loop (tests/dataset_vo.cpp:39-56): one
        StereoReprojectionErrorAutomatic block per row of (pose_index, point_index, uvd)."""
        self._register(StereoReprojectionErrorAutomatic(camera, np.zeros(3), stiffness), loss)
        pi = np.asarray(pose_index, dtype=np.int64)
        li = np.asarray(point_index, dtype=np.int64)
        pmap = np.array([self._block(self._pose_blocks, poses[k], 12) for k in range(poses.shape[0])], dtype=np.uint32)
        lmap = np.array([self._block(self._point_blocks, points[j], 3) for j in range(points.shape[0])], dtype=np.uint32)
        self._bulk.append((pmap[pi], lmap[li], np.ascontiguousarray(uvd, dtype=np.float64)))

    def SetParameterization(self, block, parameterization):
        if not isinstance(parameterization, SE3Perturbation):
            raise TypeError("only SE3Perturbation is supported on pose blocks")
        a = _addr(block)
        if a not in self._pose_blocks:
            raise KeyError("parameter block not found in the problem")   # Ceres aborts here
        self._parameterized.add(a)

    def SetParameterBlockConstant(self, block):
        a = _addr(block)
        if a not in self._pose_blocks:
            raise KeyError("parameter block not found (only pose blocks can be held constant)")
        self._constant.add(a)

    def SetParameterBlockVariable(self, block):
        self._constant.discard(_addr(block))

    def NumResidualBlocks(self) -> int:
        return len(self._obs_pose) + sum(b[0].shape[0] for b in self._bulk)

    def NumParameterBlocks(self) -> int:
        return len(self._pose_blocks) + len(self._point_blocks)


def _lower(problem: Problem, device: int = -1):
    """Gathers the caller's blocks into contiguous tables and builds the back-end handle (solver.StereoBA)."""
    from .solver import StereoBA
    P, L = len(problem._pose_blocks), len(problem._point_blocks)
    missing = [a for a in problem._pose_blocks if a not in problem._parameterized]
    if missing:
        raise ValueError("every pose block needs SetParameterization(block, SE3Perturbation): the 12-double "
                         "block is over-parameterised")
    poses = np.empty((P, 12))
    for a, (i, view) in problem._pose_blocks.items():
        poses[i] = view.reshape(12)
    points = np.empty((L, 3))
    for a, (j, view) in problem._point_blocks.items():
        points[j] = view.reshape(3)
    op = np.concatenate([np.asarray(problem._obs_pose, dtype=np.uint32)] + [b[0] for b in problem._bulk]).astype(np.uint32)
    ol = np.concatenate([np.asarray(problem._obs_point, dtype=np.uint32)] + [b[1] for b in problem._bulk]).astype(np.uint32)
    uv = np.ascontiguousarray(np.concatenate([np.asarray(problem._obs_uvd, dtype=np.float64).reshape(-1, 3)] + [b[2] for b in problem._bulk]))
    shared = problem._stiffness if problem._stiffness is not None else np.eye(3)
    per_block = [np.asarray(x) for x in problem._obs_stiffness] + [shared] * sum(b[0].shape[0] for b in problem._bulk)
    stiffness = shared if all(np.array_equal(x, shared) for x in per_block) else np.array(per_block)
    cam = problem._camera or StereoCamera(1.0, 1.0, 0.0, 0.0, 1.0)
    const = np.zeros(P, dtype=np.uint8)
    for a in problem._constant:
        const[problem._pose_blocks[a][0]] = 1
    ba = StereoBA(dict(fu=cam.fu, fv=cam.fv, cu=cam.cu, cv=cam.cv, b=cam.b), poses, points, op, ol, uv, stiffness, pose_const=const,
                  huber_a=0.0 if problem._loss in ("unset", None) else problem._loss, device=device,
                  pose_factors=problem._pose_factors or None)
    return ba


def _scatter(problem: Problem, ba):
    for a, (i, view) in problem._pose_blocks.items():
        view.reshape(12)[:] = ba.poses[i]
    for a, (j, view) in problem._point_blocks.items():
        view.reshape(3)[:] = ba.points[j]


def Solve(options: SolverOptions, problem: Problem, summary: SolverSummary, device: int = -1):
    """ceres::Solve(options, &problem, &summary) (tests/dataset_vo.cpp:81)."""
    ba = _lower(problem, device)
    s, log = ba.solve(options.as_c())
    summary.termination_type = s.termination_type
    summary.num_successful_steps = s.num_successful_steps
    summary.num_unsuccessful_steps = s.num_unsuccessful_steps
    summary.num_line_search_steps = s.num_line_search_steps
    summary.initial_cost, summary.final_cost = s.initial_cost, s.final_cost
    summary.total_time_in_seconds, summary.device_time_in_seconds = s.total_time_s, s.device_time_s
    names = ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius")
    n = len(log["cost"])
    summary.iterations = [dict({k: float(log[k][i]) for k in names}, iteration=i, step_is_successful=bool(log["step_is_successful"][i]))
                          for i in range(n)]
    if summary.IsSolutionUsable():      # parameters are user-owned and updated in place, only when the solution is usable
        _scatter(problem, ba)
    ba.close()
    return summary


class Covariance:
    """ceres::Covariance for diagonal pose blocks (tests/dataset_vo_sun.cpp:159-183): Compute evaluates the requested
    blocks of (J^T J)^-1 in the tangent space at the problem's current values; False on a rank-deficient system."""

    class Options:
        num_threads = 1

    def __init__(self, options=None):
        self._blocks = {}
        self.message = ""

    def Compute(self, covariance_blocks, problem: Problem, device: int = -1) -> bool:
        self._blocks = {}
        ba = _lower(problem, device)
        try:
            for a, b in covariance_blocks:
                if _addr(a) != _addr(b) or _addr(a) not in problem._pose_blocks:
                    self.message = "only diagonal pose blocks are supported"
                    return False
                self._blocks[_addr(a)] = ba.pose_covariance(problem._pose_blocks[_addr(a)][0])
        except capi.SsbaError as e:
            self.message = str(e)
            self._blocks = {}
            return False
        finally:
            ba.close()
        return True

    def GetCovarianceBlockInTangentSpace(self, a, b, out) -> bool:
        if _addr(a) != _addr(b) or _addr(a) not in self._blocks:
            return False
        np.asarray(out).reshape(6, 6)[:] = self._blocks[_addr(a)]
        return True

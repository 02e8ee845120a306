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
        elif cost.camera != self._camera or not np.array_equal(cost.stiffness, self._stiffness):
            raise ValueError("all residual blocks must share one camera and one stiffness matrix")
        if loss is not None and not isinstance(loss, HuberLoss):
            raise TypeError("loss must be None or HuberLoss")
        key = None if loss is None else loss.a
        if self._loss == "unset":
            self._loss = key
        elif self._loss != key:
            raise ValueError("all residual blocks must share the same loss function")

    def AddResidualBlock(self, cost, loss, pose_block, point_block):
        self._register(cost, loss)
        self._obs_pose.append(self._block(self._pose_blocks, pose_block, 12))
        self._obs_point.append(self._block(self._point_blocks, point_block, 3))
        self._obs_uvd.append(cost.observation)

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


def Solve(options: SolverOptions, problem: Problem, summary: SolverSummary, device: int = -1):
    """ceres::Solve(options, &problem, &summary) (tests/dataset_vo.cpp:81)."""
    lib = capi.load()
    P, L = len(problem._pose_blocks), len(problem._point_blocks)
    missing = [a for a in problem._pose_blocks if a not in problem._parameterized]
    if missing:
        raise ValueError("every pose block needs SetParameterization(block, SE3Perturbation): the 12-double "
                         "block is over-parameterised")
    # gather caller blocks into the contiguous tables the C ABI takes; scattered back below
    poses = np.empty((P, 12))
    for a, (i, view) in problem._pose_blocks.items():
        poses[i] = view.reshape(12)
    points = np.empty((L, 3))
    for a, (j, view) in problem._point_blocks.items():
        points[j] = view.reshape(3)
    op = np.concatenate([np.asarray(problem._obs_pose, dtype=np.uint32)] + [b[0] for b in problem._bulk]).astype(np.uint32)
    ol = np.concatenate([np.asarray(problem._obs_point, dtype=np.uint32)] + [b[1] for b in problem._bulk]).astype(np.uint32)
    uv = np.concatenate([np.asarray(problem._obs_uvd, dtype=np.float64).reshape(-1, 3)] + [b[2] for b in problem._bulk])
    uv = np.ascontiguousarray(uv)
    cam = problem._camera.as_c() if problem._camera else capi.Camera(1, 1, 0, 0, 1)
    S = np.ascontiguousarray(problem._stiffness if problem._stiffness is not None else np.eye(3)).reshape(9)

    h = C.c_void_p()
    capi.check(lib.ssba_create(C.byref(cam), device, C.byref(h)), "ssba_create")
    try:
        capi.check(lib.ssba_add_pose_blocks(h, capi.dptr(poses), P), "ssba_add_pose_blocks")
        capi.check(lib.ssba_add_point_blocks(h, capi.dptr(points), L), "ssba_add_point_blocks")
        capi.check(lib.ssba_add_stereo_observations(
            h, op.ctypes.data_as(capi._u32p), ol.ctypes.data_as(capi._u32p), capi.dptr(uv), op.shape[0],
            capi.dptr(S)), "ssba_add_stereo_observations")
        for a in problem._constant:
            capi.check(lib.ssba_set_pose_constant(h, problem._pose_blocks[a][0], 1), "ssba_set_pose_constant")
        if problem._loss not in ("unset", None):
            capi.check(lib.ssba_set_huber_loss(h, problem._loss), "ssba_set_huber_loss")
        capi.check(lib.ssba_finalize(h), "ssba_finalize")
        s = capi.Summary()
        o = options.as_c()
        rc = lib.ssba_solve(h, C.byref(o), C.byref(s))
        if rc not in (capi.SSBA_OK, -3):
            capi.check(rc, "ssba_solve")
        n = lib.ssba_iteration_log(h, 0, None, None, None, None, None, None, None)
        cols = [np.zeros(n) for _ in range(6)]
        ok = np.zeros(n, dtype=np.int32)
        lib.ssba_iteration_log(h, n, *[capi.dptr(c) for c in cols], ok.ctypes.data_as(capi._i32p))
    finally:
        lib.ssba_destroy(h)
    summary.termination_type = s.termination_type
    summary.num_successful_steps = s.num_successful_steps
    summary.num_unsuccessful_steps = s.num_unsuccessful_steps
    summary.initial_cost, summary.final_cost = s.initial_cost, s.final_cost
    summary.total_time_in_seconds, summary.device_time_in_seconds = s.total_time_s, s.device_time_s
    names = ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius")
    summary.iterations = [dict({k: float(c[i]) for k, c in zip(names, cols)}, iteration=i,
                               step_is_successful=bool(ok[i])) for i in range(n)]
    # parameters are user-owned and updated in place, only when the solution is usable
    if summary.IsSolutionUsable():
        for a, (i, view) in problem._pose_blocks.items():
            view.reshape(12)[:] = poses[i]
        for a, (j, view) in problem._point_blocks.items():
            view.reshape(3)[:] = points[j]
    return summary

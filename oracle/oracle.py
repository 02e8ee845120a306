"""ctypes binding of the CPU oracle (oracle/ssba_oracle.c).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  PARITY UNPINNED at the Ceres boundary
(see oracle/ssba_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libssba_oracle.so")

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fu", "fv", "cu", "cv", "b")]


class Problem(C.Structure):
    _fields_ = [("cam", Camera), ("num_poses", C.c_int32), ("num_points", C.c_int32),
                ("num_obs", C.c_int64), ("poses", _dp), ("points", _dp),
                ("obs_pose", _u32p), ("obs_point", _u32p), ("obs_uvd", _dp),
                ("stiffness", C.c_double * 9), ("pose_const", _u8p), ("huber_a", C.c_double),
                ("normals", _dp), ("intensity", _dp), ("normal_obs", _dp), ("phong", _dp), ("texture", _dp),
                ("material_of_point", _u32p), ("light", C.c_double * 3), ("light_type", C.c_int32),
                ("shared_free", C.c_uint32), ("int_stiffness", C.c_double), ("normal_stiffness", C.c_double * 9),
                ("num_materials", C.c_uint32), ("use_bounds", C.c_uint32),
                ("num_pose_factors", C.c_uint32), ("reserved3", C.c_uint32), ("pf_pose", _u32p), ("pf_type", _u32p),
                ("pf_data", _dp), ("pf_stiffness", _dp), ("pf_huber", _dp), ("obs_stiffness", _dp),
                ("positions_constant", C.c_uint32), ("reserved4", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("use_nonmonotonic_steps", C.c_int32),
                ("max_consecutive_nonmonotonic_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
                ("num_threads", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("trust_region_strategy_type", C.c_int32),
                ("dogleg_type", C.c_int32)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("total_time_s", C.c_double), ("linearize_time_s", C.c_double),
                ("schur_time_s", C.c_double), ("solve_time_s", C.c_double),
                ("update_time_s", C.c_double), ("num_line_search_steps", C.c_int32), ("reserved", C.c_int32)]


class IterationLog(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("cost", _dp), ("cost_change", _dp),
                ("gradient_max_norm", _dp), ("step_norm", _dp), ("relative_decrease", _dp),
                ("trust_region_radius", _dp), ("step_is_successful", _i32p)]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed recipe (oracle/Makefile)."""
    src = os.path.join(_HERE, "ssba_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_se3_transform.argtypes = [_dp, _dp, _dp]
        L.orc_so3_exp.argtypes = [_dp, _dp]
        L.orc_se3_plus.argtypes = [_dp, _dp, _dp]
        L.orc_se3_inverse.argtypes = [_dp, _dp]
        L.orc_project.argtypes = [C.POINTER(Camera), _dp, _dp, _dp]
        L.orc_triangulate.argtypes = [C.POINTER(Camera), _dp, _dp, _dp]
        L.orc_stereo_residual.argtypes = [C.POINTER(Camera), _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_stereo_residual_autodiff.argtypes = [C.POINTER(Camera), _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_set_jacobian_mode.argtypes = [C.c_int]
        L.orc_huber.argtypes = [C.c_double, C.c_double, _dp]
        L.orc_default_options.argtypes = [C.POINTER(Options)]
        L.orc_cost.argtypes = [C.POINTER(Problem), C.c_int]
        L.orc_cost.restype = C.c_double
        L.orc_linearize.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, _dp, C.c_int]
        L.orc_linearize.restype = C.c_double
        L.orc_reduced_system.argtypes = [C.POINTER(Problem), C.c_double, C.POINTER(Options), _dp, _dp, _i32p]
        L.orc_lm_step.argtypes = [C.POINTER(Problem), C.c_double, C.POINTER(Options), _dp, _dp, _dp]
        L.orc_lm_step_border.argtypes = [C.POINTER(Problem), C.c_double, C.POINTER(Options), _dp, _dp, _dp, _dp]
        L.orc_border_size.argtypes = [C.POINTER(Problem)]
        L.orc_solve.argtypes = [C.POINTER(Problem), C.POINTER(Options), C.POINTER(Summary), C.POINTER(IterationLog)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def default_options(**kw) -> Options:
    o = Options()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def driver_options(**kw) -> Options:
    """The options the reference drivers set (tests/dataset_vo.cpp:65-70)."""
    base = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, num_threads=8)
    base.update(kw)
    return default_options(**base)


def expand_relative_pose_factors(pose_factors, pose_const):
    """A RelativePoseErrorAutomatic block (type 2: pose, pose2, data = T_2_1_ref, 6x6 stiffness) becomes two half
    entries, one per pose block (types 2 / 3: data[12] the other pose, data[13] = 1 on the half that counts the
    cost -- the first non-constant one --, data[14] the index of the other half)."""
    out = []
    for f in pose_factors:
        if f["type"] != 2 or "pose2" not in f:
            out.append(f)
            continue
        k1, k2 = int(f["pose"]), int(f["pose2"])
        first_counts = not (pose_const[k1] and not pose_const[k2])
        ia, ib = len(out), len(out) + 1
        T_ref = np.asarray(f["data"], dtype=np.float64).ravel()[:12]
        for typ, k, other, counts, partner in ((2, k1, k2, first_counts, ib), (3, k2, k1, not first_counts, ia)):
            out.append(dict(pose=k, type=typ, data=np.concatenate([T_ref, [float(other), 1.0 if counts else 0.0, float(partner)]]),
                            stiffness=f["stiffness"], huber=f.get("huber", 0.0)))
    return out


class OracleProblem:
    """Owns numpy copies of a problem and the ctypes view the C oracle reads."""

    def __init__(self, camera: dict, poses, points, obs_pose, obs_point, obs_uvd, stiffness,
                 pose_const=None, huber_a: float = 0.0, lighting=None, shared_free: int = 0, use_bounds: bool = False,
                 pose_factors=None, positions_const: bool = False):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).copy()
        self.points = np.ascontiguousarray(points, dtype=np.float64).copy()
        self.obs_pose = np.ascontiguousarray(obs_pose, dtype=np.uint32)
        self.obs_point = np.ascontiguousarray(obs_point, dtype=np.uint32)
        self.obs_uvd = np.ascontiguousarray(obs_uvd, dtype=np.float64)
        P = self.poses.shape[0]
        if pose_const is None:
            pose_const = np.zeros(P, dtype=np.uint8)
            if P:
                pose_const[0] = 1          # tests/dataset_vo.cpp:62
        self.pose_const = np.ascontiguousarray(pose_const, dtype=np.uint8)
        self.c = Problem()
        self.c.cam = Camera(**camera)
        self.c.num_poses, self.c.num_points = P, self.points.shape[0]
        self.c.num_obs = self.obs_pose.shape[0]
        self.c.poses, self.c.points = _p(self.poses), _p(self.points)
        self.c.obs_pose = self.obs_pose.ctypes.data_as(_u32p)
        self.c.obs_point = self.obs_point.ctypes.data_as(_u32p)
        self.c.obs_uvd = _p(self.obs_uvd)
        S = np.asarray(stiffness, dtype=np.float64)
        if S.ndim == 3:     # one 3x3 stiffness per residual block (tests/dataset_vo_sun.cpp:56-65)
            self._obs_S = np.ascontiguousarray(S.reshape(-1, 9))
            assert self._obs_S.shape[0] == self.obs_pose.shape[0]
            self.c.obs_stiffness = _p(self._obs_S)
            S = S[0] if S.shape[0] else np.eye(3)
        self.c.stiffness = (C.c_double * 9)(*S.reshape(9))
        self.c.pose_const = self.pose_const.ctypes.data_as(_u8p)
        self.c.huber_a = float(huber_a)
        self.c.positions_constant = 1 if positions_const else 0     # stage 2 of --multistage (lighting problems only)
        self.ld = 3
        self.normals = None
        if pose_factors:        # list of dicts: pose, type (0 prior / 1 sun), data (<= 18), stiffness (36 or 4), huber
            pose_factors = expand_relative_pose_factors(pose_factors, self.pose_const)
            F = len(pose_factors)
            self._pf_pose = np.array([f["pose"] for f in pose_factors], dtype=np.uint32)
            self._pf_type = np.array([f["type"] for f in pose_factors], dtype=np.uint32)
            self._pf_data, self._pf_S, self._pf_h = np.zeros((F, 18)), np.zeros((F, 36)), np.zeros(F)
            for i, f in enumerate(pose_factors):
                dat, S = np.asarray(f["data"], dtype=np.float64).ravel(), np.asarray(f["stiffness"], dtype=np.float64).ravel()
                self._pf_data[i, :len(dat)] = dat
                self._pf_S[i, :len(S)] = S
                self._pf_h[i] = float(f.get("huber", 0.0))
            self.c.num_pose_factors = F
            self.c.pf_pose, self.c.pf_type = self._pf_pose.ctypes.data_as(_u32p), self._pf_type.ctypes.data_as(_u32p)
            self.c.pf_data, self.c.pf_stiffness, self.c.pf_huber = _p(self._pf_data), _p(self._pf_S), _p(self._pf_h)
        if lighting is not None:       # dict: normals, intensity, normal_obs, phong, texture, material_of_point, light, light_type, int_stiffness, normal_stiffness
            self.ld = 6
            self.normals = np.ascontiguousarray(lighting["normals"], dtype=np.float64).copy()
            self._lt = {k: np.ascontiguousarray(lighting[k], dtype=np.float64) for k in ("intensity", "normal_obs")}
            # shared blocks: updated in place when freed (bit 0 light, bit 1 Phong parameters, bit 2 textures)
            self.phong = np.ascontiguousarray(lighting["phong"], dtype=np.float64).copy()
            self.texture = np.ascontiguousarray(lighting["texture"], dtype=np.float64).copy()
            self._lt["phong"], self._lt["texture"] = self.phong, self.texture
            self.c.num_materials = self.texture.shape[0]
            self.c.shared_free = int(shared_free)
            self.c.use_bounds = 1 if use_bounds else 0
            self._mat = np.ascontiguousarray(lighting["material_of_point"], dtype=np.uint32)
            self.c.normals = _p(self.normals)
            self.c.intensity, self.c.normal_obs = _p(self._lt["intensity"]), _p(self._lt["normal_obs"])
            self.c.phong, self.c.texture = _p(self._lt["phong"]), _p(self._lt["texture"])
            self.c.material_of_point = self._mat.ctypes.data_as(_u32p)
            self.c.light = (C.c_double * 3)(*np.asarray(lighting["light"], dtype=np.float64))
            self.c.light_type = int(lighting["light_type"])
            self.c.int_stiffness = float(lighting["int_stiffness"])
            self.c.normal_stiffness = (C.c_double * 9)(*np.asarray(lighting["normal_stiffness"], dtype=np.float64).reshape(9))

    @classmethod
    def from_synth(cls, prob, huber_a: float = 0.0, pose_const=None):
        return cls(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point,
                   prob.obs_uvd, prob.stiffness(), pose_const=pose_const, huber_a=huber_a)

    # ---- evaluation hooks -------------------------------------------------
    def cost(self, num_threads: int = 1) -> float:
        return lib().orc_cost(C.byref(self.c), num_threads)

    def linearize(self, num_threads: int = 1):
        P, L, ld = self.c.num_poses, self.c.num_points, self.ld
        g_p, g_l = np.zeros((P, 6)), np.zeros((L, ld))
        H_pp, H_ll = np.zeros((P, 6, 6)), np.zeros((L, ld, ld))
        cost = lib().orc_linearize(C.byref(self.c), _p(g_p), _p(g_l), _p(H_pp), _p(H_ll), num_threads)
        return cost, g_p, g_l, H_pp, H_ll

    def reduced_system(self, radius: float, options: Options = None):
        o = options or default_options()
        P = self.c.num_poses
        nb = self.border_size()
        nmax = 6 * P + nb
        S, rhs = np.zeros((nmax, nmax)), np.zeros(nmax)
        free_idx = np.zeros(P, dtype=np.int32)
        rc = lib().orc_reduced_system(C.byref(self.c), radius, C.byref(o), _p(S), _p(rhs),
                                      free_idx.ctypes.data_as(_i32p))
        if rc:
            raise RuntimeError("orc_reduced_system failed")
        n = 6 * int((free_idx >= 0).sum()) + nb     # with free shared blocks: the (poses + border) arrowhead
        return S.reshape(-1)[: n * n].reshape(n, n).copy(), rhs[:n].copy(), free_idx

    @property
    def light(self) -> np.ndarray:
        return np.array(list(self.c.light))

    def border_size(self) -> int:
        return int(lib().orc_border_size(C.byref(self.c))) if self.ld == 6 else 0

    def lm_step(self, radius: float, options: Options = None, want_border: bool = False):
        o = options or default_options()
        dp, dl = np.zeros((self.c.num_poses, 6)), np.zeros((self.c.num_points, self.ld))
        db = np.zeros(max(self.border_size(), 1))
        mcc = C.c_double(0.0)
        rc = lib().orc_lm_step_border(C.byref(self.c), radius, C.byref(o), _p(dp), _p(dl), _p(db), C.byref(mcc))
        if rc:
            raise RuntimeError("orc_lm_step failed")
        if want_border:
            return dp, dl, db[: self.border_size()], mcc.value
        return dp, dl, mcc.value

    def solve(self, options: Options = None, log_capacity: int = 1024):
        o = options or driver_options()
        s = Summary()
        arrs = {k: np.zeros(log_capacity) for k in ("cost", "cost_change", "gradient_max_norm",
                                                     "step_norm", "relative_decrease", "trust_region_radius")}
        ok = np.zeros(log_capacity, dtype=np.int32)
        lg = IterationLog(log_capacity, *[_p(arrs[k]) for k in ("cost", "cost_change", "gradient_max_norm",
                                                                 "step_norm", "relative_decrease",
                                                                 "trust_region_radius")],
                          ok.ctypes.data_as(_i32p))
        lib().orc_solve(C.byref(self.c), C.byref(o), C.byref(s), C.byref(lg))
        n = min(s.num_iterations, log_capacity)
        log = {k: v[:n].copy() for k, v in arrs.items()}
        log["step_is_successful"] = ok[:n].copy()
        return s, log


# ---- thin functional wrappers over the L1/L2 restatements ---------------------
def se3_transform(T, p):
    out = np.zeros(3)
    lib().orc_se3_transform(_p(np.ascontiguousarray(T, dtype=np.float64)), _p(np.ascontiguousarray(p, dtype=np.float64)), _p(out))
    return out


def so3_exp(phi):
    out = np.zeros(9)
    lib().orc_so3_exp(_p(np.ascontiguousarray(phi, dtype=np.float64)), _p(out))
    return out.reshape(3, 3)


def se3_plus(T, eps):
    out = np.zeros(12)
    lib().orc_se3_plus(_p(np.ascontiguousarray(T, dtype=np.float64)), _p(np.ascontiguousarray(eps, dtype=np.float64)), _p(out))
    return out


def se3_inverse(T):
    out = np.zeros(12)
    lib().orc_se3_inverse(_p(np.ascontiguousarray(T, dtype=np.float64)), _p(out))
    return out


def project(camera: dict, q, jac: bool = False):
    cam = Camera(**camera)
    out, J = np.zeros(3), np.zeros(9)
    lib().orc_project(C.byref(cam), _p(np.ascontiguousarray(q, dtype=np.float64)), _p(out), _p(J) if jac else None)
    return (out, J.reshape(3, 3)) if jac else out


def triangulate(camera: dict, uvd, jac: bool = False):
    cam = Camera(**camera)
    out, J = np.zeros(3), np.zeros(9)
    lib().orc_triangulate(C.byref(cam), _p(np.ascontiguousarray(uvd, dtype=np.float64)), _p(out), _p(J) if jac else None)
    return (out, J.reshape(3, 3)) if jac else out


def stereo_residual(camera: dict, T, p, z, S, jac: bool = False, autodiff: bool = False):
    """autodiff=True: the Jet restatement (what AutoDiffCostFunction + AutoDiffLocalParameterization evaluate)."""
    cam = Camera(**camera)
    r, Jp, Jl = np.zeros(3), np.zeros(18), np.zeros(9)
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (T, p, z, np.asarray(S).reshape(9))]
    fn = lib().orc_stereo_residual_autodiff if autodiff else lib().orc_stereo_residual
    fn(C.byref(cam), *[_p(a) for a in args], _p(r), _p(Jp) if jac else None, _p(Jl) if jac else None)
    return (r, Jp.reshape(3, 6), Jl.reshape(3, 3)) if jac else r


def set_jacobian_mode(mode: int):
    """0: closed-form Jacobians (default); 1: every stereo block through the Jet restatement (CPU-baseline timing variant)."""
    lib().orc_set_jacobian_mode(int(mode))


def huber(a: float, s: float):
    rho = np.zeros(3)
    lib().orc_huber(a, s, _p(rho))
    return rho


# ---- Phong lighting rows -----------------------------------------
POINT_LIGHT, DIRECTIONAL_LIGHT = 0, 1


def _phong_setup():
    L = lib()
    if getattr(L, "_phong_ready", False):
        return L
    L.orc_phong_shade.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double]
    L.orc_phong_shade.restype = C.c_double
    L.orc_light_shade.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double]
    L.orc_light_shade.restype = C.c_double
    L.orc_unit_vector_plus.argtypes = [_dp, _dp, _dp]
    L.orc_intensity_residual.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp, C.c_double, C.c_double, _dp, _dp]
    L.orc_normal_residual.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L._phong_ready = True
    return L


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def light_shade(light_type, p, n, light, kd, ks, alpha) -> float:
    p, n, light = _a(p), _a(n), _a(light)
    return _phong_setup().orc_light_shade(light_type, _p(p), _p(n), _p(light), kd, ks, alpha)


def phong_shade(n, ldir, cdir, kd, ks, alpha) -> float:
    n, ldir, cdir = _a(n), _a(ldir), _a(cdir)
    return _phong_setup().orc_phong_shade(_p(n), _p(ldir), _p(cdir), kd, ks, alpha)


def unit_vector_plus(x, delta):
    x, delta, out = _a(x), _a(delta), np.zeros(3)
    _phong_setup().orc_unit_vector_plus(_p(x), _p(delta), _p(out))
    return out


def intensity_residual(light_type, T, p, n, phong, kd, light, colour, stiffness, jac=False):
    T, p, n, phong, light = _a(T), _a(p), _a(n), _a(phong), _a(light)
    r, J = C.c_double(), np.zeros(19)
    _phong_setup().orc_intensity_residual(light_type, _p(T), _p(p), _p(n), _p(phong), kd, _p(light), colour,
                                          stiffness, C.byref(r), _p(J) if jac else None)
    return (r.value, J) if jac else r.value


def normal_residual(T, n, n_obs, S, jac=False):
    T, n, n_obs, S = _a(T), _a(n), _a(n_obs), _a(np.asarray(S).reshape(9))
    r, Jp, Jn = np.zeros(3), np.zeros(18), np.zeros(9)
    _phong_setup().orc_normal_residual(_p(T), _p(n), _p(n_obs), _p(S), _p(r), _p(Jp) if jac else None,
                                       _p(Jn) if jac else None)
    return (r, Jp.reshape(3, 6), Jn.reshape(3, 3)) if jac else r

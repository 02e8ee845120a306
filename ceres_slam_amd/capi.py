"""ctypes binding of the C ABI in include/ssba.h (ceres_slam_amd/libssba.so).

This is plumbing only: every entry point is the C symbol of the same name.  The
library is HIP-only; if it is missing, cannot be loaded, or finds no GPU, calls
raise -- there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssba.so")

MAX_TRACK = 12

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)

SSBA_OK = 0
STATUS = {0: "SSBA_OK", -1: "SSBA_ERR_INVALID_ARGUMENT", -2: "SSBA_ERR_HIP", -3: "SSBA_ERR_NUMERICAL_FAILURE",
          -4: "SSBA_ERR_NOT_FINALIZED", -5: "SSBA_ERR_NO_DEVICE", -6: "SSBA_ERR_UNSUPPORTED", -7: "SSBA_ERR_STATE", -8: "SSBA_ERR_TIMEOUT"}

#: every symbol include/ssba.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "ssba_create", "ssba_destroy", "ssba_add_pose_blocks", "ssba_add_point_blocks",
    "ssba_add_stereo_observations", "ssba_set_pose_constant", "ssba_set_huber_loss", "ssba_finalize",
    "ssba_default_options", "ssba_solve", "ssba_brief_report", "ssba_solve_begin", "ssba_solve_step",
    "ssba_solve_end", "ssba_solve_restart", "ssba_synchronize", "ssba_iteration_log", "ssba_set_stream",
    "ssba_set_exchange", "ssba_set_distributed", "ssba_exchange_size", "ssba_set_kernel_timing", "ssba_kernel_times",
    "ssba_get_stats", "ssba_evaluate", "ssba_lm_step", "ssba_phong_evaluate", "ssba_status_string", "ssba_last_error",
    "ssba_add_normal_blocks", "ssba_add_material_blocks", "ssba_add_light_block", "ssba_set_shared_block_constant",
    "ssba_add_lighting_observations", "ssba_border_system", "ssba_set_shared_block_bounds", "ssba_set_point_blocks_constant", "ssba_release_cached_memory",
    "ssba_set_partition", "ssba_ransac_samples", "ssba_frontend_ransac", "ssba_add_pose_prior", "ssba_add_sun_observation", "ssba_add_relative_pose",
    "ssba_pose_covariance", "ssba_rccl_unique_id", "ssba_set_rccl", "ssba_frontend_vo",
    "ssba_rccl_describe", "ssba_rccl_ranks", "ssba_armijo_trace", "ssba_debug_stamps",
]


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fu", "fv", "cu", "cv", "b")]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("use_nonmonotonic_steps", C.c_int32),
                ("max_consecutive_nonmonotonic_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
                ("max_num_consecutive_invalid_steps", C.c_int32), ("minimizer_progress_to_stdout", C.c_int32),
                ("num_threads", C.c_int32), ("num_linear_solver_threads", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("trust_region_strategy_type", C.c_int32),
                ("dogleg_type", C.c_int32)]


LEVENBERG_MARQUARDT, DOGLEG = 0, 1


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("total_time_s", C.c_double), ("device_time_s", C.c_double),
                ("num_line_search_steps", C.c_int32), ("num_line_searches_on_device", C.c_int32),
                ("num_line_searches_by_host", C.c_int32), ("reserved_", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("num_poses", C.c_uint32), ("num_free_poses", C.c_uint32), ("num_points", C.c_uint32),
                ("num_active_points", C.c_uint32), ("num_observations", C.c_uint64),
                ("num_windows", C.c_uint32), ("num_superblocks", C.c_uint32),
                ("num_reduced_blocks", C.c_uint32), ("pose_bandwidth", C.c_uint32),
                ("device_bytes", C.c_uint64), ("general_structure", C.c_uint32), ("pcr_blocks", C.c_uint32),
                ("pcr_fused", C.c_uint32), ("wide_superblocks", C.c_uint32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)


class SsbaError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {STATUS.get(status, status)}" + (f" ({detail})" if detail else ""))


_lib = None


def load():
    """Load libssba.so; raises if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m ceres_slam_amd.build` "
                          "(the stereo-BA path is HIP-only, there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.ssba_create.argtypes = [C.POINTER(Camera), C.c_int, C.POINTER(H)]
    L.ssba_destroy.argtypes = [H]
    L.ssba_add_pose_blocks.argtypes = [H, _dp, C.c_uint32]
    L.ssba_add_point_blocks.argtypes = [H, _dp, C.c_uint32]
    L.ssba_add_stereo_observations.argtypes = [H, _u32p, _u32p, _dp, C.c_uint64, _dp]
    L.ssba_set_pose_constant.argtypes = [H, C.c_uint32, C.c_int]
    L.ssba_set_huber_loss.argtypes = [H, C.c_double]
    L.ssba_finalize.argtypes = [H]
    L.ssba_default_options.argtypes = [C.POINTER(Options)]
    L.ssba_default_options.restype = None
    L.ssba_solve.argtypes = [H, C.POINTER(Options), C.POINTER(Summary)]
    L.ssba_brief_report.argtypes = [C.POINTER(Summary), C.c_char_p, C.c_size_t]
    L.ssba_solve_begin.argtypes = [H, C.POINTER(Options), C.c_int]
    L.ssba_solve_step.argtypes = [H, C.c_int]
    L.ssba_solve_end.argtypes = [H, C.POINTER(Summary)]
    L.ssba_solve_restart.argtypes = [H]
    L.ssba_synchronize.argtypes = [H]
    L.ssba_iteration_log.argtypes = [H, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _i32p]
    L.ssba_set_stream.argtypes = [H, C.c_void_p]
    L.ssba_set_exchange.argtypes = [H, EXCHANGE_FN, C.c_void_p]
    L.ssba_set_distributed.argtypes = [H, C.c_int, C.c_int]
    L.ssba_set_partition.argtypes = [H, _u32p, C.c_uint32]
    L.ssba_frontend_vo.argtypes = [C.POINTER(Camera), C.c_int, C.c_uint32, _u32p, _u32p, _dp, C.c_uint32, C.c_uint32, C.c_double, C.c_int,
                                   _dp, _dp, C.POINTER(C.c_uint8), _u32p, _u32p, _dp]
    L.ssba_rccl_unique_id.argtypes = [C.c_void_p, C.c_uint64]
    L.ssba_set_rccl.argtypes = [H, C.c_void_p, C.c_uint64]
    L.ssba_rccl_describe.argtypes = [C.c_char_p, C.c_uint64]
    L.ssba_rccl_ranks.argtypes = [H, C.POINTER(C.c_int)]
    L.ssba_exchange_size.argtypes = [H, C.POINTER(C.c_uint64)]
    L.ssba_set_kernel_timing.argtypes = [H, C.c_int]
    L.ssba_kernel_times.argtypes = [H, C.POINTER(KernelTime), C.c_int32, _i32p]
    L.ssba_get_stats.argtypes = [H, C.POINTER(Stats)]
    L.ssba_evaluate.argtypes = [H, _dp, _dp, _dp, _dp, _dp]
    L.ssba_lm_step.argtypes = [H, C.POINTER(Options), C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.ssba_armijo_trace.argtypes = [_dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_double, _dp, _dp, C.c_int32, C.c_int32]
    L.ssba_phong_evaluate.argtypes = [C.c_int, C.c_int, C.c_uint64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp,
                                      _dp, _dp, _dp, _dp, _dp]
    L.ssba_add_normal_blocks.argtypes = [H, _dp, C.c_uint32]
    L.ssba_add_material_blocks.argtypes = [H, _dp, _dp, C.c_uint32, _u32p, C.c_uint32]
    L.ssba_add_light_block.argtypes = [H, _dp, C.c_int]
    L.ssba_set_shared_block_constant.argtypes = [H, C.c_int, C.c_int]
    L.ssba_set_shared_block_bounds.argtypes = [H, C.c_int, C.c_int, C.c_double, C.c_double]
    L.ssba_set_point_blocks_constant.argtypes = [H, C.c_int]
    L.ssba_border_system.argtypes = [H, C.POINTER(C.c_uint32), _dp, _dp, _dp, _dp]
    L.ssba_add_lighting_observations.argtypes = [H, _dp, C.c_double, _dp, _dp, C.c_uint64]
    L.ssba_add_pose_prior.argtypes = [H, C.c_uint32, _dp, _dp, C.c_double]
    L.ssba_add_sun_observation.argtypes = [H, C.c_uint32, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double]
    L.ssba_add_relative_pose.argtypes = [H, C.c_uint32, C.c_uint32, _dp, _dp, C.c_double]
    L.ssba_pose_covariance.argtypes = [H, C.c_uint32, _dp]
    L.ssba_ransac_samples.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _u32p]
    L.ssba_frontend_ransac.argtypes = [C.POINTER(Camera), C.c_int, C.c_uint32, _u32p, _dp, _dp, _u32p, C.c_uint32, C.c_double, _dp,
                                       C.POINTER(C.c_uint8), _u32p, _dp]
    L.ssba_status_string.argtypes = [C.c_int]
    L.ssba_status_string.restype = C.c_char_p
    L.ssba_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(status: int, where: str):
    if status != SSBA_OK:
        detail = load().ssba_last_error().decode(errors="replace")
        raise SsbaError(status, where, detail)


def default_options(**kw) -> Options:
    o = Options()
    load().ssba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def dptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def phong_evaluate(light_type, poses, points, normals, phong, texture, light, colour, stiffness, normal_obs,
                   normal_stiffness, device: int = -1):
    """Batch evaluation of the intensity + normal residual blocks on the GPU (ssba_phong_evaluate)."""
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (poses, points, normals, phong, texture, light, colour,
                                                              normal_obs, np.asarray(normal_stiffness).reshape(9))]
    n = a[0].shape[0]
    r_int, J_int = np.zeros(n), np.zeros((n, 19))
    r_nrm, J_np, J_nn = np.zeros((n, 3)), np.zeros((n, 3, 6)), np.zeros((n, 3, 3))
    check(load().ssba_phong_evaluate(device, light_type, n, dptr(a[0]), dptr(a[1]), dptr(a[2]), dptr(a[3]), dptr(a[4]),
                                     dptr(a[5]), dptr(a[6]), float(stiffness), dptr(a[7]), dptr(a[8]), dptr(r_int),
                                     dptr(J_int), dptr(r_nrm), dptr(J_np), dptr(J_nn)), "ssba_phong_evaluate")
    return r_int, J_int, r_nrm, J_np, J_nn

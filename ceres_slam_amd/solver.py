"""Array-level handle over the C ABI (include/ssba.h) for the bench harness, the parity
tests and multi-GPU drivers: one ``StereoBA`` = one ``ssba_problem``.

Everything here is a direct call into libssba.so; numpy is used for host buffers only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class StereoBA:
    def __init__(self, camera: dict, poses: np.ndarray, points: np.ndarray, obs_pose, obs_point, obs_uvd,
                 stiffness, pose_const=None, huber_a: float = 0.0, device: int = -1, finalize: bool = True,
                 world_size: int = 1, rank: int = 0, lighting: dict = None, shared_free: int = 0, use_bounds: bool = False, points_const: bool = False,
                 partition=None, pose_factors=None):
        self.lib = capi.load()
        self.poses = np.ascontiguousarray(poses, dtype=np.float64)     # caller-owned blocks, updated in place
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        self._obs_pose = np.ascontiguousarray(obs_pose, dtype=np.uint32)
        self._obs_point = np.ascontiguousarray(obs_point, dtype=np.uint32)
        self._obs_uvd = np.ascontiguousarray(obs_uvd, dtype=np.float64)
        S = np.asarray(stiffness, dtype=np.float64)
        self._S_obs = None
        if S.ndim == 3:      # one 3x3 stiffness per residual block (tests/dataset_vo_sun.cpp:56-65)
            self._S_obs = np.ascontiguousarray(S.reshape(-1, 9))
            assert self._S_obs.shape[0] == self._obs_pose.shape[0]
        else:
            self._S = np.ascontiguousarray(S.reshape(9))
        self.h = C.c_void_p()
        cam = capi.Camera(**camera)
        capi.check(self.lib.ssba_create(C.byref(cam), device, C.byref(self.h)), "ssba_create")
        P, L = self.poses.shape[0], self.points.shape[0]
        capi.check(self.lib.ssba_add_pose_blocks(self.h, capi.dptr(self.poses), P), "ssba_add_pose_blocks")
        capi.check(self.lib.ssba_add_point_blocks(self.h, capi.dptr(self.points), L), "ssba_add_point_blocks")
        if self._S_obs is None:
            capi.check(self.lib.ssba_add_stereo_observations(
                self.h, self._obs_pose.ctypes.data_as(capi._u32p), self._obs_point.ctypes.data_as(capi._u32p),
                capi.dptr(self._obs_uvd), self._obs_pose.shape[0], capi.dptr(self._S)), "ssba_add_stereo_observations")
        else:                # runs of equal stiffness go in one call each, as the C++ shim lowers them
            n = self._obs_pose.shape[0]
            cut = np.flatnonzero(np.any(self._S_obs[1:] != self._S_obs[:-1], axis=1)) + 1
            for b, e in zip(np.concatenate([[0], cut]), np.concatenate([cut, [n]])):
                b, e = int(b), int(e)
                capi.check(self.lib.ssba_add_stereo_observations(
                    self.h, self._obs_pose[b:e].ctypes.data_as(capi._u32p), self._obs_point[b:e].ctypes.data_as(capi._u32p),
                    capi.dptr(self._obs_uvd[b:e]), e - b, capi.dptr(self._S_obs[b])), "ssba_add_stereo_observations")
        if pose_const is None:
            pose_const = np.zeros(P, dtype=bool)
            if P:
                pose_const[0] = True              # tests/dataset_vo.cpp:62
        for k in np.nonzero(np.asarray(pose_const))[0]:
            capi.check(self.lib.ssba_set_pose_constant(self.h, int(k), 1), "ssba_set_pose_constant")
        if huber_a > 0:
            capi.check(self.lib.ssba_set_huber_loss(self.h, float(huber_a)), "ssba_set_huber_loss")
        for f in (pose_factors or []):      # dicts as for oracle.OracleProblem: pose, type (0 prior / 1 sun), data, stiffness, huber
            dat = np.ascontiguousarray(f["data"], dtype=np.float64).ravel()
            S = np.ascontiguousarray(f["stiffness"], dtype=np.float64).ravel()
            if f["type"] == 2:      # RelativePoseErrorAutomatic between pose and pose2 (tests/blowup_test.cpp:70-76)
                capi.check(self.lib.ssba_add_relative_pose(self.h, int(f["pose"]), int(f["pose2"]), capi.dptr(np.ascontiguousarray(dat[:12])),
                                                           capi.dptr(S), float(f.get("huber", 0.0))), "ssba_add_relative_pose")
            elif f["type"] == 0:
                capi.check(self.lib.ssba_add_pose_prior(self.h, int(f["pose"]), capi.dptr(dat), capi.dptr(S), float(f.get("huber", 0.0))),
                           "ssba_add_pose_prior")
            else:
                obs, exp = np.ascontiguousarray(dat[:3]), np.ascontiguousarray(dat[3:6])
                capi.check(self.lib.ssba_add_sun_observation(self.h, int(f["pose"]), capi.dptr(obs), capi.dptr(exp), capi.dptr(S),
                                                             float(dat[6]), float(dat[7]), float(f.get("huber", 0.0))),
                           "ssba_add_sun_observation")
        self._xcb = None
        self.normals = None
        self.shared_free = int(shared_free)
        self.use_bounds = bool(use_bounds)
        if lighting is not None:
            self._add_lighting(lighting)
        if points_const:      # stage 2 of --multistage (tests/dataset_ba_phong.cpp:210-228)
            capi.check(self.lib.ssba_set_point_blocks_constant(self.h, 1), "ssba_set_point_blocks_constant")
        if world_size > 1:
            capi.check(self.lib.ssba_set_distributed(self.h, world_size, rank), "ssba_set_distributed")
            if partition is not None:     # separator super-blocks of a super-block-aligned sharding (sharding.aligned_partition)
                sep = np.ascontiguousarray(partition, dtype=np.uint32)
                capi.check(self.lib.ssba_set_partition(self.h, sep.ctypes.data_as(capi._u32p), sep.shape[0]), "ssba_set_partition")
        if finalize:
            self.finalize()

    def _add_lighting(self, lt: dict):
        """Config-3 terms (include/ssba.h "config 3"); `lt` has the keys of synth.PhongData.as_oracle_dict()."""
        L, N = self.points.shape[0], self._obs_pose.shape[0]
        self.normals = np.ascontiguousarray(lt["normals"], dtype=np.float64).copy()   # caller-owned block, updated in place
        # shared blocks: caller-owned, updated in place when free (shared_free: bit 0 light, 1 Phong, 2 texture)
        self.phong = np.ascontiguousarray(lt["phong"], dtype=np.float64).copy()
        self.texture = np.ascontiguousarray(lt["texture"], dtype=np.float64).copy()
        self.light = np.ascontiguousarray(lt["light"], dtype=np.float64).copy()
        self._mat = np.ascontiguousarray(lt["material_of_point"], dtype=np.uint32)
        self._int = np.ascontiguousarray(lt["intensity"], dtype=np.float64)
        self._nobs = np.ascontiguousarray(lt["normal_obs"], dtype=np.float64)
        self._Sn = np.ascontiguousarray(np.asarray(lt["normal_stiffness"], dtype=np.float64).reshape(9))
        assert self.normals.shape == (L, 3) and self._int.shape == (N,) and self._nobs.shape == (N, 3)
        capi.check(self.lib.ssba_add_normal_blocks(self.h, capi.dptr(self.normals), L), "ssba_add_normal_blocks")
        capi.check(self.lib.ssba_add_material_blocks(self.h, capi.dptr(self.phong), capi.dptr(self.texture), self.texture.shape[0],
                                                     self._mat.ctypes.data_as(capi._u32p), L), "ssba_add_material_blocks")
        capi.check(self.lib.ssba_add_light_block(self.h, capi.dptr(self.light), int(lt["light_type"])), "ssba_add_light_block")
        for which in range(3):
            capi.check(self.lib.ssba_set_shared_block_constant(self.h, which, 0 if (self.shared_free >> which) & 1 else 1),
                       "ssba_set_shared_block_constant")
        if self.use_bounds:      # tests/dataset_ba_phong.cpp:142-180
            inf = float("inf")
            for which, index, lo, hi in ((1, 0, 0.0, 1.0), (1, 1, 0.0, 1.0), (1, 2, 1.0, inf), (2, 0, 0.0, 1.0)):
                capi.check(self.lib.ssba_set_shared_block_bounds(self.h, which, index, lo, hi), "ssba_set_shared_block_bounds")
        capi.check(self.lib.ssba_add_lighting_observations(self.h, capi.dptr(self._int), float(lt["int_stiffness"]),
                                                           capi.dptr(self._nobs), capi.dptr(self._Sn), N),
                   "ssba_add_lighting_observations")

    @classmethod
    def from_synth(cls, prob, **kw):
        return cls(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point,
                   prob.obs_uvd, prob.stiffness(), **kw)

    def finalize(self):
        capi.check(self.lib.ssba_finalize(self.h), "ssba_finalize")

    def close(self):
        if self.h:
            self.lib.ssba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- solving ----------------------------------------------------------------
    def solve(self, options: capi.Options = None):
        o = options or capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1)
        s = capi.Summary()
        rc = self.lib.ssba_solve(self.h, C.byref(o), C.byref(s))
        if rc not in (0, -3):
            capi.check(rc, "ssba_solve")
        return s, self.iteration_log()

    def solve_begin(self, options: capi.Options, ignore_convergence: bool = False):
        self._opts = options
        capi.check(self.lib.ssba_solve_begin(self.h, C.byref(options), int(ignore_convergence)), "ssba_solve_begin")

    def step(self, n: int = 1):
        capi.check(self.lib.ssba_solve_step(self.h, n), "ssba_solve_step")

    def restart(self):
        capi.check(self.lib.ssba_solve_restart(self.h), "ssba_solve_restart")

    def synchronize(self):
        capi.check(self.lib.ssba_synchronize(self.h), "ssba_synchronize")

    def solve_end(self):
        s = capi.Summary()
        rc = self.lib.ssba_solve_end(self.h, C.byref(s))
        if rc not in (0, -3):
            capi.check(rc, "ssba_solve_end")
        return s

    def iteration_log(self):
        n = self.lib.ssba_iteration_log(self.h, 0, None, None, None, None, None, None, None)
        names = ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius")
        cols = {k: np.zeros(n) for k in names}
        ok = np.zeros(n, dtype=np.int32)
        self.lib.ssba_iteration_log(self.h, n, *[capi.dptr(cols[k]) for k in names], ok.ctypes.data_as(capi._i32p))
        cols["step_is_successful"] = ok
        return cols

    @staticmethod
    def brief_report(s: capi.Summary) -> str:
        buf = C.create_string_buffer(256)
        capi.load().ssba_brief_report(C.byref(s), buf, 256)
        return buf.value.decode()

    # ---- streams / exchange / instrumentation --------------------------------------
    def set_stream(self, stream_ptr: int):
        capi.check(self.lib.ssba_set_stream(self.h, C.c_void_p(stream_ptr)), "ssba_set_stream")

    def set_exchange(self, fn):
        """fn(device_ptr: int, count: int, op: int) -> None, op 0 = sum, 1 = max."""
        if fn is None:
            self._xcb = None
            capi.check(self.lib.ssba_set_exchange(self.h, capi.EXCHANGE_FN(0), None), "ssba_set_exchange")
            return

        def tramp(ctx, ptr, count, op):
            try:
                fn(int(ptr), int(count), int(op))
                return 0
            except Exception:            # never let an exception cross the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._xcb = capi.EXCHANGE_FN(tramp)
        capi.check(self.lib.ssba_set_exchange(self.h, self._xcb, None), "ssba_set_exchange")

    @staticmethod
    def rccl_unique_id() -> bytes:
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
        buf = C.create_string_buffer(128)
        capi.check(capi.load().ssba_rccl_unique_id(buf, 128), "ssba_rccl_unique_id")
        return buf.raw

    def set_rccl(self, unique_id: bytes):
        """Native exchange: the library enqueues ncclAllReduce itself on its stream (collective call, every rank)."""
        self._xcb = None
        buf = C.create_string_buffer(bytes(unique_id), 128)
        capi.check(self.lib.ssba_set_rccl(self.h, buf, 128), "ssba_set_rccl")

    @staticmethod
    def rccl_describe() -> str:
        """Which librccl.so file this process uses, its version, the IPC / debug environment (ssba_rccl_describe)."""
        buf = C.create_string_buffer(1024)
        capi.check(capi.load().ssba_rccl_describe(buf, 1024), "ssba_rccl_describe")
        return buf.value.decode(errors="replace")

    def rccl_ranks(self) -> int:
        """ncclCommCount of the handle's communicator; 0 when the exchange is not the native one."""
        n = C.c_int()
        capi.check(self.lib.ssba_rccl_ranks(self.h, C.byref(n)), "ssba_rccl_ranks")
        return n.value

    def exchange_size(self) -> int:
        n = C.c_uint64()
        capi.check(self.lib.ssba_exchange_size(self.h, C.byref(n)), "ssba_exchange_size")
        return n.value

    def set_kernel_timing(self, on: bool):
        capi.check(self.lib.ssba_set_kernel_timing(self.h, int(on)), "ssba_set_kernel_timing")

    def kernel_times(self) -> dict:
        rows = (capi.KernelTime * 32)()
        n = C.c_int32()
        capi.check(self.lib.ssba_kernel_times(self.h, rows, 32, C.byref(n)), "ssba_kernel_times")
        return {rows[i].name.decode(): (int(rows[i].launches), float(rows[i].total_ms)) for i in range(n.value)}

    def stats(self) -> capi.Stats:
        st = capi.Stats()
        capi.check(self.lib.ssba_get_stats(self.h, C.byref(st)), "ssba_get_stats")
        return st

    # ---- test hooks -------------------------------------------------------------------
    def evaluate(self):
        P, L = self.poses.shape[0], self.points.shape[0]
        cost = C.c_double()
        ld = 6 if self.normals is not None else 3
        g_p, g_l = np.zeros((P, 6)), np.zeros((L, ld))
        H_pp, H_ll = np.zeros((P, 6, 6)), np.zeros((L, ld, ld))
        capi.check(self.lib.ssba_evaluate(self.h, C.byref(cost), capi.dptr(g_p), capi.dptr(g_l), capi.dptr(H_pp),
                                          capi.dptr(H_ll)), "ssba_evaluate")
        return cost.value, g_p, g_l, H_pp, H_ll

    def pose_covariance(self, pose: int) -> np.ndarray:
        """6x6 block of (J^T J)^-1 of one pose in tangent space (ceres::Covariance, tests/dataset_vo_sun.cpp:159-183)."""
        cov = np.zeros((6, 6))
        capi.check(self.lib.ssba_pose_covariance(self.h, int(pose), capi.dptr(cov)), "ssba_pose_covariance")
        return cov

    def border_system(self):
        """Border blocks of the last lm_step (free shared blocks): S_pb, S_bb (damped), rhs_b, delta_b."""
        nb = C.c_uint32()
        capi.check(self.lib.ssba_border_system(self.h, C.byref(nb), None, None, None, None), "ssba_border_system")
        nb = int(nb.value)
        n = 6 * self.stats().num_free_poses
        S_pb, S_bb, rhs_b, db = np.zeros((n, max(nb, 1))), np.zeros((max(nb, 1), max(nb, 1))), np.zeros(max(nb, 1)), np.zeros(max(nb, 1))
        if nb:
            capi.check(self.lib.ssba_border_system(self.h, None, capi.dptr(S_pb), capi.dptr(S_bb), capi.dptr(rhs_b), capi.dptr(db)),
                       "ssba_border_system")
        return S_pb[:, :nb], S_bb[:nb, :nb], rhs_b[:nb], db[:nb]

    def lm_step(self, radius: float, options: capi.Options = None, want_S: bool = True):
        P, L = self.poses.shape[0], self.points.shape[0]
        n = 6 * self.stats().num_free_poses
        S = np.zeros((n, n)) if want_S else None
        rhs = np.zeros(n)
        dp, dl = np.zeros((P, 6)), np.zeros((L, 6 if self.normals is not None else 3))
        mcc = C.c_double()
        o = options or capi.default_options()
        capi.check(self.lib.ssba_lm_step(self.h, C.byref(o), radius, capi.dptr(S) if want_S else None, capi.dptr(rhs),
                                         capi.dptr(dp), capi.dptr(dl), C.byref(mcc)), "ssba_lm_step")
        return S, rhs, dp, dl, mcc.value

"""Deterministic synthetic stereo bundle-adjustment problems.

The reference ships no data and no generator (its drivers read external CSVs:
/root/reference scripts/ba_all_sims.sh:3-5), so the inputs of every test and of
bench.py come from here.  Conventions follow the reference exactly:

* camera intrinsics are the KITTI values of tests/camera_test.cpp:11-15;
* a pose block is 12 doubles ``[t(3) | R row-major(9)]`` holding ``T_c_g``
  (camera-from-global; include/ceres_slam/geometry/se3group.hpp:425-429);
* an observation is ``(u_l, v_l, disparity)`` (stereo_camera.hpp:77-84);
* observations are ordered by state ``k`` then landmark ``j``, the file order
  the reference readers assume (src/ceres_slam/dataset_problem.cpp:89-98);
* the first pose is exact and held constant by the drivers
  (tests/dataset_vo.cpp:62).

Everything is float64 / uint32 numpy; the RNG is ``numpy.random.PCG64(seed)``
(seed 42 is the reference's own RANSAC seed, point_cloud_aligner.cpp:72).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# tests/camera_test.cpp:11-15
KITTI_CAMERA = dict(fu=707.0912, fv=707.0912, cu=601.8873, cv=183.1104, b=0.535105804)
IMAGE_W, IMAGE_H = 1242.0, 375.0

#: named BASELINE.json configurations -> (poses, landmarks)
CONFIGS = {
    "C1": (50, 2_000),
    "C2": (1_000, 100_000),
    "C4": (10_000, 1_000_000),
    "LT24": (600, 60_000),       # long tracks: 24 observations per landmark (the wide reduced system; tools/bench_general.py's P600 case)
}
CONFIG_TRACK = {"LT24": 24}       # track length where it is not 12


@dataclass
class StereoBAProblem:
    """Host-side problem in the reference's data model (dataset_problem.hpp:31-54)."""

    camera: dict
    poses_gt: np.ndarray        # (P,12)
    poses_init: np.ndarray      # (P,12)
    points_gt: np.ndarray       # (L,3)
    points_init: np.ndarray     # (L,3)
    obs_pose: np.ndarray        # (N,) uint32  state index k
    obs_point: np.ndarray       # (N,) uint32  landmark index j
    obs_uvd: np.ndarray         # (N,3) float64 noisy (u,v,d)
    stereo_obs_var: np.ndarray  # (3,) variances
    outlier_mask: np.ndarray = field(default=None)  # (N,) bool, config C5 only

    @property
    def num_poses(self) -> int:
        return self.poses_init.shape[0]

    @property
    def num_points(self) -> int:
        return self.points_init.shape[0]

    @property
    def num_obs(self) -> int:
        return self.obs_pose.shape[0]

    def stiffness(self) -> np.ndarray:
        """3x3 row-major stiffness = Sigma^{-1/2} (tests/dataset_vo.cpp:29-32)."""
        return np.diag(1.0 / np.sqrt(self.stereo_obs_var))


# ---------------------------------------------------------------- geometry ---
def so3_exp(phi: np.ndarray) -> np.ndarray:
    """Rodrigues with the reference's first-order branch (so3group.hpp:273-291)."""
    phi = np.asarray(phi, dtype=np.float64)
    angle = np.linalg.norm(phi)
    W = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    if angle <= np.finfo(np.float64).eps:
        return np.eye(3) + W
    a = phi / angle
    A = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.cos(angle) * np.eye(3) + (1 - np.cos(angle)) * np.outer(a, a) + np.sin(angle) * A


def _batch_so3_exp(phi: np.ndarray) -> np.ndarray:
    ang = np.linalg.norm(phi, axis=1)
    safe = np.where(ang > 0, ang, 1.0)
    a = phi / safe[:, None]
    c, s = np.cos(ang), np.sin(ang)
    R = np.zeros((phi.shape[0], 3, 3))
    eye = np.eye(3)[None]
    A = np.zeros_like(R)
    A[:, 0, 1], A[:, 0, 2] = -a[:, 2], a[:, 1]
    A[:, 1, 0], A[:, 1, 2] = a[:, 2], -a[:, 0]
    A[:, 2, 0], A[:, 2, 1] = -a[:, 1], a[:, 0]
    R = c[:, None, None] * eye + (1 - c)[:, None, None] * (a[:, :, None] * a[:, None, :]) + s[:, None, None] * A
    small = ang <= np.finfo(np.float64).eps
    if small.any():
        W = np.zeros((small.sum(), 3, 3))
        p = phi[small]
        W[:, 0, 1], W[:, 0, 2] = -p[:, 2], p[:, 1]
        W[:, 1, 0], W[:, 1, 2] = p[:, 2], -p[:, 0]
        W[:, 2, 0], W[:, 2, 1] = -p[:, 1], p[:, 0]
        R[small] = eye + W
    return R


def pose_pack(t: np.ndarray, R: np.ndarray) -> np.ndarray:
    """(…,3),(…,3,3) -> (…,12) in the reference's [t | R row-major] layout."""
    return np.concatenate([t, R.reshape(R.shape[:-2] + (9,))], axis=-1)


def pose_unpack(x: np.ndarray):
    return x[..., :3], x[..., 3:].reshape(x.shape[:-1] + (3, 3))


def project(cam: dict, q: np.ndarray) -> np.ndarray:
    """stereo_camera.hpp:77-84"""
    iz = 1.0 / q[..., 2]
    return np.stack([cam["fu"] * q[..., 0] * iz + cam["cu"],
                     cam["fv"] * q[..., 1] * iz + cam["cv"],
                     cam["fu"] * cam["b"] * iz], axis=-1)


def triangulate(cam: dict, uvd: np.ndarray) -> np.ndarray:
    """stereo_camera.hpp:112-120"""
    b_over_d = cam["b"] / uvd[..., 2]
    return np.stack([(uvd[..., 0] - cam["cu"]) * b_over_d,
                     (uvd[..., 1] - cam["cv"]) * b_over_d * (cam["fu"] / cam["fv"]),
                     cam["fu"] * b_over_d], axis=-1)


def circle_trajectory(P: int, spacing: float = 0.5, min_radius: float = 80.0):
    """Planar circular loop, yaw following the path (author's sims: ba_all_sims.sh:8-13).

    Camera axes: z forward, x right, y down.  Short sequences follow an arc of a
    ``min_radius`` circle instead of closing a tiny loop.
    Returns (t_cg (P,3), R_cg (P,3,3)).
    """
    r = max(P * spacing / (2 * np.pi), min_radius)
    th = spacing * np.arange(P) / r
    c = np.stack([r * (1 - np.cos(th)), np.zeros(P), r * np.sin(th)], axis=1)
    xc = np.stack([np.cos(th), np.zeros(P), -np.sin(th)], axis=1)
    yc = np.tile(np.array([0.0, 1.0, 0.0]), (P, 1))
    zc = np.stack([np.sin(th), np.zeros(P), np.cos(th)], axis=1)
    R_cg = np.stack([xc, yc, zc], axis=1)          # rows are camera axes in global
    t_cg = -np.einsum("kij,kj->ki", R_cg, c)
    return t_cg, R_cg


# --------------------------------------------------------------- generator ---
def make_problem(num_poses: int, num_points: int, *, track_len: int = 12, seed: int = 42,
                 obs_var=(4.0, 4.0, 4.0), pose_sigma=(0.05, 0.01),
                 outlier_fraction: float = 0.0, depth_range=(5.0, 40.0),
                 min_depth: float = 0.5) -> StereoBAProblem:
    """Build one synthetic stereo BA problem.

    Landmark ``j`` is anchored at pose ``a_j = floor(j*P/L)``: a uniform pixel and a
    depth in ``depth_range`` are back-projected through ``triangulate`` from the
    anchor camera; it is then observed by the ``track_len`` poses ending at the
    anchor (``a_j-T+1 .. a_j``), clipped by visibility (inside the image, depth >
    ``min_depth``).  Looking *back* along the track keeps most tracks at full
    length so that N_obs ~= track_len * L, the observation count BASELINE.json
    quotes for C2.
    """
    P, L, T = int(num_poses), int(num_points), int(track_len)
    cam = dict(KITTI_CAMERA)
    rng = np.random.Generator(np.random.PCG64(seed))

    t_gt, R_gt = circle_trajectory(P)
    poses_gt = pose_pack(t_gt, R_gt)

    anchor = (np.arange(L, dtype=np.int64) * P) // L
    u0 = rng.uniform(0.0, IMAGE_W, L)
    v0 = rng.uniform(0.0, IMAGE_H, L)
    z0 = rng.uniform(depth_range[0], depth_range[1], L)
    d0 = cam["fu"] * cam["b"] / z0
    p_c = triangulate(cam, np.stack([u0, v0, d0], axis=1))
    # global = R^T (p_c - t)
    Ra, ta = R_gt[anchor], t_gt[anchor]
    points_gt = np.einsum("nji,nj->ni", Ra, p_c - ta)

    # candidate observations (landmark-major), then visibility clipping
    offs = np.arange(-(T - 1), 1, dtype=np.int64)
    k_all = (anchor[:, None] + offs[None, :]).reshape(-1)
    j_all = np.repeat(np.arange(L, dtype=np.int64), T)
    ok = (k_all >= 0) & (k_all < P)
    k_all, j_all = k_all[ok], j_all[ok]
    q = np.einsum("nij,nj->ni", R_gt[k_all], points_gt[j_all]) + t_gt[k_all]
    front = q[:, 2] > min_depth
    k_all, j_all, q = k_all[front], j_all[front], q[front]
    uvd = project(cam, q)
    vis = (uvd[:, 0] >= 0) & (uvd[:, 0] < IMAGE_W) & (uvd[:, 1] >= 0) & (uvd[:, 1] < IMAGE_H) & (uvd[:, 2] >= 1.0)
    k_all, j_all, uvd = k_all[vis], j_all[vis], uvd[vis]

    # reference file order: by state, then by landmark
    order = np.lexsort((j_all, k_all))
    k_all, j_all, uvd = k_all[order], j_all[order], uvd[order]

    sig = np.sqrt(np.asarray(obs_var, dtype=np.float64))
    uvd_noisy = uvd + rng.standard_normal(uvd.shape) * sig[None, :]
    uvd_noisy[:, 2] = np.maximum(uvd_noisy[:, 2], 0.25)   # keep disparity positive

    outlier_mask = None
    if outlier_fraction > 0.0:
        n = uvd_noisy.shape[0]
        outlier_mask = rng.uniform(size=n) < outlier_fraction
        m = int(outlier_mask.sum())
        uvd_noisy[outlier_mask, 0] = rng.uniform(0.0, IMAGE_W, m)
        uvd_noisy[outlier_mask, 1] = rng.uniform(0.0, IMAGE_H, m)
        uvd_noisy[outlier_mask, 2] = rng.uniform(1.0, 100.0, m)

    # initial poses: exp(eps) * T_gt with the reference's "exp" (se3group.hpp:323-325)
    eps = rng.standard_normal((P, 6)) * np.array([pose_sigma[0]] * 3 + [pose_sigma[1]] * 3)[None, :]
    eps[0] = 0.0
    E = _batch_so3_exp(eps[:, 3:])
    R_init = np.einsum("kij,kjl->kil", E, R_gt)
    t_init = np.einsum("kij,kj->ki", E, t_gt) + eps[:, :3]
    R_init[0], t_init[0] = R_gt[0], t_gt[0]
    poses_init = pose_pack(t_init, R_init)

    # initial landmarks: first (lowest-k) noisy observation triangulated and mapped
    # through the *initial* pose of that state (mirrors dataset_problem.cpp:263)
    first = np.full(L, -1, dtype=np.int64)
    # observations are sorted by k, so the first occurrence of j is its lowest k
    jj, idx = np.unique(j_all, return_index=True)
    first[jj] = idx
    points_init = points_gt.copy()
    has = first >= 0
    fo = first[has]
    pc = triangulate(cam, uvd_noisy[fo] if outlier_mask is None else np.where(outlier_mask[fo, None], uvd[fo], uvd_noisy[fo]))
    kf = k_all[fo]
    points_init[has] = np.einsum("nji,nj->ni", R_init[kf], pc - t_init[kf])

    return StereoBAProblem(
        camera=cam, poses_gt=poses_gt, poses_init=poses_init,
        points_gt=points_gt, points_init=points_init,
        obs_pose=k_all.astype(np.uint32), obs_point=j_all.astype(np.uint32),
        obs_uvd=np.ascontiguousarray(uvd_noisy), stereo_obs_var=np.asarray(obs_var, dtype=np.float64),
        outlier_mask=outlier_mask)


def _in_view(cam, q, min_depth: float = 2.0):
    """Points (camera frame) in front of the camera and inside the image."""
    z = np.where(q[:, 2] > min_depth, q[:, 2], np.inf)
    u, v = cam["fu"] * q[:, 0] / z + cam["cu"], cam["fv"] * q[:, 1] / z + cam["cv"]
    return (q[:, 2] > min_depth) & (u >= 0) & (u < IMAGE_W) & (v >= 0) & (v < IMAGE_H)


def add_loop_closure(prob: StereoBAProblem, num_states: int = 3, num_landmarks: int = 60, seed: int = 2,
                     sigma: float = 0.5, max_track: int | None = None) -> StereoBAProblem:
    """Copy of `prob` in which landmarks first seen from the first three states are observed again from the last
    `num_states` states (they pass through the ground-truth poses, so the problem stays consistent): the pose
    co-visibility is no longer banded, which is what a loop closure does to the reduced camera system.  `max_track`
    keeps the re-observed landmarks within that many observations (12: the windowed layout still holds them, and the
    library carries the closing states as a border of the block-tridiagonal system instead of taking the general path)."""
    import copy
    rng = np.random.default_rng(seed)
    P = prob.num_poses
    seen_first = np.zeros(prob.num_points, bool)
    seen_first[prob.obs_point[prob.obs_pose < 3]] = True
    if max_track is not None:       # only landmarks whose track stays within max_track observations after the closure
        seen_first &= np.bincount(prob.obs_point, minlength=prob.num_points) + num_states <= max_track
    first = np.flatnonzero(seen_first)[:num_landmarks]
    k_new, j_new, uvd_new = [], [], []
    k_end = P
    if max_track is not None:       # close the loop from the latest states that still have those landmarks in front of them
        for k in range(P - 1, 15 + num_states, -1):
            t, R = prob.poses_gt[k, :3], prob.poses_gt[k, 3:].reshape(3, 3)
            if int(_in_view(prob.camera, prob.points_gt[first] @ R.T + t).sum()) >= max(8, len(first) // 8):
                k_end = k + 1
                break
    for k in range(k_end - num_states, k_end):
        already = set(prob.obs_point[prob.obs_pose == k].tolist())
        cand = np.asarray([j for j in first if j not in already], dtype=np.int64)
        if cand.size == 0:
            continue
        t, R = prob.poses_gt[k, :3], prob.poses_gt[k, 3:].reshape(3, 3)
        q = prob.points_gt[cand] @ R.T + t
        ok = q[:, 2] > 0.5 if max_track is None else _in_view(prob.camera, q)
        uvd = project(prob.camera, q[ok]) + rng.normal(size=(int(ok.sum()), 3)) * sigma
        uvd[:, 2] = np.maximum(uvd[:, 2], 0.25)
        k_new += [k] * int(ok.sum())
        j_new += cand[ok].tolist()
        uvd_new.append(uvd)
    out = copy.copy(prob)
    out.obs_pose = np.concatenate([prob.obs_pose, np.asarray(k_new, np.uint32)]).astype(np.uint32)
    out.obs_point = np.concatenate([prob.obs_point, np.asarray(j_new, np.uint32)]).astype(np.uint32)
    out.obs_uvd = np.vstack([prob.obs_uvd] + uvd_new)
    if prob.outlier_mask is not None:
        out.outlier_mask = np.concatenate([prob.outlier_mask, np.zeros(len(k_new), bool)])
    return out


def make_config(name: str, **kw) -> StereoBAProblem:
    P, L = CONFIGS[name]
    kw.setdefault("track_len", CONFIG_TRACK.get(name, 12))
    return make_problem(P, L, **kw)


# ------------------------------------------------------------------ file I/O ---
def _T44_rows(poses12: np.ndarray) -> np.ndarray:
    t, R = pose_unpack(poses12)
    T = np.zeros(poses12.shape[:-1] + (4, 4))
    T[..., :3, :3], T[..., :3, 3], T[..., 3, 3] = R, t, 1.0
    return T.reshape(poses12.shape[:-1] + (16,))


def write_reference_csv(prob: StereoBAProblem, dataset_path: str) -> tuple:
    """Writes the problem in the reference's on-disk formats: the 5-column dataset CSV of
    src/ceres_slam/dataset_problem.cpp:16-83 and the initial guess as the `_poses.csv` /
    `_map.csv` pair its write_csv emits (:121-165), at full double precision.
    Returns (dataset_path, init_poses_path, init_map_path)."""
    base = dataset_path[: dataset_path.rfind(".")] if "." in dataset_path else dataset_path
    c = prob.camera
    with open(dataset_path, "w") as f:
        f.write(f"{prob.num_poses},{prob.num_points}\n")
        f.write(",".join(repr(float(c[k])) for k in ("fu", "fv", "cu", "cv", "b")) + "\n")
        f.write(",".join(repr(float(v)) for v in prob.stereo_obs_var) + "\n")
        f.write(",".join(repr(float(v)) for v in _T44_rows(prob.poses_gt[0])) + "\n")
        for k, j, (u, v, d) in zip(prob.obs_pose, prob.obs_point, prob.obs_uvd):
            f.write(f"{int(k)},{int(j)},{float(u)!r},{float(v)!r},{float(d)!r}\n")
    poses_path, map_path = base + "_init_poses.csv", base + "_init_map.csv"
    with open(poses_path, "w") as f:
        f.write("T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33\n")
        for row in _T44_rows(prob.poses_init):
            f.write(",".join(repr(float(v)) for v in row) + "\n")
    with open(map_path, "w") as f:
        f.write("point_id, x, y, z\n")
        for j, p in enumerate(prob.points_init):
            f.write(f"{j}," + ",".join(repr(float(v)) for v in p) + "\n")
    return dataset_path, poses_path, map_path


def make_sun_data(prob: StereoBAProblem, seed: int = 0, sun_sigma: float = 0.01) -> dict:
    """Extra data of the sun-aided VO driver (tests/dataset_vo_sun.cpp, DatasetProblemSun): a 3x3 covariance per stereo
    observation, the expected sun direction per state in the global frame, and for three states out of four an
    observed direction in the camera frame (R_k s_g + noise) with a 2x2 azimuth / zenith covariance."""
    rng = np.random.default_rng(seed)
    N, P = prob.num_obs, prob.num_poses
    A = rng.normal(size=(N, 3, 3)) * 0.1
    obs_covars = A @ A.transpose(0, 2, 1) + np.diag(prob.stereo_obs_var)
    sun_g = np.array([0.3, -0.8, 0.5])
    sun_g = sun_g / np.linalg.norm(sun_g)
    sun_dir_g = np.tile(sun_g, (P, 1))
    has_sun = (np.arange(P) % 4) != 3
    R = prob.poses_gt[:, 3:].reshape(P, 3, 3)
    sun_obs = np.einsum("kij,j->ki", R, sun_g) + sun_sigma * rng.normal(size=(P, 3))
    B = rng.normal(size=(P, 2, 2)) * 0.002
    sun_covars = B @ B.transpose(0, 2, 1) + np.eye(2) * sun_sigma ** 2
    return dict(obs_covars=obs_covars, sun_dir_g=sun_dir_g, sun_obs=sun_obs, sun_covars=sun_covars, has_sun=has_sun)


def write_reference_sun_csv(prob: StereoBAProblem, sun: dict, track_path: str) -> tuple:
    """The three input files of tests/dataset_vo_sun.cpp in the formats DatasetProblemSun::read_csv parses
    (src/ceres_slam/dataset_problem_sun.cpp:16-170).  Returns (track_file, ref_sun_file, obs_sun_file)."""
    base = track_path[: track_path.rfind(".")] if "." in track_path else track_path
    c = prob.camera
    with open(track_path, "w") as f:
        f.write(f"{prob.num_poses},{prob.num_points}\n")
        f.write(",".join(repr(float(c[k])) for k in ("fu", "fv", "cu", "cv", "b")) + "\n")
        f.write(",".join(repr(float(v)) for v in _T44_rows(prob.poses_gt[0])) + "\n")
        for k, j, uvd, cov in zip(prob.obs_pose, prob.obs_point, prob.obs_uvd, sun["obs_covars"]):
            f.write(f"{int(k)},{int(j)}," + ",".join(repr(float(v)) for v in np.concatenate([uvd, cov.ravel()])) + "\n")
    ref_path, obs_path = base + "_ref_sun.csv", base + "_obs_sun.csv"
    with open(ref_path, "w") as f:
        for k, v in enumerate(sun["sun_dir_g"]):
            f.write(f"{k}," + ",".join(repr(float(x)) for x in v) + "\n")
    with open(obs_path, "w") as f:
        for k in range(prob.num_poses):
            if sun["has_sun"][k]:
                f.write(f"{k}," + ",".join(repr(float(x)) for x in np.concatenate([sun["sun_obs"][k], sun["sun_covars"][k].ravel()])) + "\n")
    return track_path, ref_path, obs_path


def read_pose_csv(path: str) -> np.ndarray:
    rows = [l for l in open(path).read().splitlines()[1:] if l.strip()]
    T = np.array([[float(x) for x in r.split(",")] for r in rows]).reshape(-1, 4, 4)
    return pose_pack(T[:, :3, 3], T[:, :3, :3])


# ------------------------------------------------------------- Phong lighting ---
@dataclass
class PhongData:
    """Lighting side of a config-3 problem (dataset_problem_phong.hpp:34-81 data model)."""
    normals_gt: np.ndarray         # (L,3) unit
    normals_init: np.ndarray       # (L,3) unit
    material_of_point: np.ndarray  # (L,) uint32
    phong: np.ndarray              # (M,3) ka, ks, alpha
    texture: np.ndarray            # (M,)  kd
    light: np.ndarray              # (3,) position (light_type 0) or direction (1), global frame
    light_type: int
    intensity: np.ndarray          # (N,) observed intensities
    normal_obs: np.ndarray         # (N,3) observed normals, camera frame
    int_var: float
    normal_obs_var: np.ndarray     # (3,)
    # initial guesses of the shared blocks when they are optimised (ssba_set_shared_blocks_free)
    phong_init: np.ndarray = None    # (M,3) perturbed truth, interior to the driver's bounds
    texture_init: np.ndarray = None  # (M,)
    light_init: np.ndarray = None    # (3,)
    phong_ref_init: np.ndarray = None    # (M,3) = (0,0,1): dataset_problem_phong.cpp:266-267
    texture_ref_init: np.ndarray = None  # (M,) median observed intensity of the material: :269-277

    @property
    def int_stiffness(self) -> float:
        return 1.0 / np.sqrt(self.int_var)                       # tests/dataset_ba_phong.cpp:44

    def normal_stiffness(self) -> np.ndarray:
        return np.diag(1.0 / np.sqrt(self.normal_obs_var))       # tests/dataset_ba_phong.cpp:39-42

    def as_oracle_dict(self, shared: str = "truth") -> dict:
        """shared = "truth" (the constant-block configuration), "perturbed" (interior initial guess) or
        "reference" (the reference's compute_initial_guess values; the light stays "perturbed")."""
        phong, texture, light = self.phong, self.texture, self.light
        if shared == "perturbed":
            phong, texture, light = self.phong_init, self.texture_init, self.light_init
        elif shared == "reference":
            phong, texture, light = self.phong_ref_init, self.texture_ref_init, self.light_init
        return dict(normals=self.normals_init, intensity=self.intensity, normal_obs=self.normal_obs, phong=phong.copy(),
                    texture=texture.copy(), material_of_point=self.material_of_point, light=light.copy(),
                    light_type=self.light_type, int_stiffness=self.int_stiffness, normal_stiffness=self.normal_stiffness())


def _shade(n_c, ell, cd, kd, ks, alpha):
    """lighting/phong.hpp:25-51 vectorised (ambient = 0, guards, clamp)."""
    ldn = (ell * n_c).sum(1)
    diffuse = np.where(ldn > 0, kd * ldn, 0.0)
    mt = 2 * ldn[:, None] * n_c - ell
    mu = np.linalg.norm(mt, axis=1)
    m = mt / np.where(mu > 0, mu, 1.0)[:, None]
    s = (m * cd).sum(1)
    spec = np.where((mu > 0) & (s > 0), ks * np.power(np.where(s > 0, s, 1.0), alpha), 0.0)
    return np.clip(diffuse + spec, 0.0, 1.0)


def make_phong_problem(num_poses: int, num_points: int, *, num_materials: int = 4, light_type: int = 0,
                       int_var: float = 1e-4, normal_var: float = 1e-4, seed: int = 42, **kw):
    """Config 3: the stereo problem of `make_problem` plus unit normals, one
    light, `num_materials` Phong materials (ks in [0,0.5], alpha in [1,20]) with textures kd in
    [0.2,0.9], intensities from the Phong model + N(0, int_var) and normal observations + N(0, var).
    Returns (StereoBAProblem, PhongData)."""
    prob = make_problem(num_poses, num_points, seed=seed, **kw)
    rng = np.random.Generator(np.random.PCG64(seed + 1000003))
    P, L = prob.num_poses, prob.num_points
    t_gt, R_gt = pose_unpack(prob.poses_gt)
    centers = -np.einsum("kji,kj->ki", R_gt, t_gt)
    anchor = (np.arange(L, dtype=np.int64) * P) // L
    if light_type == 0:
        light = centers.mean(0) + np.array([0.0, -25.0, 0.0])          # above the loop (y points down)
        to_l = light[None] - prob.points_gt
    else:
        light = np.array([0.3, -0.8, 0.52])
        light /= np.linalg.norm(light)
        to_l = np.tile(light, (L, 1))
    to_c = centers[anchor] - prob.points_gt
    bis = to_c / np.linalg.norm(to_c, axis=1)[:, None] + to_l / np.linalg.norm(to_l, axis=1)[:, None]
    n = bis / np.linalg.norm(bis, axis=1)[:, None] + 0.25 * rng.standard_normal((L, 3))
    normals_gt = n / np.linalg.norm(n, axis=1)[:, None]
    mat = rng.integers(0, num_materials, L).astype(np.uint32)
    phong = np.stack([np.zeros(num_materials), rng.uniform(0.0, 0.5, num_materials), rng.uniform(1.0, 20.0, num_materials)], 1)
    texture = rng.uniform(0.2, 0.9, num_materials)
    k, j = prob.obs_pose.astype(np.int64), prob.obs_point.astype(np.int64)
    q = np.einsum("nij,nj->ni", R_gt[k], prob.points_gt[j]) + t_gt[k]
    n_c = np.einsum("nij,nj->ni", R_gt[k], normals_gt[j])
    if light_type == 0:
        l_c = np.einsum("nij,j->ni", R_gt[k], light) + t_gt[k]
        v = l_c - q
    else:
        v = np.einsum("nij,j->ni", R_gt[k], light)
    ell = v / np.linalg.norm(v, axis=1)[:, None]
    cd = -q / np.linalg.norm(q, axis=1)[:, None]
    inten = _shade(n_c, ell, cd, texture[mat[j]], phong[mat[j], 1], phong[mat[j], 2])
    intensity = inten + rng.standard_normal(inten.shape) * np.sqrt(int_var)
    normal_obs = n_c + rng.standard_normal(n_c.shape) * np.sqrt(normal_var)
    # initial normals: first observation mapped through the initial pose of that state, normalised
    t_i, R_i = pose_unpack(prob.poses_init)
    normals_init = normals_gt.copy()
    jj, idx = np.unique(j, return_index=True)
    n0 = np.einsum("nji,nj->ni", R_i[k[idx]], normal_obs[idx])
    normals_init[jj] = n0 / np.linalg.norm(n0, axis=1)[:, None]
    # shared-block initial guesses (drawn last so that the data above do not depend on them)
    phong_init = phong * np.array([1.0, 1.0, 1.0]) + np.stack([np.zeros(num_materials), 0.05 * rng.standard_normal(num_materials),
                                                              1.0 * rng.standard_normal(num_materials)], 1)
    phong_init[:, 1] = np.clip(phong_init[:, 1], 0.02, 0.98)
    phong_init[:, 2] = np.maximum(phong_init[:, 2], 1.2)
    texture_init = np.clip(texture + 0.05 * rng.standard_normal(num_materials), 0.05, 0.95)
    if light_type == 0:
        light_init = light + 0.5 * rng.standard_normal(3)
    else:
        light_init = light + 0.05 * rng.standard_normal(3)
        light_init /= np.linalg.norm(light_init)
    texture_ref = np.zeros(num_materials)
    for m in range(num_materials):
        ints = np.sort(intensity[mat[j] == m])
        texture_ref[m] = ints[len(ints) // 2] if len(ints) else 0.5       # std::nth_element(begin + size/2)
    phong_ref = np.tile(np.array([0.0, 0.0, 1.0]), (num_materials, 1))
    return prob, PhongData(normals_gt, normals_init, mat, phong, texture, light, light_type, intensity, normal_obs,
                           int_var, np.full(3, normal_var), phong_init, texture_init, light_init, phong_ref, texture_ref)


def write_reference_phong_csv(prob: StereoBAProblem, ph: PhongData, dataset_path: str, shared: str = "reference") -> tuple:
    """Config-3 problem in the reference's on-disk formats: the 10-column dataset CSV of
    src/ceres_slam/dataset_problem_phong.cpp:16-117 (5 header rows, then `t,j,material,u,v,d,I,nx,ny,nz`)
    and the initial guess as the `_poses.csv` / `_map.csv` / `_lights.csv` triple its write_csv emits
    (:177-232), at full double precision.  `shared` picks the initial shared blocks as in
    PhongData.as_oracle_dict.  Returns (dataset, init_poses, init_map, init_lights)."""
    base = dataset_path[: dataset_path.rfind(".")] if "." in dataset_path else dataset_path
    d = ph.as_oracle_dict(shared)
    c = prob.camera
    mat = ph.material_of_point
    with open(dataset_path, "w") as f:
        f.write(f"{prob.num_poses},{prob.num_points},{len(ph.texture)}\n")
        f.write(",".join(repr(float(c[k])) for k in ("fu", "fv", "cu", "cv", "b")) + "\n")
        f.write(",".join(repr(float(v)) for v in list(prob.stereo_obs_var) + list(ph.normal_obs_var) + [ph.int_var]) + "\n")
        f.write(",".join(repr(float(v)) for v in d["light"]) + "\n")
        f.write(",".join(repr(float(v)) for v in _T44_rows(prob.poses_gt[0])) + "\n")
        for i in range(prob.num_obs):
            k, j = int(prob.obs_pose[i]), int(prob.obs_point[i])
            vals = list(prob.obs_uvd[i]) + [ph.intensity[i]] + list(ph.normal_obs[i])
            f.write(f"{float(k)!r},{j},{int(mat[j])}," + ",".join(repr(float(v)) for v in vals) + "\n")
    poses_path, map_path, light_path = base + "_init_poses.csv", base + "_init_map.csv", base + "_init_lights.csv"
    with open(poses_path, "w") as f:
        f.write("T_00, T_01, T_02, T_03,T_10, T_11, T_12, T_13,T_20, T_21, T_22, T_23,T_30, T_31, T_32, T_33\n")
        for row in _T44_rows(prob.poses_init):
            f.write(",".join(repr(float(v)) for v in row) + "\n")
    with open(map_path, "w") as f:
        f.write("point_id, x, y, z, nx, ny, nz, ka, ks, exponent, kd\n")
        for j in range(prob.num_points):
            vals = list(prob.points_init[j]) + list(ph.normals_init[j]) + list(d["phong"][mat[j]]) + [d["texture"][mat[j]]]
            f.write(f"{j}," + ",".join(repr(float(v)) for v in vals) + "\n")
    with open(light_path, "w") as f:
        f.write(("i, j, k" if ph.light_type == 1 else "x, y, z") + "\n")
        f.write(",".join(repr(float(v)) for v in d["light"]) + "\n")
    return dataset_path, poses_path, map_path, light_path


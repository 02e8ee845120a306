"""Front end: the VO initial guess the reference computes before a solve,
DatasetProblem::compute_initial_guess (src/ceres_slam/dataset_problem.cpp:179-270), with the 3-point RANSAC of
every pair of consecutive states scored on the GPU in one batch (ssba_frontend_ransac).

Host logic mirrors the reference: per state the observation list in file order; reciprocal matching by point id
(:207-223); StereoCamera::triangulate of both lists (:225-232); T_k_km1 from the RANSAC (:246-249); poses[k] =
T_k_km1 * poses[k-1] (:256); inlier points not yet initialised become poses[k-1]^-1 * p_km1 (:260-269).  The
pairs are independent, so all of them are scored together before the (cheap, sequential) chaining."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def triangulate(camera: dict, uvd: np.ndarray) -> np.ndarray:
    """StereoCamera::triangulate (include/ceres_slam/stereo_camera.hpp:112-120)."""
    b_over_d = camera["b"] / uvd[:, 2]
    return np.stack([(uvd[:, 0] - camera["cu"]) * b_over_d, (uvd[:, 1] - camera["cv"]) * b_over_d * (camera["fu"] / camera["fv"]),
                     camera["fu"] * b_over_d], axis=1)


def match_states(ids_km1: np.ndarray, ids_k: np.ndarray):
    """Reciprocal matches (:207-223): positions kept in each list, in that list's own order."""
    keep_km1 = np.isin(ids_km1, ids_k)
    keep_k = np.isin(ids_k, ids_km1[keep_km1])
    return np.nonzero(keep_km1)[0], np.nonzero(keep_k)[0]


def ransac_samples(n: int, num_iters: int = 400, variant: int = 1) -> np.ndarray:
    idx = np.zeros(3 * num_iters, dtype=np.uint32)
    capi.check(capi.load().ssba_ransac_samples(n, num_iters, variant, idx.ctypes.data_as(capi._u32p)), "ssba_ransac_samples")
    return idx.reshape(num_iters, 3)


def ransac_batch(camera: dict, pts0_list, pts1_list, num_iters: int = 400, thresh: float = 4.0, variant: int = 1, device: int = -1):
    """All pairs at once.  Returns (T (n_pairs,12), list of inlier masks, counts, device seconds)."""
    lib = capi.load()
    n_pairs = len(pts0_list)
    sizes = np.array([len(p) for p in pts0_list], dtype=np.int64)
    offset = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    pts0 = np.ascontiguousarray(np.concatenate(pts0_list) if n_pairs else np.zeros((0, 3)))
    pts1 = np.ascontiguousarray(np.concatenate(pts1_list) if n_pairs else np.zeros((0, 3)))
    cache = {}
    samples = np.zeros((n_pairs, num_iters, 3), dtype=np.uint32)
    for p, n in enumerate(sizes):
        if int(n) not in cache:                 # the draw sequence depends on n only (the generator is re-seeded per call)
            cache[int(n)] = ransac_samples(int(n), num_iters, variant)
        samples[p] = cache[int(n)]
    T = np.zeros((n_pairs, 12))
    inl = np.zeros(max(int(offset[-1]), 1), dtype=np.uint8)
    cnt = np.zeros(max(n_pairs, 1), dtype=np.uint32)
    secs = C.c_double(0.0)
    cam = capi.Camera(**camera)
    capi.check(lib.ssba_frontend_ransac(C.byref(cam), device, n_pairs, offset.ctypes.data_as(capi._u32p), capi.dptr(pts0), capi.dptr(pts1),
                                        samples.ctypes.data_as(capi._u32p), num_iters, float(thresh), capi.dptr(T),
                                        inl.ctypes.data_as(C.POINTER(C.c_uint8)), cnt.ctypes.data_as(capi._u32p), C.byref(secs)),
               "ssba_frontend_ransac")
    masks = [inl[offset[p]:offset[p + 1]].astype(bool) for p in range(n_pairs)]
    return T, masks, cnt[:n_pairs].copy(), secs.value


def se3_compose(Ta: np.ndarray, Tb: np.ndarray) -> np.ndarray:
    """T_a * T_b on 12-double [t | R row-major] blocks (se3group.hpp:224-234)."""
    Ra, Rb = Ta[3:].reshape(3, 3), Tb[3:].reshape(3, 3)
    return np.concatenate([Ra @ Tb[:3] + Ta[:3], (Ra @ Rb).ravel()])


def se3_inverse_apply(T: np.ndarray, p: np.ndarray) -> np.ndarray:
    """T^-1 * p for points p (n,3)."""
    R = T[3:].reshape(3, 3)
    return (p - T[:3]) @ R


def compute_initial_guess(camera: dict, num_states: int, num_points: int, obs_state, obs_point, obs_uvd, first_pose,
                          num_iters: int = 400, thresh: float = 4.0, variant: int = 1, device: int = -1, ransac=None):
    """DatasetProblem::compute_initial_guess(0, num_states).  `ransac` (default: the GPU batch) is a callable with
    ransac_batch's signature -- tests pass an oracle-backed one.  Returns poses (P,12), points (L,3),
    initialised flags (L,), stats dict."""
    obs_state = np.asarray(obs_state)
    obs_point = np.asarray(obs_point)
    idx_of = [np.nonzero(obs_state == k)[0] for k in range(num_states)]     # file order inside a state (:86-98)
    pairs = []
    for k in range(1, num_states):
        a, b = match_states(obs_point[idx_of[k - 1]], obs_point[idx_of[k]])
        ia, ib = idx_of[k - 1][a], idx_of[k][b]
        pairs.append((ia, triangulate(camera, obs_uvd[ia]), triangulate(camera, obs_uvd[ib])))
    usable = [len(p[0]) >= 3 for p in pairs]
    T, masks, counts, secs = (ransac or ransac_batch)(camera, [p[1] for p, u in zip(pairs, usable) if u],
                                                      [p[2] for p, u in zip(pairs, usable) if u], num_iters, thresh, variant, device)
    poses = np.zeros((num_states, 12))
    poses[0] = first_pose
    points = np.zeros((num_points, 3))
    initialized = np.zeros(num_points, dtype=bool)
    q = 0
    for k in range(1, num_states):
        ia, p_km1, _ = pairs[k - 1]
        if usable[k - 1]:
            Tk, mask = T[q], masks[q]
            q += 1
        else:                                    # fewer than 3 matches: the reference's aligner is undefined there
            Tk, mask = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0]), np.zeros(len(ia), dtype=bool)
        poses[k] = se3_compose(Tk, poses[k - 1])
        js = obs_point[ia][mask]
        new = ~initialized[js]
        points[js[new]] = se3_inverse_apply(poses[k - 1], p_km1[mask][new])
        initialized[js[new]] = True
    return poses, points, initialized, dict(pairs=len(pairs), matches=int(sum(len(p[0]) for p in pairs)),
                                            inliers=int(counts.sum()), ransac_device_s=secs)


def compute_initial_guess_phong(camera: dict, num_states: int, num_points: int, num_materials: int, obs_state, obs_point, obs_material,
                                obs_uvd, obs_intensity, obs_normal, first_pose, num_iters: int = 400, thresh: float = 9.0,
                                variant: int = 1, device: int = -1, ransac=None, reference_material_indexing: bool = True):
    """DatasetProblemPhong::compute_initial_guess(0, num_states) (src/ceres_slam/dataset_problem_phong.cpp:250-391).
    On top of the stereo version: materials start at (ka, ks, exponent) = (0, 0, 1) (:266-267), textures at the
    median observed intensity of the material (std::nth_element at size/2, :269-277); an inlier vertex without a guess
    gets position poses[k-1]^-1 * p_km1 and normal = poses[k-1]^-1 * observed normal (rotation only, :361-365).

    reference_material_indexing=True keeps the reference's indexing of the vertex's material, `material_ids[i]` with
    i the position of the match in the pair's list (:369-370) rather than the observation's own material id; False
    uses the observation's material (what the data say).  Returns poses, positions, normals, initialised flags,
    material_of_vertex, phong (M,3), texture (M,), stats."""
    obs_state, obs_point, obs_material = np.asarray(obs_state), np.asarray(obs_point), np.asarray(obs_material)
    phong = np.tile(np.array([0.0, 0.0, 1.0]), (num_materials, 1))
    texture = np.zeros(num_materials)
    for m in range(num_materials):
        ints = np.sort(np.asarray(obs_intensity)[obs_material == m])
        if len(ints):
            texture[m] = ints[len(ints) // 2]
    idx_of = [np.nonzero(obs_state == k)[0] for k in range(num_states)]
    pairs = []
    for k in range(1, num_states):
        a, b = match_states(obs_point[idx_of[k - 1]], obs_point[idx_of[k]])
        ia, ib = idx_of[k - 1][a], idx_of[k][b]
        pairs.append((ia, triangulate(camera, obs_uvd[ia]), triangulate(camera, obs_uvd[ib])))
    T, masks, counts, secs = (ransac or ransac_batch)(camera, [p[1] for p in pairs], [p[2] for p in pairs], num_iters, thresh, variant, device)
    poses = np.zeros((num_states, 12))
    poses[0] = first_pose
    positions, normals = np.zeros((num_points, 3)), np.zeros((num_points, 3))
    material_of_vertex = np.zeros(num_points, dtype=np.uint32)
    initialized = np.zeros(num_points, dtype=bool)
    for k in range(1, num_states):
        ia, p_km1, _ = pairs[k - 1]
        poses[k] = se3_compose(T[k - 1], poses[k - 1])
        R = poses[k - 1][3:].reshape(3, 3)
        for i in np.nonzero(masks[k - 1])[0]:           # in list order: "if (!initialized_vertex[j])"
            j = int(obs_point[ia[i]])
            if initialized[j]:
                continue
            positions[j] = se3_inverse_apply(poses[k - 1], p_km1[i][None])[0]
            normals[j] = obs_normal[ia[i]] @ R              # R^T n
            material_of_vertex[j] = obs_material[i] if reference_material_indexing else obs_material[ia[i]]
            initialized[j] = True
    return poses, positions, normals, initialized, material_of_vertex, phong, texture, dict(
        pairs=len(pairs), matches=int(sum(len(p[0]) for p in pairs)), inliers=int(counts.sum()), ransac_device_s=secs)


def compute_initial_guess_device(camera: dict, num_states: int, num_points: int, obs_state, obs_point, obs_uvd, first_pose,
                                 num_iters: int = 400, thresh: float = 4.0, variant: int = 1, device: int = -1):
    """DatasetProblem::compute_initial_guess(0, num_states) entirely on the device (ssba_frontend_vo): matching,
    triangulation, RANSAC, pose chaining and map initialisation.  Returns (poses, points, initialized, stats)."""
    import ctypes as C
    lib = capi.load()
    obs_state = np.asarray(obs_state)
    order = np.argsort(obs_state, kind="stable")             # observations grouped by state, file order inside a state
    start = np.zeros(num_states + 1, dtype=np.uint32)
    np.cumsum(np.bincount(obs_state, minlength=num_states), out=start[1:])
    ids = np.ascontiguousarray(np.asarray(obs_point)[order], dtype=np.uint32)
    uvd = np.ascontiguousarray(np.asarray(obs_uvd, dtype=np.float64)[order])
    poses = np.zeros((num_states, 12))
    poses[0] = first_pose
    points = np.zeros((num_points, 3))
    init = np.zeros(num_points, dtype=np.uint8)
    mcnt = np.zeros(max(num_states - 1, 1), dtype=np.uint32)
    icnt = np.zeros(max(num_states - 1, 1), dtype=np.uint32)
    secs = C.c_double(0.0)
    cam = capi.Camera(**camera)
    u32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    capi.check(lib.ssba_frontend_vo(C.byref(cam), device, num_states, start.ctypes.data_as(u32p), ids.ctypes.data_as(u32p), capi.dptr(uvd),
                                    num_points, num_iters, thresh, variant, capi.dptr(poses), capi.dptr(points), init.ctypes.data_as(u8p),
                                    mcnt.ctypes.data_as(u32p), icnt.ctypes.data_as(u32p), C.byref(secs)), "ssba_frontend_vo")
    return poses, points, init.astype(bool), dict(matches=int(mcnt[:num_states - 1].sum()), inliers=int(icnt[:num_states - 1].sum()), device_s=secs.value)

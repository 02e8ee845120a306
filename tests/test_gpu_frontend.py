"""Front end on the GPU: the batched 3-point RANSAC through the C ABI against the oracle's
restatement of point_cloud_aligner.cpp pair by pair, the whole compute_initial_guess, and dataset -> initial
guess -> solve end to end."""
import ctypes as C

import numpy as np
import pytest

from ceres_slam_amd import capi, frontend, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
_u32p = C.POINTER(C.c_uint32)
_dp = C.POINTER(C.c_double)


def _oracle_ransac(camera, pts0_list, pts1_list, num_iters, thresh, variant, device):
    L = orc.lib()
    L.orc_ransac_samples.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _u32p]
    L.orc_ransac_align.argtypes = [C.POINTER(orc.Camera), _dp, _dp, C.c_int, _u32p, C.c_int, C.c_double, _dp, C.POINTER(C.c_uint8)]
    L.orc_ransac_align.restype = C.c_int
    cam = orc.Camera(**camera)
    Ts, masks, counts = [], [], []
    for p0, p1 in zip(pts0_list, pts1_list):
        n = len(p0)
        idx = np.zeros(3 * num_iters, dtype=np.uint32)
        L.orc_ransac_samples(n, num_iters, variant, idx.ctypes.data_as(_u32p))
        T, inl = np.zeros(12), np.zeros(n, dtype=np.uint8)
        p0c, p1c = np.ascontiguousarray(p0), np.ascontiguousarray(p1)
        c = L.orc_ransac_align(C.byref(cam), p0c.ctypes.data_as(_dp), p1c.ctypes.data_as(_dp), n, idx.ctypes.data_as(_u32p), num_iters,
                               thresh, T.ctypes.data_as(_dp), inl.ctypes.data_as(C.POINTER(C.c_uint8)))
        Ts.append(T); masks.append(inl.astype(bool)); counts.append(c)
    return np.array(Ts), masks, np.array(counts, dtype=np.uint32), 0.0


def _problem(P=20, L=1500, **kw):
    # feature noise of a real tracker (sub-pixel); the generator's default sigma = 2 px is the BA stress setting
    return synth.make_problem(P, L, obs_var=(0.04, 0.04, 0.04), **kw)


def test_sampling_sequence_matches_the_oracle():
    L = orc.lib()
    L.orc_ransac_samples.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _u32p]
    for n in (3, 17, 1187):
        for variant in (0, 1):
            want = np.zeros(1200, dtype=np.uint32)
            L.orc_ransac_samples(n, 400, variant, want.ctypes.data_as(_u32p))
            assert np.array_equal(frontend.ransac_samples(n, 400, variant).ravel(), want)


@pytest.mark.parametrize("outliers", [0.0, 0.25])
def test_batched_ransac_matches_oracle_pair_by_pair(outliers):
    prob = _problem(outlier_fraction=outliers)
    idx_of = [np.nonzero(prob.obs_pose == k)[0] for k in range(prob.num_poses)]
    p0s, p1s = [], []
    for k in range(1, prob.num_poses):
        a, b = frontend.match_states(prob.obs_point[idx_of[k - 1]], prob.obs_point[idx_of[k]])
        p0s.append(frontend.triangulate(prob.camera, prob.obs_uvd[idx_of[k - 1][a]]))
        p1s.append(frontend.triangulate(prob.camera, prob.obs_uvd[idx_of[k][b]]))
    T, masks, counts, secs = frontend.ransac_batch(prob.camera, p0s, p1s, 400, 4.0, 1)
    T2, masks2, counts2, _ = _oracle_ransac(prob.camera, p0s, p1s, 400, 4.0, 1, -1)
    assert np.array_equal(counts, counts2)
    np.testing.assert_allclose(T, T2, rtol=1e-9, atol=1e-10)
    assert all(np.array_equal(a, b) for a, b in zip(masks, masks2))
    assert counts.min() > 0.5 * min(len(p) for p in p0s) * (1 - 2 * outliers)
    assert secs > 0


def test_compute_initial_guess_matches_the_oracle_backed_restatement_and_feeds_the_solver():
    prob = _problem(30, 2400)
    args = (prob.camera, prob.num_poses, prob.num_points, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.poses_gt[0])
    poses, points, init, stats = frontend.compute_initial_guess(*args)
    poses2, points2, init2, _ = frontend.compute_initial_guess(*args, ransac=_oracle_ransac)
    # each T agrees to ~1e-9 (3-point samples can be close to collinear: the SVD amplifies rounding); 29 of them chained
    np.testing.assert_allclose(poses, poses2, rtol=1e-6, atol=1e-7)
    assert np.array_equal(init, init2)
    np.testing.assert_allclose(points[init], points2[init], rtol=1e-6, atol=1e-6)
    assert init.mean() > 0.95 and stats["inliers"] > 0.8 * stats["matches"]
    # VO drift stays small on this sequence, and the guess is good enough for the bundle adjustment
    assert np.abs(poses[:, :3] - prob.poses_gt[:, :3]).max() < 1.0
    sel = init[prob.obs_point]                                  # only initialised map points get residuals (dataset_vo.cpp:45)
    ba = StereoBA(prob.camera, poses.copy(), points.copy(), prob.obs_pose[sel], prob.obs_point[sel], prob.obs_uvd[sel],
                  prob.stiffness())
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    assert s.termination_type == 0 and s.final_cost < 0.05 * s.initial_cost
    assert np.abs(ba.poses[:, :3] - prob.poses_gt[:, :3]).max() < 0.05


def test_cpp_driver_with_its_own_front_end(tmp_path):
    """examples/dataset_vo_gpu --frontend: read the dataset, compute the initial guess (GPU RANSAC), solve, write --
    the whole main() of tests/dataset_vo.cpp; same result as the Python mirror of the front end + the oracle."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples()
    prob = _problem(16, 800, track_len=8, seed=3)
    ds, _, _ = synth.write_reference_csv(prob, str(tmp_path / "sim.csv"))
    r = subprocess.run([exe, ds, "--frontend"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    variant = 1 if int(subprocess.run(["g++", "-dumpversion"], capture_output=True, text=True).stdout.split(".")[0]) >= 11 else 0
    poses0, points0, init, _ = frontend.compute_initial_guess(prob.camera, prob.num_poses, prob.num_points, prob.obs_pose, prob.obs_point,
                                                               prob.obs_uvd, prob.poses_gt[0], variant=variant, ransac=_oracle_ransac)
    sel = init[prob.obs_point]
    op = orc.OracleProblem(prob.camera, poses0, points0, prob.obs_pose[sel], prob.obs_point[sel], prob.obs_uvd[sel], prob.stiffness())
    s2, _ = op.solve(orc.driver_options(num_threads=2))
    report = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")][0]
    assert "Termination: CONVERGENCE" in report
    assert float(report.split("Final cost: ")[1].split(",")[0]) == pytest.approx(s2.final_cost, rel=1e-5)
    out = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(out - op.poses).max() < 1e-5
    assert np.abs(out[:, :3] - prob.poses_gt[:, :3]).max() < 0.05


def test_phong_front_end_and_solve():
    """DatasetProblemPhong::compute_initial_guess -> the driver's solve (free shared blocks, bounds, SUBSPACE_DOGLEG)."""
    prob, ph = synth.make_phong_problem(30, 2400, obs_var=(0.04, 0.04, 0.04))
    mat_obs = ph.material_of_point[prob.obs_point]
    out = frontend.compute_initial_guess_phong(prob.camera, prob.num_poses, prob.num_points, len(ph.texture), prob.obs_pose, prob.obs_point,
                                               mat_obs, prob.obs_uvd, ph.intensity, ph.normal_obs, prob.poses_gt[0],
                                               reference_material_indexing=False)
    poses, positions, normals, init, mat_v, phong, texture, stats = out
    assert init.mean() > 0.95 and np.array_equal(mat_v[init], ph.material_of_point[init])
    np.testing.assert_allclose(texture, ph.texture_ref_init)              # the generator restates the same median
    assert np.all(phong == np.array([0.0, 0.0, 1.0]))
    assert np.abs(np.linalg.norm(normals[init], axis=1) - 1).max() < 0.05  # observed normals carry noise, Plus renormalises
    assert np.abs(poses[:, :3] - prob.poses_gt[:, :3]).max() < 1.0
    # the reference indexes material_ids by the position in the match list (dataset_problem_phong.cpp:369-370)
    out_q = frontend.compute_initial_guess_phong(prob.camera, prob.num_poses, prob.num_points, len(ph.texture), prob.obs_pose, prob.obs_point,
                                                 mat_obs, prob.obs_uvd, ph.intensity, ph.normal_obs, prob.poses_gt[0])
    assert not np.array_equal(out_q[4][init], ph.material_of_point[init])
    sel = init[prob.obs_point]
    keep = np.nonzero(init)[0]
    remap = np.full(prob.num_points, -1)
    remap[keep] = np.arange(len(keep))
    n_unit = normals[keep] / np.linalg.norm(normals[keep], axis=1)[:, None]
    d = dict(normals=n_unit, intensity=ph.intensity[sel], normal_obs=ph.normal_obs[sel], phong=phong, texture=texture,
             material_of_point=mat_v[keep], light=ph.light_init, light_type=ph.light_type, int_stiffness=ph.int_stiffness,
             normal_stiffness=ph.normal_stiffness())
    ba = StereoBA(prob.camera, poses.copy(), positions[keep].copy(), prob.obs_pose[sel], remap[prob.obs_point[sel]].astype(np.uint32),
                  prob.obs_uvd[sel], prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1))
    assert s.termination_type == 0 and s.final_cost < 0.1 * s.initial_cost
    assert np.abs(ba.poses[:, :3] - prob.poses_gt[:, :3]).max() < 0.05
    np.testing.assert_allclose(ba.texture, ph.texture, atol=0.02)


def test_degenerate_inputs_are_rejected():
    cam = capi.Camera(**synth.KITTI_CAMERA)
    lib = capi.load()
    off = np.array([0, 2], dtype=np.uint32)
    pts = np.zeros((2, 3)) + 1.0
    smp = np.zeros(3, dtype=np.uint32)
    T = np.zeros(12)
    rc = lib.ssba_frontend_ransac(C.byref(cam), -1, 1, off.ctypes.data_as(_u32p), capi.dptr(pts), capi.dptr(pts),
                                  np.array([0, 1, 5], dtype=np.uint32).ctypes.data_as(_u32p), 1, 4.0, capi.dptr(T), None, None, None)
    assert rc == -1            # sample index outside the pair
    assert lib.ssba_ransac_samples(2, 10, 1, smp.ctypes.data_as(_u32p)) == -1      # three distinct indices need n >= 3


def test_cpp_driver_sliding_windows_with_its_own_front_end(tmp_path):
    """examples/dataset_vo_gpu --frontend --window N = main() of tests/dataset_vo.cpp:116-127: per window the initial
    guess from the state the previous window left, the solve with the window's first state constant, reset_points."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples()
    prob = _problem(14, 700, track_len=6, seed=5)
    W = 5
    ds, _, _ = synth.write_reference_csv(prob, str(tmp_path / "sim.csv"))
    r = subprocess.run([exe, ds, "--frontend", "--window", str(W)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    variant = 1 if int(subprocess.run(["g++", "-dumpversion"], capture_output=True, text=True).stdout.split(".")[0]) >= 11 else 0
    P = prob.num_poses
    poses = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0]), (P, 1))
    poses[0] = prob.poses_gt[0]
    for k1 in range(0, P - W + 1):
        sel = (prob.obs_pose >= k1) & (prob.obs_pose < k1 + W)
        st, pt, uvd = prob.obs_pose[sel] - k1, prob.obs_point[sel], prob.obs_uvd[sel]
        pw, xw, init, _ = frontend.compute_initial_guess(prob.camera, W, prob.num_points, st, pt, uvd, poses[k1], variant=variant,
                                                         ransac=_oracle_ransac)
        use = init[pt]
        op = orc.OracleProblem(prob.camera, pw, xw, st[use], pt[use], uvd[use], prob.stiffness())      # first state of the window constant
        op.solve(orc.driver_options(num_threads=2))
        poses[k1:k1 + W] = op.poses
    reports = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")]
    assert len(reports) == P - W + 1 and all("CONVERGENCE" in l for l in reports)
    out = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(out - poses).max() < 1e-5


@pytest.mark.parametrize("window", [0, 12])
def test_phong_cpp_driver_with_its_own_front_end(tmp_path, window):
    """examples/dataset_ba_phong_gpu --frontend [--window N] = main() of tests/dataset_ba_phong.cpp:296-327: the Phong
    initial guess (incl. the reference's material indexing), then solveWindow over the sliding windows."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples("dataset_ba_phong_gpu")
    prob, ph = synth.make_phong_problem(20, 1200, track_len=8, seed=7, obs_var=(0.04, 0.04, 0.04))
    files = synth.write_reference_phong_csv(prob, ph, str(tmp_path / "sim.csv"), shared="reference")
    r = subprocess.run([exe, files[0], "--frontend"] + (["--window", str(window)] if window else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    variant = 1 if int(subprocess.run(["g++", "-dumpversion"], capture_output=True, text=True).stdout.split(".")[0]) >= 11 else 0
    mat_obs = ph.material_of_point[prob.obs_point]
    poses, positions, normals, init, mat_v, phong, texture, _ = frontend.compute_initial_guess_phong(
        prob.camera, prob.num_poses, prob.num_points, len(ph.texture), prob.obs_pose, prob.obs_point, mat_obs, prob.obs_uvd, ph.intensity,
        ph.normal_obs, prob.poses_gt[0], variant=variant, ransac=_oracle_ransac)
    light = np.asarray(ph.as_oracle_dict("reference")["light"], dtype=float).copy()
    P, W = prob.num_poses, window or prob.num_poses
    kw = dict(num_threads=4, trust_region_strategy_type=1, dogleg_type=1)
    finals = []
    for k1 in range(0, P - W + 1):
        sel = (prob.obs_pose >= k1) & (prob.obs_pose < k1 + W) & init[prob.obs_point]
        pts = np.unique(prob.obs_point[sel])
        remap = np.full(prob.num_points, -1)
        remap[pts] = np.arange(len(pts))
        d = dict(normals=normals[pts], intensity=ph.intensity[sel], normal_obs=ph.normal_obs[sel], phong=phong, texture=texture,
                 material_of_point=mat_v[pts], light=light, light_type=ph.light_type, int_stiffness=ph.int_stiffness,
                 normal_stiffness=ph.normal_stiffness())
        op = orc.OracleProblem(prob.camera, poses[k1:k1 + W], positions[pts], prob.obs_pose[sel] - k1, remap[prob.obs_point[sel]].astype(np.uint32),
                               prob.obs_uvd[sel], prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
        s2, _ = op.solve(orc.driver_options(**kw))
        finals.append(s2.final_cost)
        poses[k1:k1 + W], positions[pts], normals[pts] = op.poses, op.points, op.normals
        phong, texture, light = op.phong.copy(), op.texture.copy(), op.light.copy()
    reports = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")]
    assert len(reports) == len(finals)
    got = [float(l.split("Final cost: ")[1].split(",")[0]) for l in reports]
    np.testing.assert_allclose(got, finals, rtol=2e-4)
    out = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(out - poses).max() < 1e-4
    assert np.abs(out[:, :3] - prob.poses_gt[:, :3]).max() < 0.1


@pytest.mark.parametrize("outliers", [0.0, 0.2])
def test_device_vo_front_end_matches_the_oracle_backed_pipeline(outliers):
    """ssba_frontend_vo: matching, triangulation, RANSAC, pose chaining and map initialisation on the device (nothing of
    compute_initial_guess is left to the host) against the same pipeline with the oracle's RANSAC."""
    prob = _problem(40, 3000, outlier_fraction=outliers)
    args = (prob.camera, prob.num_poses, prob.num_points, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.poses_gt[0])
    poses, points, init, stats = frontend.compute_initial_guess_device(*args)
    poses2, points2, init2, stats2 = frontend.compute_initial_guess(*args, ransac=_oracle_ransac)
    assert stats["matches"] == stats2["matches"] and stats["inliers"] == stats2["inliers"]
    np.testing.assert_allclose(poses, poses2, rtol=1e-6, atol=1e-7)
    assert np.array_equal(init, init2)
    np.testing.assert_allclose(points[init], points2[init], rtol=1e-6, atol=1e-6)
    assert stats["device_s"] > 0

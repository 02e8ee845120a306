"""Row N4 on the GPU: unary pose residual blocks (pose prior, sun sensor) of tests/dataset_vo_sun.cpp:80-124 through the
C ABI against the oracle: LM step, whole solves with the driver's SUBSPACE_DOGLEG setting, Huber on the sun blocks,
no constant pose (the prior anchors the window)."""
import numpy as np
import pytest

from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc
from test_gpu_edge_cases import assert_fixed_count_parity
from test_oracle_pose_factors import _sun_problem

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _pair(prob, factors):
    none_const = np.zeros(prob.num_poses, dtype=np.uint8)
    ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                  pose_const=none_const, pose_factors=factors)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                           pose_const=none_const, pose_factors=factors)
    return ba, op


@pytest.mark.parametrize("huber", [0.0, 0.5])
@pytest.mark.parametrize("radius", [1e4, 20.0])
def test_lm_step_with_pose_factors_matches_oracle(huber, radius):
    prob, factors = _sun_problem(huber=huber)
    ba, op = _pair(prob, factors)
    S, rhs, dp, dl, mcc = ba.lm_step(radius)
    S2, rhs2, _ = op.reduced_system(radius)
    dp2, dl2, mcc2 = op.lm_step(radius)
    assert _rel(S, S2) < 1e-9 and _rel(rhs, rhs2) < 1e-9
    assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7
    assert mcc == pytest.approx(mcc2, rel=1e-8)
    assert ba.evaluate()[0] == pytest.approx(op.cost(), rel=1e-12)


@pytest.mark.parametrize("strategy", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("huber", [0.0, 0.5])
def test_sun_aided_solve_matches_oracle(strategy, huber):
    prob, factors = _sun_problem(P=30, L=1500, seed=5, huber=huber)
    ba, op = _pair(prob, factors)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-7)
    # the end point at the north-star bar: the same solve cut at a fixed iteration count (fresh handles)
    assert_fixed_count_parity(*_pair(prob, factors), 12, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    assert np.abs(ba.poses - op.poses).max() < 1e-4
    assert np.abs(ba.poses[0] - prob.poses_init[0]).max() < 0.02       # the prior holds the first pose


@pytest.mark.parametrize("P", [8, 30])
def test_pose_covariance_block_matches_dense_inverse(P):
    """The prior for the next window (dataset_vo_sun.cpp:159-183): covariance of pose k1+1 in tangent space after the solve."""
    prob, factors = _sun_problem(P=P, L=60 * P, seed=7)
    ba, op = _pair(prob, factors)
    ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    op.poses[:], op.points[:] = ba.poses, ba.points
    S, rhs, free_idx = op.reduced_system(1e300)          # undamped reduced camera system at the solution
    Sinv = np.linalg.inv(S)
    Sg = ba.lm_step(1e300)[0]                             # the device's own assembly of the same system
    Sginv = np.linalg.inv(Sg)
    assert _rel(Sg, S) < 1e-6        # undamped landmark blocks: the Schur complement cancels ~8 digits
    for k in (1, P // 2, P - 1):
        f = int(free_idx[k])
        cov = ba.pose_covariance(k)
        # the gauge is only held by the prior: cond(S) ~ 1e9, so 1e-12 differences of the two assemblies show at 1e-4
        assert _rel(cov, Sginv[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]) < 1e-6
        assert _rel(cov, Sinv[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]) < 1e-3
        assert np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)
    # without the prior and the sun blocks nothing fixes the gauge: the reduced system is singular
    ba2 = StereoBA(prob.camera, ba.poses.copy(), ba.points.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                   pose_const=np.zeros(P, dtype=np.uint8))
    try:
        c = ba2.pose_covariance(1)
        assert np.abs(c).max() > 1e6         # numerically singular: enormous variances if the factorisation survives
    except capi.SsbaError:
        pass


def test_pose_factor_restrictions():
    prob, factors = _sun_problem()
    with pytest.raises(capi.SsbaError):           # default pose_const holds pose 0 constant: the prior would sit on it
        StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd,
                 prob.stiffness(), pose_factors=factors)


# ---- RelativePoseErrorAutomatic blocks (relative_pose_error.hpp, tests/blowup_test.cpp): pose-pose couplings ----
from test_oracle_pose_factors import _odometry_factors


@pytest.mark.parametrize("huber", [0.0, 0.05])
@pytest.mark.parametrize("radius", [1e4, 20.0])
def test_lm_step_with_relative_pose_blocks_matches_oracle(huber, radius):
    prob = synth.make_problem(9, 300, track_len=5, seed=6)
    factors = _odometry_factors(prob, huber=huber)
    ba, op = _pair(prob, factors)
    assert ba.stats().general_structure == 1          # the loop block couples the first and the last pose
    S, rhs, dp, dl, mcc = ba.lm_step(radius)
    S2, rhs2, _ = op.reduced_system(radius)
    dp2, dl2, mcc2 = op.lm_step(radius)
    assert _rel(S, S2) < 1e-9 and _rel(rhs, rhs2) < 1e-9
    assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7
    assert mcc == pytest.approx(mcc2, rel=1e-8)
    assert ba.evaluate()[0] == pytest.approx(op.cost(), rel=1e-12)


@pytest.mark.parametrize("strategy", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("huber", [0.0, 0.05])
def test_solve_with_odometry_and_loop_blocks_matches_oracle(strategy, huber):
    prob = synth.make_problem(20, 700, track_len=6, seed=11, pose_sigma=(0.1, 0.02))
    factors = _odometry_factors(prob, huber=huber)
    ba, op = _pair(prob, factors)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-7)
    assert_fixed_count_parity(*_pair(prob, factors), 12, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    assert np.abs(ba.poses - op.poses).max() < 1e-4


def test_relative_pose_block_on_a_constant_pose_and_pose_graph_only():
    prob = synth.make_problem(8, 200, track_len=4, seed=2)
    factors = _odometry_factors(prob, loop=True)[1:]          # no prior: pose 0 is held constant instead
    const = np.zeros(8, np.uint8)
    const[0] = 1
    ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                  pose_const=const, pose_factors=factors)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                           pose_const=const, pose_factors=factors)
    dp, dl, mcc = ba.lm_step(100.0)[2:]
    dp2, dl2, mcc2 = op.lm_step(100.0)
    assert _rel(dp, dp2) < 1e-7 and mcc == pytest.approx(mcc2, rel=1e-8)
    s, _ = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    s2, _ = op.solve(orc.driver_options(num_threads=2))
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6) and np.array_equal(ba.poses[0], prob.poses_init[0])
    # a pose graph without any stereo block (tests/blowup_test.cpp)
    none = (np.zeros((0, 3)), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 3)), np.eye(3))
    pg = _odometry_factors(prob)
    ba = StereoBA(prob.camera, prob.poses_init.copy(), *none, pose_const=np.zeros(8, np.uint8), pose_factors=pg)
    op = orc.OracleProblem(prob.camera, prob.poses_init, *none, pose_const=np.zeros(8, np.uint8), pose_factors=pg)
    s, _ = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    s2, _ = op.solve(orc.driver_options(num_threads=1))
    assert s.termination_type == s2.termination_type == 0 and s.num_iterations == s2.num_iterations
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6, abs=1e-18)
    assert np.abs(ba.poses - op.poses).max() < 1e-7


def test_blowup_driver_through_the_ceres_shim():
    """examples/blowup_test_gpu: two-state windows of one relative-pose block + the prior from the previous window's
    covariance (tests/blowup_test.cpp:55-125) -- the trace of the covariance grows along the chain, the poses follow
    the measurement."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples("blowup_test_gpu")
    r = subprocess.run([exe, "8", "0.1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = np.array([[float(x) for x in l.split()] for l in r.stdout.splitlines() if l.strip()])
    assert rows.shape == (8, 5)
    yaw, T = 0.05, np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0])
    meas = np.array([0, 0, -1.0, np.cos(yaw), 0, np.sin(yaw), 0, 1, 0, -np.sin(yaw), 0, np.cos(yaw)])
    for k in range(1, 8):                       # with a stiff prior the optimum is the chained measurement
        R = meas[3:].reshape(3, 3)
        T = np.concatenate([R @ T[:3] + meas[:3], (R @ T[3:].reshape(3, 3)).ravel()])
        np.testing.assert_allclose(rows[k, 1:4], T[:3], atol=1e-6)
    assert np.all(np.diff(rows[:, 4]) > 0)      # uncertainty accumulates
    assert rows[1, 4] == pytest.approx(6 * 0.1 ** 2 + 6e-12, rel=1e-3)      # first window: measurement noise on top of the prior


def test_sun_window_through_the_python_api_mirror():
    """Reads like solveWindow of tests/dataset_vo_sun.cpp:28-185: per-point stereo stiffness, sun blocks with HuberLoss,
    the pose prior, SUBSPACE_DOGLEG, then ceres::Covariance of the second state."""
    from ceres_slam_amd import ceres_api as ceres
    from test_gpu_general_structure import _per_point_stiffness
    prob, factors = _sun_problem(P=6, L=400, seed=4, huber=0.5)
    S_obs = _per_point_stiffness(prob, seed=2)
    poses, points = prob.poses_init.copy(), prob.points_init.copy()
    camera = ceres.StereoCamera(**prob.camera)
    problem = ceres.Problem()
    se3_perturbation = ceres.SE3Perturbation.Create()
    for i in range(prob.num_obs):
        k, j = int(prob.obs_pose[i]), int(prob.obs_point[i])
        problem.AddResidualBlock(ceres.StereoReprojectionErrorAutomatic.Create(camera, prob.obs_uvd[i], S_obs[i]), None, poses[k], points[j])
    for f in factors:
        if f["type"] == 1:
            d = np.asarray(f["data"])
            cost = ceres.SunSensorErrorAutomatic.Create(d[:3], d[3:6], np.asarray(f["stiffness"]).reshape(2, 2), d[6], d[7])
            problem.AddResidualBlock(cost, ceres.HuberLoss(0.5), poses[f["pose"]])
        else:
            problem.AddResidualBlock(ceres.PoseErrorAutomatic.Create(f["data"], np.asarray(f["stiffness"]).reshape(6, 6)), None, poses[f["pose"]])
    for k in range(prob.num_poses):
        problem.SetParameterization(poses[k], se3_perturbation)
    options = ceres.SolverOptions()
    options.max_num_iterations, options.use_nonmonotonic_steps = 1000, 1
    options.trust_region_strategy_type, options.dogleg_type = 1, 1
    summary = ceres.SolverSummary()
    ceres.Solve(options, problem, summary)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, S_obs,
                           pose_const=np.zeros(prob.num_poses, np.uint8), pose_factors=factors)
    s2, _ = op.solve(orc.driver_options(num_threads=2, trust_region_strategy_type=1, dogleg_type=1))
    assert summary.termination_type == ceres.CONVERGENCE
    assert summary.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    assert np.abs(poses - op.poses).max() < 1e-5
    covariance = ceres.Covariance()
    assert covariance.Compute([(poses[1], poses[1])], problem)
    cov = np.zeros((6, 6))
    assert covariance.GetCovarianceBlockInTangentSpace(poses[1], poses[1], cov)
    Sred, _, free_idx = op.reduced_system(1e300)
    f = int(free_idx[1])
    ref = np.linalg.inv(Sred)[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]
    assert np.abs(cov - ref).max() / np.abs(ref).max() < 2e-2 and np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)
    assert not covariance.GetCovarianceBlockInTangentSpace(poses[2], poses[2], cov)


def test_sun_window_matches_golden():
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sun_window.json")))
    prob, factors = _sun_problem(P=8, L=400, seed=4, huber=0.5)
    ba, _ = _pair(prob, factors)
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1))
    assert s.num_iterations == gold["num_iterations"] and log["step_is_successful"].tolist() == gold["step_is_successful"]
    np.testing.assert_allclose(log["cost"], gold["cost"], rtol=1e-7)
    np.testing.assert_allclose(ba.poses, gold["poses"], atol=1e-5)
    cov = ba.pose_covariance(1)
    ref = np.asarray(gold["covariance_pose1"])
    assert np.abs(cov - ref).max() / np.abs(ref).max() < 2e-2

"""More GPU parity cases through the C ABI: option variations, block-count edge cases of the
block-cyclic-reduction solver, full 3x3 stiffness, several constant poses, failure handling."""
import numpy as np
import pytest

from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DRIVER = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)


def _solve_both(prob, gpu_kw=None, orc_kw=None, opts=None, stiffness=None, pose_const=None, huber_a=0.0):
    S = prob.stiffness() if stiffness is None else stiffness
    o = dict(DRIVER)
    o.update(opts or {})
    ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, S,
                  pose_const=pose_const, huber_a=huber_a)
    s, log = ba.solve(capi.default_options(**o))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, S,
                           pose_const=pose_const, huber_a=huber_a)
    s2, log2 = op.solve(orc.default_options(num_threads=2, **o))
    return ba, s, log, op, s2, log2


def _assert_same_solve(ba, s, log, op, s2, log2, pose_tol=1e-6):
    assert s.termination_type == s2.termination_type
    assert s.num_iterations == s2.num_iterations
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=1e-8)
    # rejected candidates far outside the trust region (costs ~1e9) are ill-conditioned: summation order shows
    np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-6)
    assert abs(s.final_cost - s2.final_cost) <= 1e-6 * max(s2.final_cost, 1e-300)
    assert np.abs(ba.poses - op.poses).max() < pose_tol


def assert_fixed_count_parity(ba, op, K, cost_rtol=1e-7, threads=4, **kw):
    """Both sides cut at the same iteration count K and compared there at the north-star bar: identical accept / reject
    sequence, cost trace of the accepted iterates, final cost to 1e-6 relative.  (A solve that runs to convergence through a
    long flat tail stops at a rounding-sensitive iteration; its end point is not a parity statement.)  Solves from the
    handles' current state: call it on freshly built pairs."""
    kw = dict(kw, max_num_iterations=K)
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=threads, **kw))
    assert s.num_iterations == s2.num_iterations
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=cost_rtol)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    return s, log, s2, log2


@pytest.mark.parametrize("lighting", [False, True])
def test_check_inside_the_schur_launch_equals_the_check_launch(monkeypatch, lighting):
    """k_check's work done by one extra work-group of the Schur launch (the default on the single-GPU LM path) against the
    launch of its own (SSBA_CHECK_LAUNCH=1): same iteration count, accept / reject sequence and termination, cost trace to
    1e-12 (only the order of two sums over the poses differs), poses to 1e-10."""
    if lighting:
        prob, ph = synth.make_phong_problem(40, 1600, num_materials=3, seed=4)
        kw = dict(lighting=ph.as_oracle_dict("truth"))
    else:
        prob, kw = synth.make_problem(60, 2400, track_len=9, seed=12), {}
    out = []
    for env in ("0", "1"):
        monkeypatch.setenv("SSBA_CHECK_LAUNCH", env)
        ba = StereoBA.from_synth(prob, **kw)
        s, log = ba.solve(capi.default_options(**DRIVER))
        out.append((s, log, ba.poses.copy()))
        ba.close()
    (s0, l0, p0), (s1, l1, p1) = out
    assert s0.num_iterations == s1.num_iterations and s0.termination_type == s1.termination_type
    assert l0["step_is_successful"].tolist() == l1["step_is_successful"].tolist()
    np.testing.assert_allclose(l0["cost"], l1["cost"], rtol=1e-12)
    np.testing.assert_allclose(l0["gradient_max_norm"], l1["gradient_max_norm"], rtol=1e-12)
    assert np.abs(p0 - p1).max() < 1e-10


def test_many_poses_few_landmarks():
    """Fewer landmark groups than pose blocks: the copies of the best iterate that ride in the landmark kernels (twelve
    doubles per pose spread over the lanes of the evaluation launch) must still cover every pose -- 150 poses with 10
    groups of 64 landmarks are 1 800 doubles for 1 280 lanes."""
    prob = synth.make_problem(150, 600, track_len=8, seed=5)
    ba, s, log, op, s2, log2 = _solve_both(prob)
    assert ba.stats().general_structure == 0
    _assert_same_solve(ba, s, log, op, s2, log2)


@pytest.mark.parametrize("num_poses", [2, 3, 12, 13, 14, 25, 26, 37, 49, 61, 97, 150])
def test_bcr_block_count_edges(num_poses):
    """1, 2, 3, ... super-blocks incl. padded last blocks and odd/even level sizes."""
    prob = synth.make_problem(num_poses, 40 * num_poses, track_len=12, seed=num_poses)
    _assert_same_solve(*_solve_both(prob))


@pytest.mark.parametrize("num_poses", [1530, 1600, 3100])
def test_cyclic_reduction_levels_below_the_parallel_plan(num_poses):
    """More than 128 super-blocks: plain cyclic-reduction levels first (1 for 1 600 poses, 2 for 3 100), the parallel
    cyclic reduction takes over at <= 128 blocks; 1 530 poses = 128 blocks exactly."""
    prob = synth.make_problem(num_poses, 10 * num_poses, track_len=12, seed=7)
    ba, s, log, op, s2, log2 = _solve_both(prob)
    assert ba.stats().pcr_blocks == {1530: 128, 1600: 67, 3100: 65}[num_poses]
    _assert_same_solve(ba, s, log, op, s2, log2)


def test_parallel_and_plain_cyclic_reduction_agree(monkeypatch):
    prob = synth.make_problem(300, 9000, track_len=12, seed=3)
    ba = StereoBA.from_synth(prob)
    s, log = ba.solve(capi.default_options(**DRIVER))
    monkeypatch.setenv("SSBA_NO_PCR", "1")
    ba2 = StereoBA.from_synth(prob)
    s2, log2 = ba2.solve(capi.default_options(**DRIVER))
    assert ba.stats().pcr_blocks == 25 and ba2.stats().pcr_blocks == 0
    assert s.num_iterations == s2.num_iterations
    np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-10)
    assert np.abs(ba.poses - ba2.poses).max() < 1e-9


@pytest.mark.parametrize("num_poses", [14, 26, 300, 1600])
def test_fused_and_two_launch_steps_of_the_parallel_plan_agree(monkeypatch, num_poses):
    """One launch per step of the parallel cyclic reduction (the factorisation of a block also forms the Gram products the
    next step assembles its operands from: ssba_bcr_mfma.hip, PcrFused) against factor + reduce launches per step
    (SSBA_NO_PCR_FUSED=1): 2, 3, 25 super-blocks, and 67 below a plain level; both against the oracle."""
    prob = synth.make_problem(num_poses, 30 * num_poses, track_len=12, seed=3)
    ba, s, log, op, s0, log0 = _solve_both(prob)
    _assert_same_solve(ba, s, log, op, s0, log0)
    monkeypatch.setenv("SSBA_NO_PCR_FUSED", "1")
    ba2 = StereoBA.from_synth(prob)
    s2, log2 = ba2.solve(capi.default_options(**DRIVER))
    assert s.num_iterations == s2.num_iterations
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=1e-10)
    np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-7)      # rejected candidates far outside the trust region (costs ~1e10) are ill-conditioned
    assert np.abs(ba.poses - ba2.poses).max() < 1e-9


def test_full_3x3_stiffness_matrix():
    # the sun driver builds full covariances (tests/dataset_vo_sun.cpp:57-59); the stereo functor
    # takes any 3x3 stiffness (stereo_reprojection_error.hpp:49-50)
    prob = synth.make_problem(20, 600, track_len=8, seed=4)
    rng = np.random.default_rng(0)
    A = rng.normal(size=(3, 3))
    cov = A @ A.T + 3 * np.eye(3)
    w, V = np.linalg.eigh(cov)
    S = V @ np.diag(w ** -0.5) @ V.T
    _assert_same_solve(*_solve_both(prob, stiffness=S))


def test_several_constant_poses_and_nothing_free():
    prob = synth.make_problem(15, 500, track_len=6, seed=5)
    const = np.zeros(15, bool)
    const[[0, 7, 14]] = True
    ba, s, log, op, s2, log2 = _solve_both(prob, pose_const=const)
    _assert_same_solve(ba, s, log, op, s2, log2)
    for k in (0, 7, 14):
        assert np.array_equal(ba.poses[k], prob.poses_init[k])
    # all poses constant: pure triangulation refinement, the reduced system is empty
    ba, s, log, op, s2, log2 = _solve_both(prob, pose_const=np.ones(15, bool))
    _assert_same_solve(ba, s, log, op, s2, log2)
    assert np.array_equal(ba.poses, prob.poses_init)


@pytest.mark.parametrize("opts", [dict(use_nonmonotonic_steps=0), dict(jacobi_scaling=0), dict(initial_trust_region_radius=1.0),
                                  dict(max_num_iterations=3), dict(max_num_iterations=0), dict(function_tolerance=1e-12),
                                  dict(min_relative_decrease=0.5)])
def test_option_variations(opts):
    prob = synth.make_problem(16, 500, track_len=8, seed=6)
    ba, s, log, op, s2, log2 = _solve_both(prob, opts=opts)
    _assert_same_solve(ba, s, log, op, s2, log2, pose_tol=1e-5)
    if opts.get("max_num_iterations") == 0:
        assert s.termination_type == 1 and s.num_iterations == 1      # NO_CONVERGENCE after iteration 0
        assert np.array_equal(ba.poses, prob.poses_init)


def test_huber_medium_scale():
    prob = synth.make_problem(120, 8000, track_len=12, seed=8, outlier_fraction=0.3)
    ba = StereoBA.from_synth(prob, huber_a=1.345)
    op = orc.OracleProblem.from_synth(prob, huber_a=1.345)
    assert_fixed_count_parity(ba, op, 20, cost_rtol=1e-8, threads=2, use_nonmonotonic_steps=1)
    assert np.abs(ba.poses - op.poses).max() < 1e-6


def test_point_behind_camera_gives_rejected_steps_not_garbage():
    # the reference does not guard z <= 0 (stereo_camera.hpp:79): inf/NaN must end in rejected or
    # invalid steps / FAILURE, never in a silently accepted non-finite state
    prob = synth.make_problem(6, 60, track_len=4, seed=9)
    pts = prob.points_init.copy()
    pts[3] = prob.points_init[3] * 0 + np.array([0.0, 0.0, -1e3])     # far behind every camera
    ba = StereoBA(prob.camera, prob.poses_init.copy(), pts, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness())
    s, log = ba.solve(capi.default_options(max_num_iterations=30, use_nonmonotonic_steps=1))
    assert np.all(np.isfinite(ba.poses)) and np.all(np.isfinite(ba.points))
    assert np.all(np.isfinite(log["cost"][log["step_is_successful"] == 1]))


def test_handle_reuse_solve_twice_and_set_huber_after_finalize():
    prob = synth.make_problem(14, 400, track_len=6, seed=10)
    ba = StereoBA.from_synth(prob)
    s1, _ = ba.solve(capi.default_options(**DRIVER))
    # second solve starts from the optimum: converges immediately, parameters barely move
    before = ba.poses.copy()
    s2, _ = ba.solve(capi.default_options(**DRIVER))
    assert s2.num_iterations <= 3 and np.abs(ba.poses - before).max() < 1e-6
    assert s2.initial_cost == pytest.approx(s1.final_cost, rel=1e-9)
    # changing the loss on a finalized handle takes effect (and invalidates the captured graph)
    ba.poses[:] = prob.poses_init
    ba.points[:] = prob.points_init
    capi.check(ba.lib.ssba_set_huber_loss(ba.h, 1.345), "ssba_set_huber_loss")
    s3, _ = ba.solve(capi.default_options(**DRIVER))
    op = orc.OracleProblem.from_synth(prob, huber_a=1.345)
    s4, _ = op.solve(orc.driver_options(num_threads=2))
    assert s3.final_cost == pytest.approx(s4.final_cost, rel=1e-6)


def test_handle_reuse_with_other_options_recaptures_the_graph():
    """A second ssba_solve on the same handle with a larger iteration cap reallocates the iteration log; the captured
    hipGraph has the old log pointers baked in and must be dropped (it used to report final_cost = 0 and an all-zero
    log).  Same for solve -> evaluate -> solve with a small cap, and for LM -> dogleg on one handle."""
    prob = synth.make_problem(14, 400, track_len=6, seed=10)
    op = orc.OracleProblem.from_synth(prob)
    s_ref, log_ref = op.solve(orc.driver_options(num_threads=2))
    ba = StereoBA.from_synth(prob)
    s1, log1 = ba.solve(capi.default_options(max_num_iterations=5, use_nonmonotonic_steps=1))
    assert s1.num_iterations <= 6
    np.testing.assert_allclose(log1["cost"], log_ref["cost"][:len(log1["cost"])], rtol=1e-9)
    ba.poses[:] = prob.poses_init
    ba.points[:] = prob.points_init
    s2, log2 = ba.solve(capi.default_options(**DRIVER))          # cap 1000: the log is reallocated
    assert s2.num_iterations == s_ref.num_iterations
    np.testing.assert_allclose(log2["cost"], log_ref["cost"], rtol=1e-9)
    assert s2.final_cost == pytest.approx(s_ref.final_cost, rel=1e-6) and s2.final_cost > 0
    # solve (small cap) -> evaluate (needs 16 log entries) -> solve
    ba2 = StereoBA.from_synth(prob)
    ba2.solve(capi.default_options(max_num_iterations=3, use_nonmonotonic_steps=1))
    ba2.poses[:] = prob.poses_init
    ba2.points[:] = prob.points_init
    cost0 = ba2.evaluate()[0]
    assert cost0 == pytest.approx(s_ref.initial_cost, rel=1e-12)
    s3, log3 = ba2.solve(capi.default_options(max_num_iterations=4, use_nonmonotonic_steps=1))
    np.testing.assert_allclose(log3["cost"], log_ref["cost"][:len(log3["cost"])], rtol=1e-9)
    # LM graph, then a dogleg solve on the same handle
    ba2.poses[:] = prob.poses_init
    ba2.points[:] = prob.points_init
    s4, _ = ba2.solve(capi.default_options(trust_region_strategy_type=1, **DRIVER))
    s5, _ = op.__class__.from_synth(prob).solve(orc.driver_options(trust_region_strategy_type=1, num_threads=2))
    assert s4.num_iterations == s5.num_iterations
    assert s4.final_cost == pytest.approx(s5.final_cost, rel=1e-6)


def test_batched_graph_is_dropped_when_the_strategy_changes():
    """ssba_solve_step(n >= 10) as the FIRST call of a handle captures only the ten-iteration graph (the single-iteration
    one stays empty); a following solve with another trust-region strategy on the same handle must not replay it (it used
    to: the LM kernel sequence ran under DOGLEG options)."""
    prob = synth.make_problem(14, 400, track_len=6, seed=10)
    ba = StereoBA.from_synth(prob)
    logs = {}
    for name, kw in (("lm", {}), ("dogleg", dict(trust_region_strategy_type=1, dogleg_type=1)), ("lm2", {})):
        ba.poses[:] = prob.poses_init
        ba.points[:] = prob.points_init
        ba.solve_begin(capi.default_options(**dict(DRIVER, **kw)))
        ba.step(10)
        ba.step(10)
        ba.solve_end()
        logs[name] = ba.iteration_log()
        op = orc.OracleProblem.from_synth(prob)
        s2, log2 = op.solve(orc.driver_options(num_threads=2, **kw))
        n = min(len(logs[name]["cost"]), len(log2["cost"]))
        assert n >= 5
        assert logs[name]["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
        np.testing.assert_allclose(logs[name]["cost"][:n], log2["cost"][:n], rtol=1e-8)
        np.testing.assert_allclose(logs[name]["trust_region_radius"][:n], log2["trust_region_radius"][:n], rtol=1e-6)
    assert logs["lm"]["cost"].tolist() == logs["lm2"]["cost"].tolist()


@pytest.mark.parametrize("dogleg_type", [0, 1])
@pytest.mark.parametrize("huber_a", [0.0, 1.345])
@pytest.mark.parametrize("size", [(16, 500, 8), (50, 2000, 12)])
def test_dogleg_strategy_matches_oracle(size, huber_a, dogleg_type):
    """trust_region_strategy_type = DOGLEG, TRADITIONAL_DOGLEG and SUBSPACE_DOGLEG
    (tests/dataset_vo_sun.cpp:142-143)."""
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=12)
    ba, s, log, op, s2, log2 = _solve_both(prob, opts=dict(trust_region_strategy_type=1, dogleg_type=dogleg_type), huber_a=huber_a)
    _assert_same_solve(ba, s, log, op, s2, log2)
    np.testing.assert_allclose(log["trust_region_radius"], log2["trust_region_radius"], rtol=1e-6)
    if huber_a == 0.0:   # same minimum as Levenberg-Marquardt (the robustified runs stop on different flat tails)
        ba_lm, s_lm, *_ = _solve_both(prob)
        assert s.final_cost == pytest.approx(s_lm.final_cost, rel=1e-4)       # two different minimisers, each stopping on its own tolerance: not a parity statement

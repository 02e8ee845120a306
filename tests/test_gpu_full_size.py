"""BASELINE.json configs at their stated sizes against the CPU oracle: config 3 (C2 + Phong lighting
+ normal blocks), config 5's shape (C2 + 30 % outlier observations + HuberLoss) and config 4's problem (10 000 poses /
1 M landmarks / 12 M observations) -- on one GPU, and sharded over four ranks that share the one device of the test box.

Whole solves of these sizes end in long flat tails where the stop iteration is rounding-sensitive, so both sides are
cut at the same iteration count and compared there at the north-star bar (final cost 1e-6 relative, trajectory 1e-6);
the converged C2 solve is the job of tests/test_gpu_parity.py."""
import numpy as np
import pytest

from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc
from test_sharding import _run_ranks

pytestmark = pytest.mark.gpu


def _compare(s, log, s2, log2, ba, op, cost_rtol=1e-8, pose_tol=1e-6):
    assert s.num_iterations == s2.num_iterations
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True          # the candidate cost of a rejected step is a far-off point: compared through accept / reject only
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=cost_rtol)
    assert s.initial_cost == pytest.approx(s2.initial_cost, rel=1e-12)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)           # north-star bar
    assert np.abs(ba.poses - op.poses).max() < pose_tol


def test_c3_full_size_lighting_solve_matches_cpu_oracle():
    """configs[2]: 1 000 poses / 100 000 landmarks, stereo + intensity + normal blocks (7 residual rows per observation,
    6-D landmark blocks), shared light / material / texture blocks constant as in stage 1 of the reference driver."""
    K = 12
    prob, ph = synth.make_phong_problem(*synth.CONFIGS["C2"])
    d = ph.as_oracle_dict("truth")
    ba = StereoBA.from_synth(prob, lighting=d)
    s, log = ba.solve(capi.default_options(max_num_iterations=K, use_nonmonotonic_steps=1))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, max_num_iterations=K))
    _compare(s, log, s2, log2, ba, op)
    assert np.abs(ba.normals - op.normals).max() < 1e-6
    assert np.abs(np.linalg.norm(ba.normals, axis=1) - 1).max() < 1e-12


def test_c3_full_size_lighting_solve_converges_like_the_cpu_oracle():
    """configs[2] run to CONVERGENCE on both sides (tests/dataset_ba_phong.cpp:250-252 solves until Ceres stops): the same
    iteration count and accept / reject sequence, final cost within 1e-6 relative (the north-star bar), trajectory and
    normals within 1e-6 -- what test_c2_full_size_solve_matches_cpu_oracle (tests/test_gpu_parity.py) is for configs[1]."""
    prob, ph = synth.make_phong_problem(*synth.CONFIGS["C2"])
    d = ph.as_oracle_dict("truth")
    opts = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)
    ba = StereoBA.from_synth(prob, lighting=d)
    s, log = ba.solve(capi.default_options(**opts))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d)
    s2, log2 = op.solve(orc.driver_options(num_threads=16))
    assert s.termination_type == s2.termination_type == 0
    _compare(s, log, s2, log2, ba, op, cost_rtol=1e-7)
    assert np.abs(ba.normals - op.normals).max() < 1e-6


def test_c4_full_size_single_gpu_converges_like_the_cpu_oracle():
    """configs[3]'s problem (10 000 poses / 1 M landmarks / 12 M observations) on ONE GPU, run to convergence on both sides:
    same iteration count, accept / reject sequence, final cost 1e-6, trajectory 1e-6 (the oracle's solve takes about a minute
    on 16 host threads)."""
    prob = synth.make_config("C4")
    opts = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)
    ba = StereoBA.from_synth(prob)
    s, log = ba.solve(capi.default_options(**opts))
    op = orc.OracleProblem.from_synth(prob)
    s2, log2 = op.solve(orc.driver_options(num_threads=16))
    assert s.termination_type == s2.termination_type == 0
    # (82 iterations on a chain of 10 000 poses: iteration count, accept / reject sequence, cost trace and final cost agree at
    # the bars above; the poses of the converged end of the chain agree to 4e-6 -- r04 -- where the shorter configurations reach 1e-6)
    _compare(s, log, s2, log2, ba, op, cost_rtol=1e-7, pose_tol=1e-5)


def test_c3_full_size_in_the_phong_drivers_own_configuration():
    """configs[2] the way tests/dataset_ba_phong.cpp:84-87,143-181 configures it: SUBSPACE_DOGLEG, non-monotonic steps, light /
    material / texture blocks free, bounds on the material and texture blocks (projected Plus + Armijo line search -- on the
    device, csrc/ssba_linesearch.h), the reference's initial material values."""
    K = 10
    prob, ph = synth.make_phong_problem(*synth.CONFIGS["C2"])
    d = ph.as_oracle_dict("reference")
    kw = dict(max_num_iterations=K, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1)
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True)
    s, log = ba.solve(capi.default_options(**kw))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, **kw))
    assert s.num_iterations == s2.num_iterations
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=1e-6)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    assert s.num_line_search_steps == s2.num_line_search_steps and s.num_line_searches_on_device > 0
    assert np.abs(ba.poses - op.poses).max() < 1e-5
    assert np.all(ba.phong[:, :2] >= 0) and np.all(ba.phong[:, :2] <= 1) and np.all(ba.phong[:, 2] >= 1)
    assert np.all(ba.texture >= 0) and np.all(ba.texture <= 1)
    np.testing.assert_allclose(ba.texture, op.texture, rtol=1e-5, atol=1e-7)


def test_c5_shape_full_size_huber_outliers_matches_cpu_oracle():
    """configs[4]'s problem on one GPU: C2 with 30 % of the observations replaced by outliers, HuberLoss(1.345) on
    every stereo block."""
    K = 20
    prob = synth.make_config("C2", outlier_fraction=0.3)
    ba = StereoBA.from_synth(prob, huber_a=1.345)
    s, log = ba.solve(capi.default_options(max_num_iterations=K, use_nonmonotonic_steps=1))
    op = orc.OracleProblem.from_synth(prob, huber_a=1.345)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, max_num_iterations=K))
    _compare(s, log, s2, log2, ba, op)


def test_lt24_full_size_converges_like_the_cpu_oracle():
    """bench.py --config LT24 (600 poses / 60 000 landmarks / 24 observations per landmark: the wide reduced system of
    ssba_wide.hip), run to convergence on both sides: same iteration count, accept / reject sequence, final cost, trajectory."""
    prob = synth.make_config("LT24")
    ba = StereoBA.from_synth(prob)
    assert ba.stats().wide_superblocks == 25
    opts = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)
    s, log = ba.solve(capi.default_options(**opts))
    op = orc.OracleProblem.from_synth(prob)
    s2, log2 = op.solve(orc.driver_options(num_threads=16))
    assert s.termination_type == s2.termination_type == 0
    _compare(s, log, s2, log2, ba, op)


@pytest.mark.parametrize("wide", [True, False])
def test_general_path_beyond_4096_poses_matches_cpu_oracle(wide, monkeypatch):
    """Long tracks on 5 000 poses.  wide: 209 super-blocks of 24 poses (ssba_wide.hip: eight steps of parallel cyclic reduction).
    Not wide (SSBA_NO_WIDE=1): the blocked Cholesky of ssba_dense.hip -- the dense array of its reduced system (7 GB) is bounded
    by memory, not by a pose count: an iteration zero-fills and factors only its structurally non-zero tiles (r02 stopped at
    4 096 free poses)."""
    K = 5
    if not wide:
        monkeypatch.setenv("SSBA_NO_WIDE", "1")
    prob = synth.make_problem(5000, 40000, track_len=16, seed=31)
    ba = StereoBA.from_synth(prob)
    assert ba.stats().general_structure == 1 and ba.stats().num_free_poses == 4999
    assert (ba.stats().wide_superblocks > 0) == wide
    s, log = ba.solve(capi.default_options(max_num_iterations=K, use_nonmonotonic_steps=1))
    op = orc.OracleProblem.from_synth(prob)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, max_num_iterations=K))
    _compare(s, log, s2, log2, ba, op)


def test_c4_full_size_single_gpu_and_four_rank_partitioned(tmp_path):
    """configs[3]: 10 000 poses / 1 000 000 landmarks / 12 M observations.  (i) one GPU against the oracle's first
    iterations; (ii) the landmarks sharded over four ranks (all on the one device here, gloo exchange) with the
    partitioned reduced solve: the ranks agree bit for bit and follow the same path as the single-GPU solve."""
    K = 10
    size = (10000, 1000000, 12)
    prob = synth.make_problem(*size[:2], track_len=size[2], seed=21)
    ba = StereoBA.from_synth(prob)
    s, log = ba.solve(capi.default_options(max_num_iterations=K, use_nonmonotonic_steps=1))
    assert ba.stats().num_superblocks == 834
    op = orc.OracleProblem.from_synth(prob)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, max_num_iterations=K))
    _compare(s, log, s2, log2, ba, op, cost_rtol=1e-7)
    res = _run_ranks("gpu_part", str(tmp_path / "c4"), 4, size=size, timeout=900, extra_env={"SSBA_TEST_MAXIT": str(K)})
    for r in res:
        assert r["num_iterations"] == s.num_iterations
        assert r["accept"] == log["step_is_successful"].tolist()
        ok = np.asarray(log["step_is_successful"], dtype=bool)
        ok[0] = True
        np.testing.assert_allclose(np.asarray(r["cost"])[ok], log["cost"][ok], rtol=1e-8)
        assert r["final_cost"] == pytest.approx(s.final_cost, rel=1e-9)
        assert np.abs(np.asarray(r["poses"]) - ba.poses).max() < 1e-7
    for r in res[1:]:
        assert r["poses"] == res[0]["poses"]           # bit for bit across ranks


def test_c4_size_loop_closure_on_the_border_path_matches_cpu_oracle():
    """A loop closure at configs[3]'s size: 10 000 poses / 1 000 000 landmarks, the last states re-observe 300 landmarks
    of the first ones.  The general path stores the reduced system densely (<= 4 096 poses); here the three closing states
    ride as an 18-column border of the 834-block chain (plain cyclic-reduction levels + parallel top + multi-right-hand-side
    sweeps).  The oracle factors the same system with its profile Cholesky (18 long rows)."""
    K = 8
    prob = synth.add_loop_closure(synth.make_problem(10000, 1000000, track_len=12, seed=21), num_landmarks=300, max_track=12)
    ba = StereoBA.from_synth(prob)
    st = ba.stats()
    assert st.general_structure == 2 and st.num_superblocks == 834
    s, log = ba.solve(capi.default_options(max_num_iterations=K, use_nonmonotonic_steps=1))
    op = orc.OracleProblem.from_synth(prob)
    s2, log2 = op.solve(orc.driver_options(num_threads=16, max_num_iterations=K))
    _compare(s, log, s2, log2, ba, op, cost_rtol=1e-7)

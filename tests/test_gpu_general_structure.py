"""General problem structure through the C ABI: tracks longer than SSBA_MAX_TRACK observations and pose co-visibility
that is not banded (loop closures, arbitrary state numbering).  Ceres itself takes any structure
(tests/dataset_vo.cpp:41-56 adds whatever the dataset file holds), so the drop-in has to as well: such problems run
the dense reduced-system kernels of ssba_dense.hip, with the same oracle parity bar as the windowed layout."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc
from test_gpu_edge_cases import _assert_same_solve, _solve_both, assert_fixed_count_parity

pytestmark = pytest.mark.gpu


@contextmanager
def _force_dense():
    os.environ["SSBA_FORCE_DENSE"] = "1"
    try:
        yield
    finally:
        del os.environ["SSBA_FORCE_DENSE"]


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _permuted(prob, seed):
    """Same problem with the states renumbered at random (state 0 stays the constant first pose)."""
    rng = np.random.default_rng(seed)
    P = prob.num_poses
    new_of_old = np.concatenate([[0], 1 + rng.permutation(P - 1)])
    old_of_new = np.argsort(new_of_old)
    import copy
    q = copy.copy(prob)
    q.poses_init = prob.poses_init[old_of_new].copy()
    q.poses_gt = prob.poses_gt[old_of_new].copy()
    q.obs_pose = new_of_old[prob.obs_pose].astype(np.uint32)
    return q, new_of_old


def _track_lengths(prob):
    return np.bincount(prob.obs_point, minlength=prob.num_points)


@pytest.mark.parametrize("radius", [1e4, 3.0])
@pytest.mark.parametrize("size", [(5, 120, 4), (11, 300, 8), (30, 900, 12)])     # n = 24, 60, 174: below / across 64-blocks
def test_dense_reduced_system_and_step_match_oracle(size, radius):
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=3)
    with _force_dense():
        ba = StereoBA.from_synth(prob)
    assert ba.stats().general_structure == 1
    op = orc.OracleProblem.from_synth(prob)
    S, rhs, dp, dl, mcc = ba.lm_step(radius)
    S2, rhs2, _ = op.reduced_system(radius)
    dp2, dl2, mcc2 = op.lm_step(radius)
    assert _rel(S, S2) < 1e-10 and _rel(rhs, rhs2) < 1e-10
    assert _rel(dp, dp2) < 1e-8 and _rel(dl, dl2) < 1e-8
    assert mcc == pytest.approx(mcc2, rel=1e-9)


def test_forced_dense_solve_equals_windowed_solve():
    prob = synth.make_problem(40, 1600, track_len=10, seed=21)
    ba_w = StereoBA.from_synth(prob)
    s_w, log_w = ba_w.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    with _force_dense():
        ba_d = StereoBA.from_synth(prob)
    s_d, log_d = ba_d.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    assert ba_w.stats().general_structure == 0 and ba_d.stats().general_structure == 1
    assert s_d.num_iterations == s_w.num_iterations
    assert log_d["step_is_successful"].tolist() == log_w["step_is_successful"].tolist()
    np.testing.assert_allclose(log_d["cost"], log_w["cost"], rtol=1e-9)
    assert np.abs(ba_d.poses - ba_w.poses).max() < 1e-8


@contextmanager
def _no_wide():
    os.environ["SSBA_NO_WIDE"] = "1"
    try:
        yield
    finally:
        del os.environ["SSBA_NO_WIDE"]


@pytest.mark.parametrize("size", [(30, 1500, 20), (60, 2400, 40), (150, 4000, 16), (1400, 14000, 14), (1000, 20000, 24)])
def test_long_tracks_match_oracle(size):
    """Tracks of 13..24 observations run on 144-row super-blocks (ssba_wide.hip: matrix-core Schur items + parallel cyclic
    reduction), longer ones on the blocked Cholesky of the general path; both against the oracle's whole solve."""
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=size[2])
    assert _track_lengths(prob).max() > 12
    ba, s, log, op, s2, log2 = _solve_both(prob)
    st = ba.stats()
    assert st.general_structure == 1
    assert st.wide_superblocks == (0 if size[2] > 24 else (st.num_free_poses + 23) // 24)
    _assert_same_solve(ba, s, log, op, s2, log2)


@pytest.mark.parametrize("radius", [1e4, 3.0])
@pytest.mark.parametrize("size", [(14, 300, 13), (30, 900, 24), (75, 1800, 20), (200, 3000, 17)])     # 1, 2, 4 and 9 super-blocks of 24 poses
def test_wide_reduced_system_and_step_match_oracle(size, radius):
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=3)
    ba = StereoBA.from_synth(prob)
    st = ba.stats()
    assert st.general_structure == 1 and st.wide_superblocks == (st.num_free_poses + 23) // 24
    op = orc.OracleProblem.from_synth(prob)
    S, rhs, dp, dl, mcc = ba.lm_step(radius)
    S2, rhs2, _ = op.reduced_system(radius)
    dp2, dl2, mcc2 = op.lm_step(radius)
    assert _rel(S, S2) < 1e-10 and _rel(rhs, rhs2) < 1e-10
    assert _rel(dp, dp2) < 1e-8 and _rel(dl, dl2) < 1e-8
    assert mcc == pytest.approx(mcc2, rel=1e-9)


@pytest.mark.parametrize("strategy", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("huber_a", [0.0, 1.345])
def test_wide_super_blocks_equal_the_blocked_cholesky(strategy, huber_a):
    """The same long-track problem on the 144-row super-blocks and (SSBA_NO_WIDE=1) on the blocked Cholesky of the general
    path: LM, TRADITIONAL and SUBSPACE dogleg, with and without a loss; and both against the oracle."""
    prob = synth.make_problem(90, 2700, track_len=18, seed=11, outlier_fraction=0.1 if huber_a else 0.0)
    opts = dict(trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    ba, s, log, op, s2, log2 = _solve_both(prob, opts=opts, huber_a=huber_a)
    assert ba.stats().wide_superblocks == 4
    _assert_same_solve(ba, s, log, op, s2, log2)
    with _no_wide():
        ba_d = StereoBA.from_synth(prob, huber_a=huber_a)
    assert ba_d.stats().wide_superblocks == 0 and ba_d.stats().general_structure == 1
    s_d, log_d = ba_d.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1, **opts))
    assert s_d.num_iterations == s.num_iterations
    assert log_d["step_is_successful"].tolist() == log["step_is_successful"].tolist()
    ok = np.asarray(log["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log_d["cost"][ok], log["cost"][ok], rtol=1e-9)
    assert np.abs(ba_d.poses - ba.poses).max() < 1e-7


def test_wide_handle_hands_over_to_the_blocked_cholesky_for_a_covariance():
    """ceres::Covariance on a long-track problem (tests/dataset_vo_sun.cpp:159-183): the 144-row super-blocks have no
    covariance sweep, the handle runs its symbolic phase again and continues on the blocked Cholesky."""
    prob = synth.make_problem(40, 1200, track_len=16, seed=4)
    ba = StereoBA.from_synth(prob)
    assert ba.stats().wide_superblocks == 2
    ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    ba2 = StereoBA(prob.camera, ba.poses.copy(), ba.points.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness())
    cov = ba2.pose_covariance(7)
    assert ba2.stats().wide_superblocks == 0 and ba2.stats().general_structure == 1
    with _no_wide():
        ba3 = StereoBA(prob.camera, ba.poses.copy(), ba.points.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness())
    np.testing.assert_allclose(cov, ba3.pose_covariance(7), rtol=1e-9)


def test_matrix_core_and_valu_factorisations_agree(monkeypatch):
    """The blocked Cholesky of the general path on the fp64 matrix cores (k_dn_trsm_mf / k_dn_syrk_mf, back-substitution
    in one work-group) against the r01 kernels on the fp64 VALU (SSBA_DENSE_VALU=1): same accept / reject sequence,
    cost trace to 1e-9, poses to 1e-8 -- on a banded problem (long tracks) and on one with fill (renumbered states)."""
    for prob in (synth.make_problem(150, 4000, track_len=16, seed=16), _permuted(synth.make_problem(90, 2700, track_len=8, seed=3), 5)[0]):
        out = []
        for valu in ("0", "1"):
            monkeypatch.setenv("SSBA_DENSE_VALU", valu)
            ba = StereoBA.from_synth(prob)
            assert ba.stats().general_structure == 1
            s, log = ba.solve(capi.default_options(max_num_iterations=30, use_nonmonotonic_steps=1))
            out.append((s, log, ba.poses.copy()))
            ba.close()
        (s0, l0, p0), (s1, l1, p1) = out
        assert s0.num_iterations == s1.num_iterations
        assert l0["step_is_successful"].tolist() == l1["step_is_successful"].tolist()
        ok = np.asarray(l0["step_is_successful"], dtype=bool)
        ok[0] = True
        np.testing.assert_allclose(l0["cost"][ok], l1["cost"][ok], rtol=1e-9)
        assert np.abs(p0 - p1).max() < 1e-8


@pytest.mark.parametrize("strategy", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("huber_a", [0.0, 1.345])
def test_renumbered_states_match_oracle(strategy, huber_a):
    """Random state numbering = co-visibility all over the reduced system (what a loop closure does locally)."""
    base = synth.make_problem(36, 1400, track_len=8, seed=17, outlier_fraction=0.1 if huber_a else 0.0)
    prob, new_of_old = _permuted(base, seed=5)
    opts = dict(trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    ba, s, log, op, s2, log2 = _solve_both(prob, opts=opts, huber_a=huber_a)
    st = ba.stats()
    assert st.general_structure == 1 and st.pose_bandwidth > 12
    _assert_same_solve(ba, s, log, op, s2, log2)
    if huber_a == 0.0 and strategy == (0, 0):      # and the same minimum as the banded numbering
        ba0, s0, *_ = _solve_both(base)
        assert s.final_cost == pytest.approx(s0.final_cost, rel=1e-9)
        assert np.abs(ba.poses[new_of_old] - ba0.poses).max() < 1e-6


@contextmanager
def _no_closure_border():
    os.environ["SSBA_NO_CLOSURE_BORDER"] = "1"
    try:
        yield
    finally:
        del os.environ["SSBA_NO_CLOSURE_BORDER"]


@pytest.mark.parametrize("border", [True, False])
def test_loop_closure_observations(border):
    """A banded trajectory plus landmarks of the first states seen again from the last ones.  With at most 12
    observations per landmark the closing states become a border of the block-tridiagonal system (general_structure
    == 2, windowed kernels); SSBA_NO_CLOSURE_BORDER=1 sends the same problem down the general path."""
    prob = synth.make_problem(40, 1600, track_len=6, seed=9)
    q = synth.add_loop_closure(prob, num_states=3, num_landmarks=60)
    assert q.num_obs > prob.num_obs + 20
    if border:
        ba, s, log, op, s2, log2 = _solve_both(q, huber_a=1.345)
    else:
        with _no_closure_border():
            ba, s, log, op, s2, log2 = _solve_both(q, huber_a=1.345)
    st = ba.stats()
    assert st.general_structure == (2 if border else 1)
    if not border:
        assert st.pose_bandwidth >= 36
    _assert_same_solve(ba, s, log, op, s2, log2)


@pytest.mark.parametrize("size,pcr_max", [((100, 3000, 8), None), ((100, 3000, 8), "4"), ((300, 6000, 9), None)])
def test_closure_border_chain_lengths(size, pcr_max):
    """The closure border on chains of 9 and 25 super-blocks, with the parallel plan and with plain cyclic-reduction
    levels below it (SSBA_PCR_MAX_BLOCKS): same iterates as the oracle's banded solve of the whole system."""
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=5, pose_sigma=(0.004, 0.001))     # little drift: a closure that fits
    q = synth.add_loop_closure(prob, num_states=3, num_landmarks=80, max_track=12)
    if pcr_max:
        os.environ["SSBA_PCR_MAX_BLOCKS"] = pcr_max
    try:
        ba, s, log, op, s2, log2 = _solve_both(q)
    finally:
        os.environ.pop("SSBA_PCR_MAX_BLOCKS", None)
    assert ba.stats().general_structure == 2
    _assert_same_solve(ba, s, log, op, s2, log2)


def test_closure_border_equals_general_path():
    """Both routes for the same loop closure: chain + border against the blocked Cholesky of the general path."""
    prob = synth.make_problem(120, 4000, track_len=9, seed=12, pose_sigma=(0.004, 0.001))
    q = synth.add_loop_closure(prob, num_states=4, num_landmarks=100, max_track=12)
    opt = capi.default_options(max_num_iterations=15)
    ba = StereoBA.from_synth(q)
    s, log = ba.solve(opt)
    with _no_closure_border():
        ba2 = StereoBA.from_synth(q)
        s2, log2 = ba2.solve(opt)
    assert ba.stats().general_structure == 2 and ba2.stats().general_structure == 1
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-9)
    assert np.abs(ba.poses - ba2.poses).max() < 1e-7


def test_closure_border_hands_over_to_the_general_path_for_what_it_cannot_do():
    """DOGLEG and the pose covariance are not implemented on the closure border: a handle that was finalized with one runs
    its symbolic phase again and takes the general-structure path (r02 returned SSBA_ERR_UNSUPPORTED) -- against the
    oracle, and against a handle that was put on the general path from the start."""
    prob = synth.make_problem(40, 1600, track_len=6, seed=9)
    q = synth.add_loop_closure(prob, num_states=3, num_landmarks=60)
    kw = dict(max_num_iterations=12, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1)
    ba = StereoBA.from_synth(q)
    assert ba.stats().general_structure == 2
    s, log = ba.solve(capi.default_options(**kw))
    assert ba.stats().general_structure == 1          # re-finalized: the border is gone
    with _no_closure_border():
        bb = StereoBA.from_synth(q)
    sb, logb = bb.solve(capi.default_options(**kw))
    assert np.array_equal(log["cost"], logb["cost"]) and np.array_equal(ba.poses, bb.poses) and np.array_equal(ba.points, bb.points)
    # against the oracle: the drifted closure is ill-conditioned under DOGLEG -- the oracle itself moves by 5e-6 in the cost of
    # iteration 3 when the landmarks are scaled by 1 + 1e-14 (1.9e-8 at iteration 2), so only the first steps are comparable
    op = orc.OracleProblem.from_synth(q)
    s2, log2 = op.solve(orc.driver_options(num_threads=2, **kw))
    assert log["step_is_successful"][:7].tolist() == log2["step_is_successful"][:7].tolist()
    np.testing.assert_allclose(log["cost"][:3], log2["cost"][:3], rtol=1e-6)
    np.testing.assert_allclose(log["cost"][:7], log2["cost"][:7], rtol=1e-4)
    # covariance: a border handle after an LM solve, against a handle on the general path from the start
    ba1 = StereoBA.from_synth(q)
    ba1.solve(capi.default_options(max_num_iterations=30, use_nonmonotonic_steps=1))
    assert ba1.stats().general_structure == 2
    c1 = ba1.pose_covariance(39)
    assert ba1.stats().general_structure == 1
    with _no_closure_border():
        ba2 = StereoBA.from_synth(q)
    ba2.poses[:] = ba1.poses
    ba2.points[:] = ba1.points
    c2 = ba2.pose_covariance(39)
    np.testing.assert_allclose(c1, c2, rtol=1e-9, atol=1e-18)


def test_constant_states_and_nothing_free_on_the_general_path():
    prob = synth.make_problem(20, 700, track_len=16, seed=4)
    const = np.zeros(20, bool)
    const[[0, 9, 19]] = True
    ba, s, log, op, s2, log2 = _solve_both(prob, pose_const=const)
    assert ba.stats().general_structure == 1
    _assert_same_solve(ba, s, log, op, s2, log2)
    for k in (0, 9, 19):
        assert np.array_equal(ba.poses[k], prob.poses_init[k])
    ba, s, log, op, s2, log2 = _solve_both(prob, pose_const=np.ones(20, bool))
    _assert_same_solve(ba, s, log, op, s2, log2)
    assert np.array_equal(ba.poses, prob.poses_init)


def test_pose_factors_on_the_general_path():
    from test_oracle_pose_factors import _sun_problem
    prob, factors = _sun_problem(P=24, L=900, seed=8, huber=0.5)
    none_const = np.zeros(prob.num_poses, dtype=np.uint8)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1)
    with _force_dense():
        ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd,
                      prob.stiffness(), pose_const=none_const, pose_factors=factors)
    s, log = ba.solve(capi.default_options(**kw))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                           pose_const=none_const, pose_factors=factors)
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-7)
    with _force_dense():        # the end point at the north-star bar: the same solve cut at a fixed iteration count
        ba_k = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd,
                        prob.stiffness(), pose_const=none_const, pose_factors=factors)
    op_k = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                             pose_const=none_const, pose_factors=factors)
    assert_fixed_count_parity(ba_k, op_k, 12, **{k: v for k, v in kw.items() if k != "max_num_iterations"})


def _per_point_stiffness(prob, seed=0):
    """One 3x3 inverse square root of a random covariance per map point, expanded to the residual blocks
    (tests/dataset_vo_sun.cpp:56-59: SelfAdjointEigenSolver(stereo_obs_covars[j]).operatorInverseSqrt())."""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(prob.num_points, 3, 3))
    cov = A @ A.transpose(0, 2, 1) + 2.0 * np.eye(3)
    w, V = np.linalg.eigh(cov)
    S = np.einsum("nij,nj,nkj->nik", V, w ** -0.5, V)
    return np.ascontiguousarray(S[prob.obs_point])


@pytest.mark.parametrize("strategy", [(0, 0), (1, 1)])
@pytest.mark.parametrize("huber_a", [0.0, 1.345])
def test_per_point_stiffness_matches_oracle(strategy, huber_a):
    prob = synth.make_problem(24, 900, track_len=8, seed=31)
    S = _per_point_stiffness(prob)
    opts = dict(trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    ba, s, log, op, s2, log2 = _solve_both(prob, stiffness=S, opts=opts, huber_a=huber_a)
    assert ba.stats().general_structure == 1          # per-block stiffness lives in the general layout
    _assert_same_solve(ba, s, log, op, s2, log2)
    # and the reduced system of one LM step
    ba2 = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, S, huber_a=huber_a)
    op2 = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, S, huber_a=huber_a)
    Sg, rhs, dp, dl, mcc = ba2.lm_step(50.0)
    S2, rhs2, _ = op2.reduced_system(50.0)
    assert _rel(Sg, S2) < 1e-10 and _rel(rhs, rhs2) < 1e-10
    # the same matrix for every block is the shared-stiffness problem
    same = np.broadcast_to(prob.stiffness(), (prob.num_obs, 3, 3)).copy()
    ba3 = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, same)
    assert ba3.stats().general_structure == 0


@pytest.mark.parametrize("P", [6, 30])
def test_pose_covariance_on_the_general_path(P):
    """dataset_vo_sun.cpp:159-183 with the driver's per-point stiffness: the covariance block comes from the dense factor."""
    from test_oracle_pose_factors import _sun_problem
    prob, factors = _sun_problem(P=P, L=60 * P, seed=7)
    S = _per_point_stiffness(prob, seed=3)
    none_const = np.zeros(prob.num_poses, dtype=np.uint8)
    ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, S,
                  pose_const=none_const, pose_factors=factors)
    assert ba.stats().general_structure == 1
    ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1))
    op = orc.OracleProblem(prob.camera, ba.poses, ba.points, prob.obs_pose, prob.obs_point, prob.obs_uvd, S,
                           pose_const=none_const, pose_factors=factors)
    S2, _, free_idx = op.reduced_system(1e300)
    Sg = ba.lm_step(1e300)[0]
    assert _rel(Sg, S2) < 1e-5        # undamped landmark blocks: the Schur complement cancels ~8 digits
    Sginv = np.linalg.inv(Sg)
    for k in (1, P // 2, P - 1):
        f = int(free_idx[k])
        cov = ba.pose_covariance(k)
        assert _rel(cov, Sginv[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]) < 1e-6
        # only the prior holds the gauge: cond(S) ~ 1e9 amplifies the 1e-6 assembly difference of the two sides into per cents of
        # the inverse, and the figure moves with the summation order of H_pp (r04: 1.x e-2 with the shuffle tree of the wave
        # reductions, 2.9e-2 with the DPP tree, 3.3e-2 with H_pp formed as E^T (A^T A) E) -- a bound on noise, not an accuracy test
        assert _rel(cov, np.linalg.inv(S2)[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]) < 6e-2
        assert np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)


def test_structure_beyond_the_general_path_is_rejected_loudly():
    # landmark sharding: the windowed layout and tracks of 13..24 observations (144-row super-blocks); what needs the blocked
    # Cholesky of the general path -- longer tracks, or SSBA_NO_WIDE=1 -- is single GPU
    prob = synth.make_problem(20, 400, track_len=16, seed=1)
    assert StereoBA.from_synth(prob, world_size=2, rank=0).stats().wide_superblocks == 1
    with _no_wide():
        with pytest.raises(capi.SsbaError) as e:
            StereoBA.from_synth(prob, world_size=2, rank=0)
    assert e.value.status == -6      # SSBA_ERR_UNSUPPORTED
    with pytest.raises(capi.SsbaError) as e:
        StereoBA.from_synth(synth.make_problem(40, 400, track_len=30, seed=1), world_size=2, rank=0)
    assert e.value.status == -6


# ---- lighting terms on the general layout (the Phong driver takes any dataset) ----
def _phong_pair(prob, ph, shared_free=0, init="truth", **kw):
    d = ph.as_oracle_dict(init)
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=shared_free, **kw)
    okw = dict(kw)
    if "points_const" in okw:
        okw["positions_const"] = okw.pop("points_const")
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=shared_free, **okw)
    return ba, op


@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("shared_free", [0, 7])
def test_phong_step_on_the_general_layout(light_type, shared_free):
    prob, ph = synth.make_phong_problem(14, 500, track_len=20, seed=3, light_type=light_type)
    assert _track_lengths(prob).max() > 12
    ba, op = _phong_pair(prob, ph, shared_free, "perturbed" if shared_free else "truth")
    assert ba.stats().general_structure == 1
    for radius in (1e4, 5.0):
        S, rhs, dp, dl, mcc = ba.lm_step(radius)
        dp2, dl2, mcc2 = op.lm_step(radius)
        S2, rhs2, _ = op.reduced_system(radius)
        n = S.shape[0]
        assert _rel(S, S2[:n, :n]) < 1e-9 and _rel(rhs, rhs2[:n]) < 1e-9
        assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7
        assert mcc == pytest.approx(mcc2, rel=1e-8)


def test_forced_general_layout_equals_windowed_phong_solve():
    prob, ph = synth.make_phong_problem(20, 800, track_len=8, seed=5)
    d = ph.as_oracle_dict("perturbed")
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)
    ba_w = StereoBA.from_synth(prob, lighting=d, shared_free=7)
    s_w, log_w = ba_w.solve(capi.default_options(**kw))
    with _force_dense():
        ba_d = StereoBA.from_synth(prob, lighting=ph.as_oracle_dict("perturbed"), shared_free=7)
    s_d, log_d = ba_d.solve(capi.default_options(**kw))
    assert ba_d.stats().general_structure == 1 and ba_w.stats().general_structure == 0
    n = min(len(log_w["cost"]), len(log_d["cost"]), 12)
    assert log_d["step_is_successful"][:n].tolist() == log_w["step_is_successful"][:n].tolist()
    np.testing.assert_allclose(log_d["cost"][:n], log_w["cost"][:n], rtol=1e-8)
    assert s_d.final_cost == pytest.approx(s_w.final_cost, rel=1e-6)


@pytest.mark.parametrize("config", ["lm_const", "driver", "stage2"])
def test_phong_solves_with_long_tracks_match_oracle(config):
    """dataset_ba_phong on a sequence whose tracks exceed 12 observations: LM with constant shared blocks, the driver's
    own configuration (free shared blocks, bounds, SUBSPACE_DOGLEG) and the lighting-only stage of --multistage."""
    prob, ph = synth.make_phong_problem(24, 900, track_len=18, seed=8)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1)
    if config == "lm_const":
        ba, op = _phong_pair(prob, ph)
    elif config == "driver":
        kw.update(trust_region_strategy_type=1, dogleg_type=1)
        ba, op = _phong_pair(prob, ph, 7, "reference", use_bounds=True)
    else:
        kw.update(trust_region_strategy_type=1, dogleg_type=1)
        ba, op = _phong_pair(prob, ph, 7, "reference", use_bounds=True, pose_const=np.ones(prob.num_poses, np.uint8), points_const=True)
    assert ba.stats().general_structure == 1
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **{k: v for k, v in kw.items() if k.startswith(("trust", "dogleg"))}))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-6)
    # the end point at the north-star bar: the same solve cut at a fixed iteration count (fresh handles)
    if config == "lm_const":
        pair_k = _phong_pair(prob, ph)
    elif config == "driver":
        pair_k = _phong_pair(prob, ph, 7, "reference", use_bounds=True)
    else:
        pair_k = _phong_pair(prob, ph, 7, "reference", use_bounds=True, pose_const=np.ones(prob.num_poses, np.uint8), points_const=True)
    assert_fixed_count_parity(*pair_k, 10, cost_rtol=1e-6, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    assert np.abs(ba.poses - op.poses).max() < 1e-4
    assert np.abs(ba.normals - op.normals).max() < 1e-4


def test_phong_cpp_driver_on_a_long_track_dataset(tmp_path):
    """examples/dataset_ba_phong_gpu (the reference's solveWindow through the C++ shim) on a dataset with tracks of up
    to 20 observations: same result as the oracle with the driver's settings."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples("dataset_ba_phong_gpu")
    prob, ph = synth.make_phong_problem(30, 1200, track_len=20, seed=12)
    assert _track_lengths(prob).max() > 12
    files = synth.write_reference_phong_csv(prob, ph, str(tmp_path / "sim.csv"), shared="reference")
    r = subprocess.run([exe, *files], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = ph.as_oracle_dict("reference")
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s2, _ = op.solve(orc.driver_options(num_threads=4, trust_region_strategy_type=1, dogleg_type=1))
    report = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")][0]
    assert "Termination: CONVERGENCE" in report
    assert float(report.split("Final cost: ")[1].split(",")[0]) == pytest.approx(s2.final_cost, rel=1e-4)
    poses = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(poses - op.poses).max() < 1e-4


def test_two_residual_blocks_on_one_pose_landmark_pair():
    """Ceres takes any number of residual blocks on the same parameter blocks; a repeated (pose, landmark) observation
    has no slot in the windowed layout and runs on the general one."""
    prob = synth.make_problem(12, 300, track_len=5, seed=13)
    import copy
    q = copy.copy(prob)
    rng = np.random.default_rng(1)
    dup = rng.choice(prob.num_obs, 40, replace=False)
    q.obs_pose = np.concatenate([prob.obs_pose, prob.obs_pose[dup]]).astype(np.uint32)
    q.obs_point = np.concatenate([prob.obs_point, prob.obs_point[dup]]).astype(np.uint32)
    q.obs_uvd = np.vstack([prob.obs_uvd, prob.obs_uvd[dup] + rng.normal(size=(40, 3))])
    ba0 = StereoBA.from_synth(q)
    S, rhs, dp, dl, mcc = ba0.lm_step(30.0)               # at the initial guess
    S2, rhs2, _ = orc.OracleProblem.from_synth(q).reduced_system(30.0)
    assert _rel(S, S2) < 1e-10 and _rel(rhs, rhs2) < 1e-10
    ba, s, log, op, s2, log2 = _solve_both(q)
    assert ba.stats().general_structure == 1
    _assert_same_solve(ba, s, log, op, s2, log2)

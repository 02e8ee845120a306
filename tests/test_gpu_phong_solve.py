"""Config 3 on the GPU (stereo + Phong intensity + normal residual blocks, landmark block =
[position | normal], tests/dataset_ba_phong.cpp:101-204 with the shared light / material / texture
blocks held constant), called through the C ABI and compared with the CPU oracle on the same seeded
inputs.  fp64; block-level quantities <= 1e-10 relative, whole solves: the same accept / reject
sequence and final cost <= 1e-6 relative."""
import numpy as np
import pytest

from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from test_gpu_edge_cases import assert_fixed_count_parity
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _pair(prob, ph, shared_free=0):
    d = ph.as_oracle_dict("perturbed" if shared_free else "truth")
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=shared_free)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=shared_free)
    return ba, op


@pytest.mark.parametrize("light_type", [0, 1])
def test_phong_normal_equation_blocks_match_oracle(light_type):
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7, light_type=light_type)
    ba, op = _pair(prob, ph)
    cost, g_p, g_l, H_pp, H_ll = ba.evaluate()
    c2, gp2, gl2, Hp2, Hl2 = op.linearize()
    gp2[0] = 0
    Hp2[0] = 0
    assert g_l.shape == (60, 6) and H_ll.shape == (60, 6, 6)
    assert cost == pytest.approx(c2, rel=1e-12)
    assert _rel(g_p, gp2) < 1e-10 and _rel(g_l, gl2) < 1e-10
    assert _rel(H_pp, Hp2) < 1e-10 and _rel(H_ll, Hl2) < 1e-10


@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("radius", [1e4, 5.0])
def test_phong_reduced_system_and_step_match_oracle(light_type, radius):
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7, light_type=light_type)
    ba, op = _pair(prob, ph)
    S, rhs, dp, dl, mcc = ba.lm_step(radius)
    S2, rhs2, _ = op.reduced_system(radius)
    dp2, dl2, mcc2 = op.lm_step(radius)
    assert _rel(S, S2) < 1e-9 and _rel(rhs, rhs2) < 1e-9
    assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7
    assert mcc == pytest.approx(mcc2, rel=1e-8)


@pytest.mark.parametrize("light_type", [0, 1])
def test_phong_solve_matches_oracle(light_type):
    prob, ph = synth.make_phong_problem(50, 2000, light_type=light_type)
    ba, op = _pair(prob, ph)
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    s2, log2 = op.solve(orc.driver_options(num_threads=4))
    assert s.termination_type == s2.termination_type == 0
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-8)
    np.testing.assert_allclose(log["trust_region_radius"], log2["trust_region_radius"], rtol=1e-5)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    assert np.abs(ba.poses - op.poses).max() < 1e-6
    assert np.abs(ba.points - op.points).max() < 1e-5
    assert np.abs(ba.normals - op.normals).max() < 1e-6
    assert np.abs(np.linalg.norm(ba.normals, axis=1) - 1).max() < 1e-12     # UnitVectorPerturbation keeps |n| = 1
    assert np.array_equal(ba.poses[0], prob.poses_init[0])


def test_phong_ragged_tracks_and_long_window():
    # ragged masks, several Schur work items per window, more than one super-block
    prob, ph = synth.make_phong_problem(30, 3000, seed=11)
    keep = np.ones(prob.num_obs, dtype=bool)
    keep[::7] = False
    d = ph.as_oracle_dict()
    d["intensity"] = d["intensity"][keep]
    d["normal_obs"] = d["normal_obs"][keep]
    args = (prob.camera, prob.poses_init, prob.points_init, prob.obs_pose[keep], prob.obs_point[keep], prob.obs_uvd[keep],
            prob.stiffness())
    ba = StereoBA(args[0], args[1].copy(), args[2].copy(), *args[3:], lighting=d)
    op = orc.OracleProblem(*args, lighting=d)
    S, rhs, dp, dl, mcc = ba.lm_step(1e3)
    dp2, dl2, mcc2 = op.lm_step(1e3)
    assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7
    assert mcc == pytest.approx(mcc2, rel=1e-8)


# ---- free shared blocks: light, Phong parameters, textures (the border of the reduced system) ----
@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("shared_free,nb", [(7, 19), (1, 3), (6, 16), (4, 4)])
def test_border_system_and_step_match_oracle(light_type, shared_free, nb):
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7, light_type=light_type)
    for radius in (1e4, 5.0):
        ba, op = _pair(prob, ph, shared_free)
        S, rhs, dp, dl, mcc = ba.lm_step(radius)
        S_pb, S_bb, rhs_b, db = ba.border_system()
        assert S_pb.shape[1] == nb == op.border_size()
        A, b, _ = op.reduced_system(radius)
        n = A.shape[0] - nb
        assert _rel(S, A[:n, :n]) < 1e-9 and _rel(rhs, b[:n]) < 1e-9
        assert _rel(S_pb, A[:n, n:]) < 1e-9
        assert _rel(S_bb, A[n:, n:]) < 1e-9 and _rel(rhs_b, b[n:]) < 1e-9
        dp2, dl2, db2, mcc2 = op.lm_step(radius, want_border=True)
        assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7 and _rel(db, db2) < 1e-7
        assert mcc == pytest.approx(mcc2, rel=1e-8)
        # the multi-right-hand-side block cyclic reduction solves the same arrowhead system
        x = np.linalg.solve(A, b)
        assert _rel(dp[1:].ravel(), x[:n]) < 1e-8 and _rel(db, x[n:]) < 1e-8


@pytest.mark.parametrize("light_type", [0, 1])
def test_free_shared_blocks_solve_matches_oracle(light_type):
    prob, ph = synth.make_phong_problem(50, 2000, light_type=light_type)
    ba, op = _pair(prob, ph, 7)
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
    s2, log2 = op.solve(orc.driver_options(num_threads=4))
    assert s.termination_type == s2.termination_type == 0
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=1e-7)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    assert np.abs(ba.poses - op.poses).max() < 1e-6
    assert np.abs(ba.normals - op.normals).max() < 1e-6
    np.testing.assert_allclose(ba.phong, op.phong, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ba.texture, op.texture, rtol=1e-6)
    np.testing.assert_allclose(ba.light, op.light, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ba.texture, ph.texture, atol=5e-3)         # and they are the true values
    d0 = ph.as_oracle_dict("perturbed")
    assert not np.array_equal(ba.light, d0["light"])


def test_constant_shared_blocks_are_untouched_on_the_device():
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7)
    ba, op = _pair(prob, ph, 1)            # only the light is free
    d0 = ph.as_oracle_dict("perturbed")
    s, _ = ba.solve(capi.default_options(max_num_iterations=200, use_nonmonotonic_steps=1))
    s2, _ = op.solve(orc.driver_options(num_threads=2, max_num_iterations=200))
    np.testing.assert_array_equal(ba.phong, d0["phong"])
    np.testing.assert_array_equal(ba.texture, d0["texture"])
    np.testing.assert_allclose(ba.light, op.light, rtol=1e-6, atol=1e-7)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)


# ---- DOGLEG (tests/dataset_ba_phong.cpp:85-86 selects DOGLEG / SUBSPACE_DOGLEG) ----
@pytest.mark.parametrize("dogleg_type", [0, 1])
@pytest.mark.parametrize("shared_free", [0, 7])
@pytest.mark.parametrize("nonmono", [0, 1])
def test_phong_dogleg_solve_matches_oracle(dogleg_type, shared_free, nonmono):
    prob, ph = synth.make_phong_problem(50, 2000)
    ba, op = _pair(prob, ph, shared_free)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=nonmono, trust_region_strategy_type=1, dogleg_type=dogleg_type)
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 14)          # the first iterations agree tightly ...
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-6)
    np.testing.assert_allclose(log["trust_region_radius"][:n], log2["trust_region_radius"][:n], rtol=1e-5)
    # ... and the end point of the same solve cut at a fixed iteration count holds the 1e-6 bar (long non-monotonic tails
    # stop at a rounding-sensitive iteration)
    assert_fixed_count_parity(*_pair(prob, ph, shared_free), 14, cost_rtol=1e-6, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    if not nonmono:
        assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
        assert s.num_iterations == s2.num_iterations
        assert np.abs(ba.poses - op.poses).max() < 1e-5
        assert np.abs(ba.normals - op.normals).max() < 1e-5


# ---- bounds on the Phong / texture blocks: projected Plus + Armijo line search ----
@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("init", ["perturbed", "reference"])
@pytest.mark.parametrize("strategy", [(0, 0), (1, 1)])       # LM, and the driver's DOGLEG / SUBSPACE_DOGLEG
def test_bounded_solve_matches_oracle(light_type, init, strategy):
    prob, ph = synth.make_phong_problem(50, 2000, light_type=light_type)
    d = ph.as_oracle_dict(init)
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-6)
    ba_k = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True)
    op_k = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                             prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    assert_fixed_count_parity(ba_k, op_k, 12, cost_rtol=1e-6, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    assert np.all(ba.phong[:, :2] >= 0) and np.all(ba.phong[:, :2] <= 1) and np.all(ba.phong[:, 2] >= 1)
    assert np.all(ba.texture >= 0) and np.all(ba.texture <= 1)
    if len(log["cost"]) == len(log2["cost"]):
        np.testing.assert_allclose(ba.texture, op.texture, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("strategy", [(0, 0), (1, 1)])
def test_line_search_on_the_device_is_the_search_the_host_drives(strategy, monkeypatch):
    """The Armijo state machine (csrc/ssba_linesearch.h) runs in k_ph_ls_reduce for the evaluations enqueued with every
    iteration (SSBA_LS_ROUNDS; unset: two, and as many as a search took once one ran out of them -- at most four);
    SSBA_LS_ROUNDS=0 leaves every search to the host (the r02 path), 1 makes the
    hand-over of a search that needs more evaluations than were enqueued the common case.  Same code, same evaluations:
    the solves must agree bit for bit, and the evaluation count must be the oracle's (Summary::num_line_search_steps)."""
    prob, ph = synth.make_phong_problem(50, 2000)
    d = ph.as_oracle_dict("reference")
    kw = dict(max_num_iterations=25, use_nonmonotonic_steps=1, trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    runs = {}
    for rounds in ("3", "0", "1", "20", "unset"):
        if rounds == "unset":
            monkeypatch.delenv("SSBA_LS_ROUNDS")
        else:
            monkeypatch.setenv("SSBA_LS_ROUNDS", rounds)
        ba = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True)
        s, log = ba.solve(capi.default_options(**kw))
        runs[rounds] = (s, log, ba.poses.copy(), ba.texture.copy())
    s3, log3 = runs["3"][:2]
    assert s3.num_line_searches_on_device > 0 and runs["0"][0].num_line_searches_on_device == 0
    assert runs["0"][0].num_line_searches_by_host == s3.num_line_searches_on_device + s3.num_line_searches_by_host
    assert runs["1"][0].num_line_searches_by_host > 0 and runs["20"][0].num_line_searches_by_host == 0
    for rounds in ("0", "1", "20", "unset"):
        s, log, poses, tex = runs[rounds]
        assert s.num_iterations == s3.num_iterations and s.num_line_search_steps == s3.num_line_search_steps
        assert log["step_is_successful"].tolist() == log3["step_is_successful"].tolist()
        # bit for bit: the state machine uses no libm beyond sqrt and contracts no a * b + c (ssba_linesearch.h)
        np.testing.assert_array_equal(log["cost"], log3["cost"])
        np.testing.assert_array_equal(poses, runs["3"][2])
        np.testing.assert_array_equal(tex, runs["3"][3])
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    n = 12
    assert log3["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    if len(log3["cost"]) == len(log2["cost"]) and log3["step_is_successful"].tolist() == log2["step_is_successful"].tolist():
        assert s3.num_line_search_steps == s2.num_line_search_steps


def test_armijo_state_machine_gives_the_same_bits_on_the_device_and_on_the_host():
    """csrc/ssba_linesearch.h in a one-lane kernel against the same header on the host (ssba_armijo_trace): every step the
    search asks for, through cubic and quintic interpolations and the Aberth root finder, bit for bit."""
    from test_oracle_bounds import _product_trace, _search_cases
    long_searches = 0
    for phi, dphi in _search_cases(60, seed=17):
        host, host_opt = _product_trace(phi, dphi, on_device=0)
        dev, dev_opt = _product_trace(phi, dphi, on_device=1)
        assert dev == host and dev_opt == host_opt
        long_searches += len(host) >= 3
    assert long_searches >= 15


def test_infeasible_start_is_projected_on_the_device():
    prob, ph = synth.make_phong_problem(20, 600, track_len=8, seed=2)
    d = ph.as_oracle_dict("perturbed")
    d["phong"][:, 1] = -0.2
    d["phong"][0, 2] = 0.5
    d["texture"][1] = 1.4
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s, log, s2, log2 = assert_fixed_count_parity(ba, op, 15, cost_rtol=1e-6, use_nonmonotonic_steps=1)
    assert s.initial_cost == pytest.approx(s2.initial_cost, rel=1e-12)
    assert np.all(ba.phong[:, 1] >= 0) and np.all(ba.phong[:, 2] >= 1) and np.all(ba.texture <= 1)


@pytest.mark.parametrize("light_type", [0, 1])
def test_config1_phong_driver_through_the_ceres_shim(tmp_path, light_type):
    """BASELINE.json configs[0]: dataset_ba_phong on a 50-pose / 2 000-landmark sequence.
    examples/dataset_ba_phong_gpu.cpp = the reference's solveWindow (tests/dataset_ba_phong.cpp:26-255)
    written against the C++ shim with the driver's own settings -- DOGLEG / SUBSPACE_DOGLEG, non-monotonic
    steps, free light / material / texture blocks with their bounds, the reference's initial material
    values -- fed through the reference's own CSV formats, against the oracle with the same settings."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples("dataset_ba_phong_gpu")
    prob, ph = synth.make_phong_problem(50, 2000, light_type=light_type)
    files = synth.write_reference_phong_csv(prob, ph, str(tmp_path / "sim.csv"), shared="reference")
    r = subprocess.run([exe, *files] + (["--dirlight"] if light_type else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = ph.as_oracle_dict("reference")
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s2, _ = op.solve(orc.driver_options(num_threads=4, trust_region_strategy_type=1, dogleg_type=1))
    report = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")][0]
    assert "Termination: CONVERGENCE" in report
    final = float(report.split("Final cost: ")[1].split(",")[0])
    assert final == pytest.approx(s2.final_cost, rel=1e-4)
    poses = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(poses - op.poses).max() < 1e-4
    m = np.loadtxt(str(tmp_path / "sim_map.csv"), delimiter=",", skiprows=1)
    assert m.shape == (2000, 11)
    assert np.abs(np.linalg.norm(m[:, 4:7], axis=1) - 1).max() < 1e-12
    assert np.all(m[:, 7:9] >= 0) and np.all(m[:, 7:9] <= 1) and np.all(m[:, 9] >= 1) and np.all((m[:, 10] >= 0) & (m[:, 10] <= 1))
    lights = np.loadtxt(str(tmp_path / "sim_lights.csv"), delimiter=",", skiprows=1)
    np.testing.assert_allclose(lights, op.light, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shared_free", [0, 7])
def test_phong_with_huber_loss_on_the_stereo_blocks(shared_free):
    """BASELINE config 5 shape (30 % outlier observations, Huber a = 1.345) on the config-3 graph: the loss
    sits on the stereo residual blocks only, the lighting blocks keep their NULL loss."""
    prob, ph = synth.make_phong_problem(50, 2000, outlier_fraction=0.3)
    d = ph.as_oracle_dict("perturbed" if shared_free else "truth")
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=shared_free, huber_a=1.345)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=shared_free, huber_a=1.345)
    S, rhs, dp, dl, mcc = ba.lm_step(1e4)
    dp2, dl2, mcc2 = op.lm_step(1e4)
    assert _rel(dp, dp2) < 1e-7 and _rel(dl, dl2) < 1e-7 and mcc == pytest.approx(mcc2, rel=1e-8)
    assert_fixed_count_parity(ba, op, 12, cost_rtol=1e-7, use_nonmonotonic_steps=1)


def test_phong_unsupported_combinations_fail_loudly():
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7)
    # lighting observations must pair with the stereo observations
    d = ph.as_oracle_dict()
    d["intensity"] = d["intensity"][:-1]
    d["normal_obs"] = d["normal_obs"][:-1]
    with pytest.raises((capi.SsbaError, AssertionError)):
        StereoBA.from_synth(prob, lighting=d)


# ---- --multistage (tests/dataset_ba_phong.cpp:96-246): stage 2 holds every pose and position block constant ----
@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("poses_free", [False, True])
def test_constant_positions_lighting_only_stage(light_type, poses_free):
    prob, ph = synth.make_phong_problem(20, 800, track_len=8, seed=5, light_type=light_type)
    d = ph.as_oracle_dict("reference")
    const = np.ones(prob.num_poses, dtype=np.uint8)
    if poses_free:                      # not a driver setting, but the same machinery: positions fixed, poses free
        const[1:] = 0
    kw = dict(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1)
    ba = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True, pose_const=const, points_const=True)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True, pose_const=const, positions_const=True)
    S, rhs, dp, dl, mcc = ba.lm_step(1e3)
    dp2, dl2, mcc2 = op.lm_step(1e3)
    assert np.abs(dl[:, :3]).max() == 0.0 and np.abs(dl2[:, :3]).max() == 0.0
    assert _rel(dl, dl2) < 1e-7 and mcc == pytest.approx(mcc2, rel=1e-8)
    if poses_free:
        assert _rel(dp, dp2) < 1e-7
    s, log = ba.solve(capi.default_options(**kw))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, **kw))
    assert s.termination_type == s2.termination_type == 0
    n = min(len(log["cost"]), len(log2["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == log2["step_is_successful"][:n].tolist()
    ok = np.asarray(log2["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], log2["cost"][:n][ok], rtol=1e-6)
    ba_k = StereoBA.from_synth(prob, lighting=d, shared_free=7, use_bounds=True, pose_const=const, points_const=True)
    op_k = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                             prob.stiffness(), lighting=d, shared_free=7, use_bounds=True, pose_const=const, positions_const=True)
    assert_fixed_count_parity(ba_k, op_k, 10, cost_rtol=1e-6, **{k: v for k, v in kw.items() if k != "max_num_iterations"})
    assert np.array_equal(ba.points, prob.points_init)            # constant blocks come back bit for bit
    if not poses_free:
        assert np.array_equal(ba.poses, prob.poses_init)
    assert np.abs(ba.normals - op.normals).max() < 1e-4
    np.testing.assert_allclose(ba.light, op.light, rtol=1e-4, atol=1e-5)


def test_phong_driver_multistage(tmp_path):
    """dataset_ba_phong --multistage (tests/dataset_ba_phong.cpp:96-100, 210-252): stage 1 without lighting terms,
    stage 2 lighting only with every pose and position block constant, stage 3 jointly; against the same three
    solves on the oracle."""
    import subprocess
    from ceres_slam_amd import build
    exe = build.build_examples("dataset_ba_phong_gpu")
    prob, ph = synth.make_phong_problem(30, 1200, track_len=8, seed=6)
    files = synth.write_reference_phong_csv(prob, ph, str(tmp_path / "sim.csv"), shared="reference")
    r = subprocess.run([exe, *files, "--multistage"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    reports = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")]
    assert len(reports) == 3 and all("Termination: CONVERGENCE" in l for l in reports)
    kw = dict(num_threads=4, trust_region_strategy_type=1, dogleg_type=1)
    args = (prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness())
    s1 = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, *args)
    r1, _ = s1.solve(orc.driver_options(**kw))
    d = ph.as_oracle_dict("reference")
    s2 = orc.OracleProblem(prob.camera, s1.poses, s1.points, *args, lighting=d, shared_free=7, use_bounds=True,
                           pose_const=np.ones(prob.num_poses, np.uint8), positions_const=True)
    r2, _ = s2.solve(orc.driver_options(**kw))
    d3 = dict(d, normals=s2.normals, phong=s2.phong, texture=s2.texture, light=s2.light)
    s3 = orc.OracleProblem(prob.camera, s2.poses, s2.points, *args, lighting=d3, shared_free=7, use_bounds=True)
    r3, _ = s3.solve(orc.driver_options(**kw))
    finals = [float(l.split("Final cost: ")[1].split(",")[0]) for l in reports]
    assert finals[0] == pytest.approx(r1.final_cost, rel=1e-5)
    assert finals[1] == pytest.approx(r2.final_cost, rel=1e-4)
    assert finals[2] == pytest.approx(r3.final_cost, rel=1e-4)
    poses = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    assert np.abs(poses - s3.poses).max() < 1e-4
    lights = np.loadtxt(str(tmp_path / "sim_lights.csv"), delimiter=",", skiprows=1)
    np.testing.assert_allclose(lights, s3.light, rtol=1e-4, atol=1e-5)


def test_free_shared_blocks_beyond_the_parallel_plan():
    """More than 128 super-blocks with free shared blocks: the border columns go through plain cyclic reduction
    (the parallel plan with kept factors covers <= 128 blocks)."""
    prob, ph = synth.make_phong_problem(1600, 8000, track_len=10, seed=4)
    ba, op = _pair(prob, ph, 7)
    assert ba.stats().pcr_blocks == 0
    S, rhs, dp, dl, mcc = ba.lm_step(1e3)
    dp2, dl2, mcc2 = op.lm_step(1e3)
    assert _rel(dp, dp2) < 1e-6 and _rel(dl, dl2) < 1e-6
    assert mcc == pytest.approx(mcc2, rel=1e-7)
    small, ph2 = synth.make_phong_problem(60, 1500, track_len=10, seed=4)
    assert _pair(small, ph2, 7)[0].stats().pcr_blocks == 5


def test_c1_phong_driver_configuration_matches_golden():
    """BASELINE.json configs[0] with the driver's settings against the committed golden solve."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c1_phong_driver.json")))
    prob, ph = synth.make_phong_problem(50, 2000)
    ba = StereoBA.from_synth(prob, lighting=ph.as_oracle_dict("reference"), shared_free=7, use_bounds=True)
    s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1))
    assert s.initial_cost == pytest.approx(gold["initial_cost"], rel=1e-12)
    n = min(len(log["cost"]), len(gold["cost"]), 12)
    assert log["step_is_successful"][:n].tolist() == gold["step_is_successful"][:n]
    ok = np.asarray(gold["step_is_successful"][:n], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][:n][ok], np.asarray(gold["cost"])[:n][ok], rtol=1e-6)
    # the end point at the north-star bar: the same solve cut at 12 iterations on both sides (the converged run ends on a
    # flat, rounding-sensitive tail)
    g12 = gold["at_12_iterations"]
    ba12 = StereoBA.from_synth(prob, lighting=ph.as_oracle_dict("reference"), shared_free=7, use_bounds=True)
    s12, _ = ba12.solve(capi.default_options(max_num_iterations=12, use_nonmonotonic_steps=1, trust_region_strategy_type=1, dogleg_type=1))
    assert s12.final_cost == pytest.approx(g12["final_cost"], rel=1e-6)
    np.testing.assert_allclose(ba12.poses[[1, 25, 49]], g12["poses_1_25_49"], atol=1e-6)
    np.testing.assert_allclose(ba12.light, g12["light"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ba12.texture, g12["texture"], rtol=1e-6)


def test_twelve_materials_with_free_light_and_textures():
    """The reference sizes materials / textures from the dataset file (dataset_problem_phong.cpp:266-278); more than the
    first build's seven: 12 materials, light and texture blocks free (a border of 3 + 12 columns), Phong parameters
    constant, against the oracle."""
    prob, ph = synth.make_phong_problem(30, 1500, num_materials=12, seed=5)
    shared_free = 0b101
    ba, op = _pair(prob, ph, shared_free=shared_free)
    s, log = ba.solve(capi.default_options(max_num_iterations=25, use_nonmonotonic_steps=1))
    s2, log2 = op.solve(orc.driver_options(num_threads=4, max_num_iterations=25))
    assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
    ok = np.asarray(log2["step_is_successful"], dtype=bool)
    ok[0] = True
    np.testing.assert_allclose(log["cost"][ok], log2["cost"][ok], rtol=1e-7)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-6)
    assert np.abs(ba.poses - op.poses).max() < 1e-6
    assert np.abs(ba.texture - op.texture).max() < 1e-6 and np.abs(ba.light - op.light).max() < 1e-5


@pytest.mark.parametrize("switch", ["SSBA_BORDER_POSE_KERNEL", "SSBA_BORDER_SWEEPS", "SSBA_BCR_LEGACY_BORDER"])
def test_border_routes_agree(switch):
    """The production route of the free shared blocks -- pose rows of S_pb out of the Schur product's border tiles, border
    columns riding in the matrix-core factor / reduce launches -- against the first-generation kernels it replaced
    (pose-by-pose S_pb; separate forward / update sweeps per level), on a chain with the parallel plan and on one with
    plain cyclic-reduction levels."""
    import os
    for size, pcr_max in (((60, 2400), None), ((60, 2400), "2")):
        prob, ph = synth.make_phong_problem(*size, seed=8)
        env = {"SSBA_BORDER_POSE_KERNEL": {"SSBA_BORDER_POSE_KERNEL": "1"}, "SSBA_BORDER_SWEEPS": {"SSBA_BORDER_SWEEPS": "1"},
               "SSBA_BCR_LEGACY_BORDER": {"SSBA_BORDER_POSE_KERNEL": "1", "SSBA_BORDER_SWEEPS": "1"}}[switch]
        if pcr_max:
            env = dict(env, SSBA_PCR_MAX_BLOCKS=pcr_max)
            os.environ["SSBA_PCR_MAX_BLOCKS"] = pcr_max
        try:
            ba, _ = _pair(prob, ph, 7)
            s, log = ba.solve(capi.default_options(max_num_iterations=12, use_nonmonotonic_steps=1))
            for k, v in env.items():
                os.environ[k] = v
            ba2, _ = _pair(prob, ph, 7)
            s2, log2 = ba2.solve(capi.default_options(max_num_iterations=12, use_nonmonotonic_steps=1))
        finally:
            for k in list(env) + ["SSBA_PCR_MAX_BLOCKS"]:
                os.environ.pop(k, None)
        assert log["step_is_successful"].tolist() == log2["step_is_successful"].tolist()
        np.testing.assert_allclose(log["cost"], log2["cost"], rtol=1e-9)
        assert np.abs(ba.poses - ba2.poses).max() < 1e-8
        np.testing.assert_allclose(ba.light, ba2.light, rtol=1e-8, atol=1e-10)


def test_inversion_beside_the_pose_linearisation_is_the_same_arithmetic(monkeypatch):
    """Constant shared blocks: the inversion of the damped landmark blocks runs in the pose linearisation's launch
    (k_ph_linpose_invert) by default and as a launch of its own with SSBA_PH_INVERT_LAUNCH=1 -- the same per-work-group
    code either way, so the two solves agree bit for bit."""
    prob, ph = synth.make_phong_problem(50, 2000, num_materials=4, seed=6)
    out = []
    for env in ("0", "1"):
        monkeypatch.setenv("SSBA_PH_INVERT_LAUNCH", env)
        ba = StereoBA.from_synth(prob, lighting=ph.as_oracle_dict("truth"))
        s, log = ba.solve(capi.default_options(max_num_iterations=25, use_nonmonotonic_steps=1))
        out.append((s, log, ba.poses.copy(), ba.points.copy()))
        ba.close()
    (s0, l0, p0, x0), (s1, l1, p1, x1) = out
    assert s0.num_iterations == s1.num_iterations
    assert np.array_equal(l0["cost"], l1["cost"]) and np.array_equal(p0, p1) and np.array_equal(x0, x1)


@pytest.mark.parametrize("strategy", [(0, 0), (1, 1)])
def test_right_hand_side_in_the_border_columns_backward_sweep_is_bit_identical(strategy, monkeypatch):
    """With the border columns riding through the parallel plan, the decoupled last step's right-hand side is solved by
    k_bcrm_bwd as one more column (the padding column NBP - 1) instead of by a k_bcr_backsub launch of its own: the same sweep,
    lane for lane in the same order.  SSBA_NO_RHS_RIDE=1 keeps the launch: the two solves must agree bit for bit."""
    prob, ph = synth.make_phong_problem(50, 2000)
    d = ph.as_oracle_dict("perturbed")
    kw = dict(max_num_iterations=15, use_nonmonotonic_steps=1, trust_region_strategy_type=strategy[0], dogleg_type=strategy[1])
    runs = []
    for off in ("0", "1"):
        monkeypatch.setenv("SSBA_NO_RHS_RIDE", off)
        ba = StereoBA.from_synth(prob, lighting=d, shared_free=7)
        s, log = ba.solve(capi.default_options(**kw))
        runs.append((log["cost"].copy(), ba.poses.copy(), ba.texture.copy()))
        ba.close()
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    np.testing.assert_array_equal(runs[0][2], runs[1][2])

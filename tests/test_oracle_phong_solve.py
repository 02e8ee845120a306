"""Config 3 (stereo + Phong intensity + normal residual blocks with 6-D landmark blocks): the oracle's
normal equations / LM step against an independent numpy restatement with complex-step Jacobians
through the reference's Plus operators, and whole-solve sanity.  CPU only."""
import json
import os

import numpy as np
import pytest

import np_reference as npr
from ceres_slam_amd import synth
from oracle import oracle as orc


def _oracle(prob, ph):
    return orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                             prob.stiffness(), lighting=ph.as_oracle_dict())


@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("radius", [1e4, 5.0])
def test_phong_lm_step_matches_independent_sparse_solve(light_type, radius):
    prob, ph = synth.make_phong_problem(6, 40, track_len=4, seed=3, light_type=light_type)
    op = _oracle(prob, ph)
    dp, dl, mcc = op.lm_step(radius)
    d = ph.as_oracle_dict()
    dp2, dl2, mcc2, cost2 = npr.phong_lm_step(prob.camera, prob.poses_init, prob.points_init, ph.normals_init, prob.obs_pose,
                                              prob.obs_point, prob.obs_uvd, prob.stiffness(), d, radius)
    assert op.cost() == pytest.approx(cost2, rel=1e-12)
    np.testing.assert_allclose(dp, dp2, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(dl, dl2, rtol=1e-7, atol=1e-9)
    assert mcc == pytest.approx(mcc2, rel=1e-8)
    assert dl.shape == (40, 6)


def test_phong_solve_converges_to_the_statistical_optimum():
    prob, ph = synth.make_phong_problem(50, 2000)
    op = _oracle(prob, ph)
    s, log = op.solve(orc.driver_options(num_threads=4))
    assert s.termination_type == 0
    # 7 residuals per observation with unit-variance whitening: optimum ~ (7N - dof)/2
    dof = 6 * 49 + 5 * 2000
    assert s.final_cost == pytest.approx(0.5 * (7 * prob.num_obs - dof), rel=0.05)
    assert np.abs(np.linalg.norm(op.normals, axis=1) - 1).max() < 1e-12     # UnitVectorPerturbation keeps |n| = 1
    assert np.abs(op.normals - ph.normals_gt).max() < np.abs(ph.normals_init - ph.normals_gt).max()
    assert np.array_equal(op.poses[0], prob.poses_init[0])


def test_stereo_only_problem_is_unchanged_by_the_generalisation(c1_problem):
    # lighting=None must reproduce the committed golden solve
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c1_solve.json")))
    op = orc.OracleProblem.from_synth(c1_problem)
    s, log = op.solve(orc.driver_options(num_threads=1))
    np.testing.assert_allclose(log["cost"], gold["cost"], rtol=1e-9)   # summation order changed, values did not


def test_c1_phong_driver_configuration_matches_golden():
    """BASELINE.json configs[0] with the driver's settings against tests/golden/c1_phong_driver.json (oracle output)."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c1_phong_driver.json")))
    prob, ph = synth.make_phong_problem(50, 2000)
    d = ph.as_oracle_dict("reference")
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    assert op.cost() == pytest.approx(gold["initial_cost"], rel=1e-12)
    s, log = op.solve(orc.driver_options(num_threads=4, trust_region_strategy_type=1, dogleg_type=1))
    assert s.num_iterations == gold["num_iterations"] and log["step_is_successful"].tolist() == gold["step_is_successful"]
    # the fixture was written single-threaded; 4 threads reorder the sums and the path amplifies that to ~1e-8
    np.testing.assert_allclose(log["cost"], gold["cost"], rtol=1e-6)
    assert s.final_cost == pytest.approx(gold["final_cost"], rel=1e-7)
    np.testing.assert_allclose(op.poses[[1, 25, 49]], gold["poses_1_25_49"], atol=1e-5)
    np.testing.assert_allclose(op.light, gold["light"], rtol=1e-6)

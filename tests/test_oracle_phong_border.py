"""Config 3 with FREE shared blocks (light, Phong parameters and texture of each material -- the
blocks every intensity residual of tests/dataset_ba_phong.cpp:108-139 touches): the oracle's bordered
normal equations / LM step against the independent numpy restatement, and whole-solve recovery of the
shared parameters.  CPU only."""
import numpy as np
import pytest

import np_reference as npr
from ceres_slam_amd import synth
from oracle import oracle as orc


def _oracle(prob, ph, shared_free, shared="perturbed"):
    return orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                             prob.stiffness(), lighting=ph.as_oracle_dict(shared), shared_free=shared_free)


@pytest.mark.parametrize("light_type", [0, 1])
@pytest.mark.parametrize("shared_free,nb", [(7, 19), (1, 3), (6, 16)])
def test_bordered_lm_step_matches_independent_sparse_solve(light_type, shared_free, nb):
    prob, ph = synth.make_phong_problem(6, 40, track_len=4, seed=3, light_type=light_type)
    op = _oracle(prob, ph, shared_free)
    assert op.border_size() == nb
    d = ph.as_oracle_dict("perturbed")
    for radius in (1e4, 5.0):
        dp, dl, db, mcc = op.lm_step(radius, want_border=True)
        dp2, dl2, mcc2, cost2, db2 = npr.phong_lm_step(prob.camera, prob.poses_init, prob.points_init, ph.normals_init,
                                                        prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(), d,
                                                        radius, shared_free=shared_free)
        assert op.cost() == pytest.approx(cost2, rel=1e-12)
        np.testing.assert_allclose(dp, dp2, rtol=1e-7, atol=1e-9 * np.abs(dp2).max())
        np.testing.assert_allclose(dl, dl2, rtol=1e-7, atol=1e-9 * np.abs(dl2).max())
        np.testing.assert_allclose(db, db2, rtol=1e-7, atol=1e-9 * np.abs(db2).max())
        assert mcc == pytest.approx(mcc2, rel=1e-8)
    # the arrowhead system the test hook returns reproduces the step
    S, rhs, _ = op.reduced_system(5.0)
    x = np.linalg.solve(S, rhs)
    n = S.shape[0] - nb
    np.testing.assert_allclose(x[:n], dp[1:].ravel(), atol=1e-11)
    np.testing.assert_allclose(x[n:], db, atol=1e-11)


@pytest.mark.parametrize("light_type", [0, 1])
def test_solve_recovers_the_shared_blocks(light_type):
    prob, ph = synth.make_phong_problem(50, 2000, light_type=light_type)
    op = _oracle(prob, ph, 7)
    s, log = op.solve(orc.driver_options(num_threads=4))
    assert s.termination_type == 0
    # more freedom than the constant-block problem started from the true shared values: lower optimum
    op0 = _oracle(prob, ph, 0, shared="truth")
    s0, _ = op0.solve(orc.driver_options(num_threads=4))
    assert s.final_cost < s0.final_cost * (1 + 1e-4)
    np.testing.assert_allclose(op.phong[:, 1], ph.phong[:, 1], atol=5e-3)          # ks
    np.testing.assert_allclose(op.phong[:, 2], ph.phong[:, 2], rtol=0.05)          # alpha
    np.testing.assert_array_equal(op.phong[:, 0], ph.phong_init[:, 0])             # ka: zero Jacobian, never moves
    np.testing.assert_allclose(op.texture, ph.texture, atol=5e-3)
    if light_type == 0:
        np.testing.assert_allclose(op.light, ph.light, atol=0.2)
    else:
        assert abs(np.linalg.norm(op.light) - 1) < 1e-12                           # UnitVectorPerturbation
        np.testing.assert_allclose(op.light, ph.light, atol=5e-3)


def test_constant_blocks_are_untouched():
    prob, ph = synth.make_phong_problem(8, 60, track_len=5, seed=7)
    op = _oracle(prob, ph, 1)           # only the light is free
    d = ph.as_oracle_dict("perturbed")
    s, _ = op.solve(orc.driver_options(num_threads=2))
    np.testing.assert_array_equal(op.phong, d["phong"])
    np.testing.assert_array_equal(op.texture, d["texture"])
    assert not np.array_equal(op.light, d["light"])
    # DOGLEG with a border is rejected loudly by the oracle as well
    with pytest.raises(Exception):
        op2 = _oracle(prob, ph, 7)
        s2, _ = op2.solve(orc.driver_options(num_threads=2, trust_region_strategy_type=1))
        assert s2.termination_type != 2

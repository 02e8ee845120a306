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


def test_polynomial_roots_match_numpy():
    """The quartic of DoglegStrategy::FindMinimumOnTrustRegionBoundary is solved by Aberth iteration in
    the oracle (Ceres: companion-matrix eigenvalues); the real parts of all roots must agree with numpy."""
    import ctypes as C
    lib = orc.lib()
    lib.orc_poly_roots_real.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    rng = np.random.default_rng(3)
    for deg in (1, 2, 3, 4):
        for _ in range(50):
            c = rng.normal(size=deg + 1) * 10.0 ** rng.integers(-3, 4, deg + 1)
            out = np.zeros(8)
            n = lib.orc_poly_roots_real(c.ctypes.data_as(C.POINTER(C.c_double)), deg + 1, out.ctypes.data_as(C.POINTER(C.c_double)))
            assert n == deg
            ref = np.sort(np.roots(c).real)
            np.testing.assert_allclose(np.sort(out[:n]), ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())
    # leading zeros are dropped first
    c = np.array([0.0, 0.0, 2.0, -6.0, 4.0])
    out = np.zeros(8)
    assert lib.orc_poly_roots_real(c.ctypes.data_as(C.POINTER(C.c_double)), 5, out.ctypes.data_as(C.POINTER(C.c_double))) == 2
    np.testing.assert_allclose(np.sort(out[:2]), [1.0, 2.0], rtol=1e-14)


@pytest.mark.parametrize("dogleg_type", [0, 1])
@pytest.mark.parametrize("shared_free", [0, 7])
def test_dogleg_reaches_the_levenberg_marquardt_minimum(dogleg_type, shared_free):
    prob, ph = synth.make_phong_problem(50, 2000)
    init = "perturbed" if shared_free else "truth"
    kw = dict(num_threads=4, use_nonmonotonic_steps=0)
    s_lm, _ = _oracle(prob, ph, shared_free, init).solve(orc.driver_options(**kw))
    op = _oracle(prob, ph, shared_free, init)
    s, log = op.solve(orc.driver_options(trust_region_strategy_type=1, dogleg_type=dogleg_type, **kw))
    assert s.termination_type == 0
    assert s.final_cost == pytest.approx(s_lm.final_cost, rel=1e-4)
    # the trust region only ever halves on a rejected step and the radius sequence starts at 1e4
    assert log["trust_region_radius"][0] == 1e4
    assert np.abs(np.linalg.norm(op.normals, axis=1) - 1).max() < 1e-12


def test_subspace_dogleg_step_is_never_worse_than_the_traditional_one():
    """On the trust-region boundary the 2-D subspace minimiser contains the dogleg path, so its model
    decrease is at least the traditional one; inside the region both take the Gauss-Newton step."""
    prob, ph = synth.make_phong_problem(12, 200, track_len=6, seed=5)
    for radius in (1e4, 30.0, 3.0, 0.3):
        costs = []
        for dt in (0, 1):
            op = _oracle(prob, ph, 7)
            s, log = op.solve(orc.driver_options(num_threads=2, trust_region_strategy_type=1, dogleg_type=dt, max_num_iterations=1,
                                                 initial_trust_region_radius=radius, use_nonmonotonic_steps=0))
            costs.append(log["cost"][1] if len(log["cost"]) > 1 else log["cost"][0])
        if radius >= 1e4:
            assert costs[0] == pytest.approx(costs[1], rel=1e-12)

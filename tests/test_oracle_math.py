"""Pins the oracle's L1/L2 arithmetic against the reference's own constants and
identities (SURVEY.md section 8(c) "Fixtures / known answers") and against
complex-step / finite-difference derivatives.  CPU only."""
import numpy as np
import pytest

import np_reference as npr
from ceres_slam_amd import synth
from oracle import oracle as orc

CAM = synth.KITTI_CAMERA          # /root/reference tests/camera_test.cpp:11-15


def test_camera_round_trip_known_answer():
    # tests/camera_test.cpp:25  obs = (60, 71, 12): triangulate then project must
    # return the observation; the camera-frame point is the value SURVEY.md
    # section 4 records from the reference formulas.
    obs = np.array([60.0, 71.0, 12.0])
    q = orc.triangulate(CAM, obs)
    np.testing.assert_allclose(q, [-24.1639199, -4.9992438, 31.5307171], rtol=0, atol=5e-8)
    np.testing.assert_allclose(orc.project(CAM, q), obs, rtol=0, atol=1e-12)


def test_camera_jacobians_are_mutual_inverses_and_match_fd():
    obs = np.array([60.0, 71.0, 12.0])
    q, Jt = orc.triangulate(CAM, obs, jac=True)
    _, Jp = orc.project(CAM, q, jac=True)
    np.testing.assert_allclose(Jp @ Jt, np.eye(3), atol=1e-12)
    for i in range(3):
        h = 1e-6
        dq = np.zeros(3)
        dq[i] = h
        fd = (orc.project(CAM, q + dq) - orc.project(CAM, q - dq)) / (2 * h)
        np.testing.assert_allclose(Jp[:, i], fd, rtol=1e-8, atol=1e-9)


def test_so3_exp_branches_and_orthogonality():
    rng = np.random.default_rng(0)
    for _ in range(20):
        phi = rng.normal(size=3)
        R = orc.so3_exp(phi)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert abs(np.linalg.det(R) - 1) < 1e-14
        np.testing.assert_allclose(R, npr.so3_exp(phi), atol=1e-15)
        np.testing.assert_allclose(R, synth.so3_exp(phi), atol=1e-15)
    # first-order branch (so3group.hpp:277-280): angle <= DBL_EPSILON -> I + phi^
    phi = np.array([1e-17, -2e-17, 5e-18])
    np.testing.assert_array_equal(orc.so3_exp(phi), np.eye(3) + npr.wedge(phi))
    np.testing.assert_array_equal(orc.so3_exp(np.zeros(3)), np.eye(3))
    # rotation about z by 90 degrees
    np.testing.assert_allclose(orc.so3_exp([0, 0, np.pi / 2]), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)


def test_se3_plus_is_exp_times_T_not_true_se3_exp():
    # perturbations.hpp:61-62 + se3group.hpp:323-325,176-183: R+ = E R, t+ = E t + rho
    rng = np.random.default_rng(1)
    T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3)))
    eps = rng.normal(size=6) * 0.3
    E = synth.so3_exp(eps[3:])
    t, R = synth.pose_unpack(T)
    expect = synth.pose_pack(E @ t + eps[:3], E @ R)
    np.testing.assert_allclose(orc.se3_plus(T, eps), expect, atol=1e-15)
    np.testing.assert_allclose(orc.se3_plus(T, eps), npr.se3_plus(T, eps), atol=1e-15)
    np.testing.assert_array_equal(orc.se3_plus(T, np.zeros(6)), T)


def test_se3_inverse_and_transform_identities():
    # identities exercised by tests/geometry_test.cpp: T * T^-1 = I, transform round trip
    rng = np.random.default_rng(2)
    T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3)))
    Ti = orc.se3_inverse(T)
    p = rng.normal(size=3)
    np.testing.assert_allclose(orc.se3_transform(Ti, orc.se3_transform(T, p)), p, atol=1e-14)
    t, R = synth.pose_unpack(T)
    np.testing.assert_allclose(orc.se3_transform(T, p), R @ p + t, atol=1e-15)
    np.testing.assert_allclose(orc.se3_inverse(Ti), T, atol=1e-15)


@pytest.mark.parametrize("full_stiffness", [False, True])
def test_stereo_residual_and_local_jacobians(full_stiffness):
    rng = np.random.default_rng(3)
    for trial in range(10):
        T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3) * 0.5))
        q = np.array([rng.uniform(-5, 5), rng.uniform(-2, 2), rng.uniform(4, 30)])
        t, R = synth.pose_unpack(T)
        p = R.T @ (q - t)
        z = synth.project(CAM, q) + rng.normal(size=3)
        S = np.diag([0.5, 0.5, 0.5])
        if full_stiffness:
            A = rng.normal(size=(3, 3))
            S = np.linalg.inv(np.linalg.cholesky(A @ A.T + np.eye(3))).T  # arbitrary full 3x3
        r, Jp, Jl = orc.stereo_residual(CAM, T, p, z, S, jac=True)
        np.testing.assert_allclose(r, npr.residual_global(CAM, T, p, z, S), atol=1e-11)
        np.testing.assert_allclose(r, orc.stereo_residual(CAM, T, p, z, S), atol=0)
        # autodiff route: complex step through the reference's Plus at eps = 0
        Jp_cs, Jl_cs = npr.jacobians_complex_step(CAM, T, p, z, S)
        np.testing.assert_allclose(Jp, Jp_cs, rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(Jl, Jl_cs, rtol=1e-12, atol=1e-10)
        # central differences through the oracle's own Plus (full Rodrigues branch)
        h = 1e-6
        for c in range(6):
            e = np.zeros(6)
            e[c] = h
            fd = (orc.stereo_residual(CAM, orc.se3_plus(T, e), p, z, S) - orc.stereo_residual(CAM, orc.se3_plus(T, -e), p, z, S)) / (2 * h)
            np.testing.assert_allclose(Jp[:, c], fd, rtol=2e-6, atol=1e-5)


def test_jacobian_equals_ceres_chain_global_times_plus_jacobian():
    # Ceres forms J_local = J_global(3x12) * PlusJacobian(12x6).  Global Jacobian
    # pattern: se3group.hpp:199-206.  Plus-Jacobian at eps=0: d/d eps of
    # [ (I+phi^) t + rho ; vec_rowmajor((I+phi^) R) ].
    rng = np.random.default_rng(4)
    T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3) * 0.5))
    t, R = synth.pose_unpack(T)
    q = np.array([1.0, -0.5, 12.0])
    p = R.T @ (q - t)
    z = synth.project(CAM, q)
    S = np.diag([0.5, 0.5, 0.5])
    _, Jpi = orc.project(CAM, q, jac=True)
    Jtrans = np.zeros((3, 12))
    Jtrans[:, :3] = np.eye(3)
    for i in range(3):
        Jtrans[i, 3 + 3 * i: 6 + 3 * i] = p
    Jglobal = S @ Jpi @ Jtrans
    plusJ = np.zeros((12, 6))
    h = 1e-30
    for c in range(6):
        e = np.zeros(6, dtype=complex)
        e[c] = 1j * h
        plusJ[:, c] = npr.se3_plus(T.astype(complex), e).imag / h
    _, Jp, _ = orc.stereo_residual(CAM, T, p, z, S, jac=True)
    np.testing.assert_allclose(Jp, Jglobal @ plusJ, rtol=1e-12, atol=1e-11)


def test_jet_restatement_of_the_autodiff_block_matches_the_closed_forms():
    """The 15-lane Jet evaluation of the functor + the 12x6 Plus Jacobian (what AutoDiffCostFunction<..., 3, 12, 3> and
    AutoDiffLocalParameterization<SE3Perturbation, 12, 6> compute) against the closed-form block, and a whole solve run
    in that mode against the default one."""
    rng = np.random.default_rng(11)
    for trial in range(20):
        T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3) * 0.7))
        q = np.array([rng.uniform(-5, 5), rng.uniform(-2, 2), rng.uniform(4, 30)])
        t, R = synth.pose_unpack(T)
        p = R.T @ (q - t)
        z = synth.project(CAM, q) + rng.normal(size=3)
        A = rng.normal(size=(3, 3))
        S = np.linalg.inv(np.linalg.cholesky(A @ A.T + np.eye(3))).T
        r, Jp, Jl = orc.stereo_residual(CAM, T, p, z, S, jac=True)
        r2, Jp2, Jl2 = orc.stereo_residual(CAM, T, p, z, S, jac=True, autodiff=True)
        np.testing.assert_allclose(r2, r, rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(Jp2, Jp, rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(Jl2, Jl, rtol=1e-12, atol=1e-10)
    prob = synth.make_problem(12, 300, track_len=6, seed=5)
    s1, log1 = orc.OracleProblem.from_synth(prob).solve(orc.driver_options(num_threads=2))
    orc.set_jacobian_mode(1)
    try:
        s2, log2 = orc.OracleProblem.from_synth(prob).solve(orc.driver_options(num_threads=2))
    finally:
        orc.set_jacobian_mode(0)
    assert s2.num_iterations == s1.num_iterations
    np.testing.assert_allclose(log2["cost"], log1["cost"], rtol=1e-10)


def test_huber_rho_and_corrector_gradient_identity():
    a = 1.345                                  # scripts/ba_all_devon.sh:86
    for s in (0.0, 0.5, a * a, a * a + 1e-9, 10.0, 1e4):
        rho = orc.huber(a, s)
        np.testing.assert_allclose(rho, npr.huber(a, s), rtol=1e-15)
    assert orc.huber(a, 1.0).tolist() == [1.0, 1.0, 0.0]
    rho = orc.huber(a, 100.0)
    np.testing.assert_allclose(rho[0], 2 * a * 10 - a * a)
    np.testing.assert_allclose(rho[1], a / 10)
    np.testing.assert_allclose(rho[2], -rho[1] / 200)

"""Pins the oracle's Phong-lighting rows against the reference's
light_test constants, the shading guards, and complex-step Jacobians of an independent numpy
restatement of the forward formulas.  CPU only."""
import numpy as np
import pytest

import np_reference as npr
from ceres_slam_amd import synth
from oracle import oracle as orc

# /root/reference tests/light_test.cpp:26-50 : material (ka, ks, alpha), texture, vertices, light
KA, KS, ALPHA, KD = 0.1, 0.3, 10.0, 0.6
V28 = (np.array([0.823015, 0.60803428, 0.0]), np.array([0.0, 0.0, 1.0]))
V245 = (np.array([0.08868649, 1.0, 0.7597348]), np.array([0.0, -1.0, 0.0]))
LIGHT = np.array([-2.0, -2.0, 2.0])


def test_light_test_scene_known_answers():
    # values of the reference formulas on its own light_test scene:
    # shade(v28) = 0.27697118 (diffuse only), shade(v245) = 0.48917229 (0.46630139 + 0.02287090)
    assert orc.light_shade(orc.POINT_LIGHT, *V28, LIGHT, KD, KS, ALPHA) == pytest.approx(0.27697118, abs=5e-9)
    assert orc.light_shade(orc.POINT_LIGHT, *V245, LIGHT, KD, KS, ALPHA) == pytest.approx(0.48917229, abs=5e-9)
    # diffuse / specular split of v245
    d = orc.light_shade(orc.POINT_LIGHT, *V245, LIGHT, KD, 0.0, ALPHA)
    s = orc.light_shade(orc.POINT_LIGHT, *V245, LIGHT, 0.0, KS, ALPHA)
    assert d == pytest.approx(0.46630139, abs=5e-9) and s == pytest.approx(0.02287090, abs=5e-9)
    # ambient is disabled (phong.hpp:33): ka never enters
    for p, n in (V28, V245):
        ell = (LIGHT - p) / np.linalg.norm(LIGHT - p)
        cd = -p / np.linalg.norm(p)
        assert orc.phong_shade(n, ell, cd, KD, KS, ALPHA) == pytest.approx(
            orc.light_shade(orc.POINT_LIGHT, p, n, LIGHT, KD, KS, ALPHA), rel=1e-15)
        assert orc.phong_shade(n, ell, cd, KD, KS, ALPHA) == pytest.approx(npr.phong_shade(n, ell, cd, KD, KS, ALPHA).real, rel=1e-14)


def test_shading_guards_and_clamp():
    n = np.array([0.0, 0.0, 1.0])
    cd = np.array([0.0, 0.6, 0.8])
    # light behind the surface: diffuse guard (phong.hpp:68-70) -> only specular could remain
    below = np.array([0.0, 0.6, -0.8])
    assert orc.phong_shade(n, below, cd, 0.9, 0.0, 5.0) == 0.0
    # mirror direction facing away from the camera: specular guard (phong.hpp:96-98)
    ell = np.array([0.0, 0.6, 0.8])
    assert orc.phong_shade(n, ell, -cd, 0.0, 0.7, 5.0) == 0.0
    # clamp to [0, 1] (phong.hpp:136-139)
    assert orc.phong_shade(n, n, n, 5.0, 5.0, 1.0) == 1.0
    assert orc.phong_shade(n, ell, cd, 0.5, 0.2, 3.0) == pytest.approx(0.5 * 0.8 + 0.2 * (np.array([0, -0.6, 0.8]) @ cd) ** 3.0)


def test_unit_vector_plus():
    rng = np.random.default_rng(0)
    x = rng.normal(size=3)
    x /= np.linalg.norm(x)
    d = rng.normal(size=3) * 0.1
    y = orc.unit_vector_plus(x, d)
    assert abs(np.linalg.norm(y) - 1) < 1e-15
    np.testing.assert_allclose(y, npr.unit_vector_plus(x, d), atol=1e-15)
    np.testing.assert_allclose(orc.unit_vector_plus(x, np.zeros(3)), x, atol=1e-16)
    np.testing.assert_allclose(orc.unit_vector_plus(x, 3.0 * x), x, atol=1e-15)   # radial part is removed


def _random_scene(rng, light_type):
    T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3) * 0.4))
    t, R = synth.pose_unpack(T)
    q = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(4, 15)])
    p = R.T @ (q - t)
    nc = -q / np.linalg.norm(q) + rng.normal(size=3) * 0.3         # roughly facing the camera
    nc /= np.linalg.norm(nc)
    n = R.T @ nc
    if light_type == 0:
        light = R.T @ (np.array([rng.uniform(-2, 2), rng.uniform(-3, -1), rng.uniform(0, 3)]) - t)
    else:
        lc = nc + rng.normal(size=3) * 0.3
        light = R.T @ (lc / np.linalg.norm(lc))
    phong = np.array([rng.uniform(0, 1), rng.uniform(0.1, 0.5), rng.uniform(1, 20)])
    return T, p, n, phong, rng.uniform(0.2, 0.9), light


@pytest.mark.parametrize("light_type", [0, 1])
def test_intensity_residual_and_jacobian_match_complex_step(light_type):
    rng = np.random.default_rng(10 + light_type)
    active = 0
    for _ in range(40):
        T, p, n, phong, kd, light = _random_scene(rng, light_type)
        colour, stiff = rng.uniform(0, 1), 100.0      # int_var 1e-4 -> stiffness 100
        r, J = orc.intensity_residual(light_type, T, p, n, phong, kd, light, colour, stiff, jac=True)
        r2 = npr.intensity_residual(light_type, T, p, n, phong, kd, light, colour, stiff).real
        assert r == pytest.approx(r2, rel=1e-12, abs=1e-12)
        assert r == orc.intensity_residual(light_type, T, p, n, phong, kd, light, colour, stiff)
        J2 = npr.intensity_jacobian_complex_step(light_type, T, p, n, phong, kd, light, colour, stiff)
        np.testing.assert_allclose(J, J2, rtol=1e-9, atol=1e-9)
        assert J[12] == 0.0                           # d/d ka: ambient disabled
        active += int(np.abs(J[13:15]).max() > 0)
    assert active > 5                                 # the specular branch was exercised


def test_intensity_jacobian_is_zero_where_the_clamp_is_active():
    rng = np.random.default_rng(3)
    T, p, n, phong, kd, light = _random_scene(rng, 0)
    r, J = orc.intensity_residual(0, T, p, n, np.array([0, 5.0, 1.0]), 5.0, light, 0.5, 10.0, jac=True)
    assert r == pytest.approx(10.0 * (1.0 - 0.5)) and np.all(J == 0)


def test_normal_residual_and_jacobians():
    rng = np.random.default_rng(4)
    for _ in range(10):
        T = synth.pose_pack(rng.normal(size=3), synth.so3_exp(rng.normal(size=3)))
        t, R = synth.pose_unpack(T)
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        n_obs = R @ n + rng.normal(size=3) * 0.01
        S = np.eye(3) * 100.0
        r, Jp, Jn = orc.normal_residual(T, n, n_obs, S, jac=True)
        np.testing.assert_allclose(r, S @ (R @ n - n_obs), atol=1e-12)      # normal_error.hpp:37-38
        h = 1e-30
        for k in range(6):
            e = np.zeros(6, dtype=complex); e[k] = 1j * h
            Tn = npr.se3_plus(T.astype(complex), e)
            np.testing.assert_allclose(Jp[:, k], (S @ (Tn[3:].reshape(3, 3) @ n - n_obs)).imag / h, atol=1e-9)
        for k in range(3):
            e = np.zeros(3, dtype=complex); e[k] = 1j * h
            np.testing.assert_allclose(Jn[:, k], (S @ (R @ npr.unit_vector_plus(n.astype(complex), e) - n_obs)).imag / h, atol=1e-9)
        assert np.all(Jp[:, :3] == 0)       # a direction does not see the translation

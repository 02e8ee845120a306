"""Independent numpy/scipy restatement used to cross-check the C oracle.

Deliberately written a different way from oracle/ssba_oracle.c: residuals are
evaluated through the reference's *global* 12+3 parameterisation and the
Jacobians come from complex-step differentiation through ``Plus`` (the route
Ceres's AutoDiffCostFunction + AutoDiffLocalParameterization take), the damped
normal equations are solved WITHOUT a Schur complement by a sparse direct solve,
and the trust-region loop is re-implemented on top of that.

Citations are into /root/reference.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(np.float64).eps


def wedge(phi):
    # so3group.hpp:248-254
    return np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]], dtype=phi.dtype)


def so3_exp(phi):
    # so3group.hpp:273-291; complex-safe (norm via sqrt of sum of squares)
    angle = np.sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2])
    if abs(angle) <= EPS:
        return np.eye(3, dtype=phi.dtype) + wedge(phi)
    a = phi / angle
    return np.cos(angle) * np.eye(3) + (1 - np.cos(angle)) * np.outer(a, a) + np.sin(angle) * wedge(a)


def se3_plus(T, eps):
    # perturbations.hpp:61-62 with se3group.hpp:176-183,323-325
    t, R = T[:3], T[3:].reshape(3, 3)
    E = so3_exp(eps[3:])
    return np.concatenate([E @ t + eps[:3], (E @ R).reshape(9)])


def residual_global(cam, T, p, z, S):
    # stereo_reprojection_error.hpp:38-50 ; stereo_camera.hpp:79-84
    t, R = T[:3], T[3:].reshape(3, 3)
    q = R @ p + t
    iz = 1.0 / q[2]
    pred = np.array([cam["fu"] * q[0] * iz + cam["cu"], cam["fv"] * q[1] * iz + cam["cv"], cam["fu"] * cam["b"] * iz])
    return S @ (pred - z)


def jacobians_complex_step(cam, T, p, z, S, h=1e-30):
    """Local Jacobians d r / d eps (3x6) and d r / d p (3x3) by complex step.

    At eps = 0 the first-order branch of so3_exp is taken, exactly as when Ceres
    evaluates the plus-Jacobian with Jets.
    """
    Jp, Jl = np.zeros((3, 6)), np.zeros((3, 3))
    Tc, pc = T.astype(complex), p.astype(complex)
    for c in range(6):
        e = np.zeros(6, dtype=complex)
        e[c] = 1j * h
        Jp[:, c] = residual_global(cam, se3_plus(Tc, e), pc, z, S).imag / h
    for c in range(3):
        d = pc.copy()
        d[c] += 1j * h
        Jl[:, c] = residual_global(cam, Tc, d, z, S).imag / h
    return Jp, Jl


def huber(a, s):
    # [Ceres loss_function.cc]
    b = a * a
    if s > b:
        r = np.sqrt(s)
        rho1 = max(np.finfo(np.float64).tiny, a / r)
        return 2 * a * r - b, rho1, -rho1 / (2 * s)
    return s, 1.0, 0.0


class NumpyBA:
    """Whole-problem evaluation with vectorised closed forms + sparse algebra."""

    def __init__(self, cam, poses, points, obs_pose, obs_point, obs_uvd, S, pose_const=None, huber_a=0.0):
        self.cam = cam
        self.poses, self.points = poses.copy(), points.copy()
        self.k, self.j = obs_pose.astype(np.int64), obs_point.astype(np.int64)
        self.z, self.S = obs_uvd, np.asarray(S).reshape(3, 3)
        P, L = poses.shape[0], points.shape[0]
        if pose_const is None:
            pose_const = np.zeros(P, dtype=bool)
            pose_const[0] = True
        seen = np.bincount(self.k, minlength=P) > 0
        self.free = (~np.asarray(pose_const, dtype=bool)) & seen
        self.free_idx = np.full(P, -1)
        self.free_idx[self.free] = np.arange(self.free.sum())
        self.active = np.bincount(self.j, minlength=L) > 0
        self.nf = int(self.free.sum())
        self.huber_a = huber_a

    def residuals(self, poses, points, jac=False):
        cam = self.cam
        t, R = poses[self.k, :3], poses[self.k, 3:].reshape(-1, 3, 3)
        p = points[self.j]
        q = np.einsum("nij,nj->ni", R, p) + t
        iz = 1.0 / q[:, 2]
        pred = np.stack([cam["fu"] * q[:, 0] * iz + cam["cu"], cam["fv"] * q[:, 1] * iz + cam["cv"], cam["fu"] * cam["b"] * iz], 1)
        r = (pred - self.z) @ self.S.T
        sq = (r * r).sum(1)
        if self.huber_a > 0:
            a, b = self.huber_a, self.huber_a ** 2
            out = sq > b
            rho0 = np.where(out, 2 * a * np.sqrt(np.where(out, sq, 1.0)) - b, sq)
            rho1 = np.where(out, a / np.sqrt(np.where(out, sq, 1.0)), 1.0)
            cost = 0.5 * rho0.sum()
        else:
            rho1 = np.ones_like(sq)
            cost = 0.5 * sq.sum()
        if not jac:
            return cost, r * np.sqrt(rho1)[:, None]
        N = q.shape[0]
        Jpi = np.zeros((N, 3, 3))
        Jpi[:, 0, 0] = cam["fu"] * iz
        Jpi[:, 0, 2] = -cam["fu"] * q[:, 0] * iz * iz
        Jpi[:, 1, 1] = cam["fv"] * iz
        Jpi[:, 1, 2] = -cam["fv"] * q[:, 1] * iz * iz
        Jpi[:, 2, 2] = -cam["fu"] * cam["b"] * iz * iz
        A = np.einsum("ij,njk->nik", self.S, Jpi)
        G = np.zeros((N, 3, 6))
        G[:, 0, 0] = G[:, 1, 1] = G[:, 2, 2] = 1.0
        G[:, 0, 4], G[:, 0, 5] = q[:, 2], -q[:, 1]
        G[:, 1, 3], G[:, 1, 5] = -q[:, 2], q[:, 0]
        G[:, 2, 3], G[:, 2, 4] = q[:, 1], -q[:, 0]
        Jp = np.einsum("nij,njk->nik", A, G)
        Jl = np.einsum("nij,njk->nik", A, R)
        # Huber (rho'' <= 0 always): scale by sqrt(rho')  [Ceres corrector.cc]
        s1 = np.sqrt(rho1)
        return cost, r * s1[:, None], Jp * s1[:, None, None], Jl * s1[:, None, None]

    def sparse_jacobian(self, Jp, Jl):
        """J over [free poses (6 each) | all landmarks (3 each)] as CSR."""
        N = Jp.shape[0]
        nf, L = self.nf, self.points.shape[0]
        rows = np.arange(3 * N).reshape(N, 3)
        f = self.free_idx[self.k]
        m = f >= 0
        rp = np.repeat(rows[m][:, :, None], 6, axis=2).reshape(-1)
        cp = (6 * f[m][:, None, None] + np.arange(6)[None, None, :]).repeat(3, axis=1).reshape(-1)
        vp = Jp[m].reshape(-1)
        rl = np.repeat(rows[:, :, None], 3, axis=2).reshape(-1)
        cl = (6 * nf + 3 * self.j[:, None, None] + np.arange(3)[None, None, :]).repeat(3, axis=1).reshape(-1)
        vl = Jl.reshape(-1)
        J = sp.csr_matrix((np.concatenate([vp, vl]), (np.concatenate([rp, rl]), np.concatenate([cp, cl]))),
                          shape=(3 * N, 6 * nf + 3 * L))
        return J

    def lm_step(self, poses, points, radius, scale=None, min_diag=1e-6, max_diag=1e32):
        """One Ceres LM step; returns (dp (P,6), dl (L,3), model_cost_change, scale)."""
        cost, r, Jp, Jl = self.residuals(poses, points, jac=True)
        J = self.sparse_jacobian(Jp, Jl)
        nf, L = self.nf, points.shape[0]
        keep = np.concatenate([np.ones(6 * nf, bool), np.repeat(self.active, 3)])
        J = J[:, keep]
        colsq = np.asarray(J.multiply(J).sum(0)).ravel()
        if scale is None:
            scale = 1.0 / (1.0 + np.sqrt(colsq))
        Js = J @ sp.diags(scale)
        diag = np.clip(np.asarray(Js.multiply(Js).sum(0)).ravel(), min_diag, max_diag)
        H = (Js.T @ Js + sp.diags(diag / radius)).tocsc()
        y = spla.spsolve(H, Js.T @ r.reshape(-1))
        delta = -y * scale
        Jd = J @ delta
        mcc = -Jd @ (r.reshape(-1) + 0.5 * Jd)
        full = np.zeros(6 * nf + 3 * L)
        full[keep] = delta
        dp = np.zeros((poses.shape[0], 6))
        dp[self.free] = full[: 6 * nf].reshape(nf, 6)
        dl = full[6 * nf:].reshape(L, 3)
        return dp, dl, mcc, scale, cost

    def plus(self, poses, points, dp, dl):
        out = poses.copy()
        for k in np.nonzero(self.free)[0]:
            out[k] = se3_plus(poses[k], dp[k])
        pts = points.copy()
        pts[self.active] += dl[self.active]
        return out, pts

    def solve(self, max_iter=1000, nonmonotonic=True, f_tol=1e-6, p_tol=1e-8, min_rd=1e-3):
        """Independent re-implementation of the trust-region loop (cost log only)."""
        x_p, x_l = self.poses.copy(), self.points.copy()
        radius, dec = 1e4, 2.0
        scale = None
        max_nm = 5 if nonmonotonic else 0
        cost0, _ = self.residuals(x_p, x_l)
        x_cost = cost0
        minimum = current = reference = candidate = x_cost
        acc_ref = acc_cand = 0.0
        n_nm = 0
        log = [(x_cost, False)]
        xs = np.concatenate([x_p[self.free].ravel(), x_l[self.active].ravel()])
        x_norm = np.linalg.norm(xs)
        for it in range(1, max_iter + 1):
            dp, dl, mcc, scale, _ = self.lm_step(x_p, x_l, radius, scale)
            if not (mcc > 0):
                radius /= dec
                dec *= 2
                log.append((x_cost, False))
                continue
            c_p, c_l = self.plus(x_p, x_l, dp, dl)
            c_cost, _ = self.residuals(c_p, c_l)
            step = np.sqrt(((c_p[self.free] - x_p[self.free]) ** 2).sum() + ((c_l[self.active] - x_l[self.active]) ** 2).sum())
            if step <= p_tol * (x_norm + p_tol):
                break
            if abs(x_cost - c_cost) <= f_tol * x_cost:
                break
            rd = max((current - c_cost) / mcc, (reference - c_cost) / (acc_ref + mcc))
            if rd > min_rd:
                x_p, x_l, x_cost = c_p, c_l, c_cost
                xs = np.concatenate([x_p[self.free].ravel(), x_l[self.active].ravel()])
                x_norm = np.linalg.norm(xs)
                radius = min(1e16, radius / max(1 / 3, 1 - (2 * rd - 1) ** 3))
                dec = 2.0
                current = c_cost
                acc_cand += mcc
                acc_ref += mcc
                if current < minimum:
                    minimum, n_nm, candidate, acc_cand = current, 0, current, 0.0
                else:
                    n_nm += 1
                    if current > candidate:
                        candidate, acc_cand = current, 0.0
                if n_nm == max_nm:
                    reference, acc_ref = candidate, acc_cand
                log.append((x_cost, True))
            else:
                radius /= dec
                dec *= 2
                log.append((c_cost, False))
        return x_p, x_l, log


# ---- Phong lighting rows: forward formulas only (complex-step differentiable) ----------------
def _fmax0(col):      # utils/utils.hpp:16-19 with a = 0
    return 0.0 * col if 0.0 >= col.real else col


def _fmin1(col):      # utils/utils.hpp:22-25 with a = 1
    return 1.0 + 0.0 * col if 1.0 <= col.real else col


def phong_shade(nc, ell, cd, kd, ks, alpha):
    """lighting/phong.hpp:25-51,59-104,136-139 (ambient = 0)."""
    nc, ell, cd = (np.asarray(v, dtype=complex) for v in (nc, ell, cd))
    diffuse = 0.0
    ldn = ell @ nc
    if np.all(np.isfinite(ell)) and not (ldn.real <= 0):
        diffuse = kd * ldn
    specular = 0.0
    mt = 2 * ldn * nc - ell
    if not ((mt @ mt).real <= 0):
        m = mt / np.sqrt(mt @ mt)
        s = m @ cd
        if not (s.real <= 0):
            specular = ks * s ** alpha
    col = 1.0 * (0.0 + diffuse + specular) + 0j
    return _fmin1(_fmax0(col))


def unit_vector_plus(x, delta):
    """perturbations.hpp:98-102"""
    y = x + delta - (delta @ x) / (x @ x) * x
    return y / np.sqrt(y @ y)


def intensity_residual(light_type, T, p, n, phong, kd, light, colour, stiffness):
    """intensity_error_point_light.hpp:24-96 / intensity_error_directional_light.hpp:24-96"""
    t, R = T[:3], T[3:].reshape(3, 3)
    q, nc = R @ p + t, R @ n
    if light_type == 0:
        v = (R @ light + t) - q                       # point_light.hpp:79-81
        ell = v / np.sqrt(v @ v)
    else:
        d = R @ light                                 # directional_light.hpp:32-35 normalises
        ell = d / np.sqrt(d @ d)
    cd = -q / np.sqrt(q @ q)
    return stiffness * (phong_shade(nc, ell, cd, kd, phong[1], phong[2]) - colour)


def intensity_jacobian_complex_step(light_type, T, p, n, phong, kd, light, colour, stiffness, h=1e-30):
    """[pose(6) | point(3) | normal(3) | phong(3) | texture(1) | light(3)] local Jacobian the way
    Ceres forms it: through SE3Perturbation / UnitVectorPerturbation Plus at delta = 0."""
    J = np.zeros(19)
    c = lambda a: np.asarray(a, dtype=complex)
    T, p, n, phong, light = c(T), c(p), c(n), c(phong), c(light)
    base = dict(light_type=light_type, T=T, p=p, n=n, phong=phong, kd=kd + 0j, light=light, colour=colour, stiffness=stiffness)

    def f(**kw):
        a = dict(base)
        a.update(kw)
        return intensity_residual(**a).imag / h
    for k in range(6):
        e = np.zeros(6, dtype=complex); e[k] = 1j * h
        J[k] = f(T=se3_plus(T, e))
    for k in range(3):
        e = np.zeros(3, dtype=complex); e[k] = 1j * h
        J[6 + k] = f(p=p + e)
        J[9 + k] = f(n=unit_vector_plus(n, e))
        J[12 + k] = f(phong=phong + e)
        J[16 + k] = f(light=(light + e) if light_type == 0 else unit_vector_plus(light, e))
    J[15] = f(kd=kd + 1j * h)
    return J


# ---- traditional dogleg [Ceres 1.x dogleg_strategy.cc], written directly in the Jacobi-scaled
# ---- space the way Ceres does (the C oracle works in unscaled coordinates)
def dogleg_solve(ba: "NumpyBA", max_iter=1000, nonmonotonic=True, f_tol=1e-6, p_tol=1e-8, min_rd=1e-3,
                 min_diag=1e-6, max_diag=1e32):
    x_p, x_l = ba.poses.copy(), ba.points.copy()
    radius, mu, reuse = 1e4, 1e-8, False
    scale = None
    max_nm = 5 if nonmonotonic else 0
    x_cost, _ = ba.residuals(x_p, x_l)
    minimum = current = reference = candidate = x_cost
    acc_ref = acc_cand = 0.0
    n_nm = 0
    log = [(x_cost, False)]
    nf, L = ba.nf, x_l.shape[0]
    keep = np.concatenate([np.ones(6 * nf, bool), np.repeat(ba.active, 3)])
    xs = np.concatenate([x_p[ba.free].ravel(), x_l[ba.active].ravel()])
    x_norm = np.linalg.norm(xs)
    state = {}
    for it in range(1, max_iter + 1):
        if not reuse:
            reuse = True
            cost, r, Jp, Jl = ba.residuals(x_p, x_l, jac=True)
            J = ba.sparse_jacobian(Jp, Jl)[:, keep]
            if scale is None:
                scale = 1.0 / (1.0 + np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel()))
            Js = J @ sp.diags(scale)
            rv = r.reshape(-1)
            D = np.sqrt(np.clip(np.asarray(Js.multiply(Js).sum(0)).ravel(), min_diag, max_diag))
            grad = (Js.T @ rv) / D                                   # ComputeGradient
            Jg = Js @ (grad / D)
            alpha = (grad @ grad) / (Jg @ Jg)                        # ComputeCauchyPoint
            H = (Js.T @ Js + sp.diags(mu * D * D)).tocsc()           # ComputeGaussNewtonStep
            gn = -D * spla.spsolve(H, Js.T @ rv)
            state = dict(Js=Js, rv=rv, D=D, grad=grad, alpha=alpha, gn=gn)
        D, grad, alpha, gn, Js, rv = (state[k] for k in ("D", "grad", "alpha", "gn", "Js", "rv"))
        gnorm, nnorm = np.linalg.norm(grad), np.linalg.norm(gn)
        if nnorm <= radius:
            step = gn.copy()
        elif gnorm * alpha >= radius:
            step = -(radius / gnorm) * grad
        else:
            b_dot_a = -alpha * (grad @ gn)
            a_sq = (alpha * gnorm) ** 2
            bma = a_sq - 2 * b_dot_a + nnorm ** 2
            c = b_dot_a - a_sq
            d = np.sqrt(c * c + bma * (radius ** 2 - a_sq))
            beta = (d - c) / bma if c <= 0 else (radius ** 2 - a_sq) / (d + c)
            step = (-alpha * (1 - beta)) * grad + beta * gn
        step_norm_scaled = np.linalg.norm(step)
        step_js = step / D
        model = Js @ step_js
        mcc = -model @ (rv + 0.5 * model)
        if not (mcc > 0):
            mu *= 10.0
            reuse = False
            log.append((x_cost, False))
            continue
        delta = np.zeros(6 * nf + 3 * L)
        delta[keep] = step_js * scale
        dp = np.zeros((x_p.shape[0], 6))
        dp[ba.free] = delta[: 6 * nf].reshape(nf, 6)
        dl = delta[6 * nf:].reshape(L, 3)
        c_p, c_l = ba.plus(x_p, x_l, dp, dl)
        c_cost, _ = ba.residuals(c_p, c_l)
        stepn = np.sqrt(((c_p[ba.free] - x_p[ba.free]) ** 2).sum() + ((c_l[ba.active] - x_l[ba.active]) ** 2).sum())
        if stepn <= p_tol * (x_norm + p_tol):
            break
        if abs(x_cost - c_cost) <= f_tol * x_cost:
            break
        rd = max((current - c_cost) / mcc, (reference - c_cost) / (acc_ref + mcc))
        if rd > min_rd:
            x_p, x_l, x_cost = c_p, c_l, c_cost
            xs = np.concatenate([x_p[ba.free].ravel(), x_l[ba.active].ravel()])
            x_norm = np.linalg.norm(xs)
            if rd < 0.25:
                radius *= 0.5
            if rd > 0.75:
                radius = max(radius, 3.0 * step_norm_scaled)
            mu = max(1e-8, 2.0 * mu / 10.0)
            reuse = False
            current = c_cost
            acc_cand += mcc
            acc_ref += mcc
            if current < minimum:
                minimum, n_nm, candidate, acc_cand = current, 0, current, 0.0
            else:
                n_nm += 1
                if current > candidate:
                    candidate, acc_cand = current, 0.0
            if n_nm == max_nm:
                reference, acc_ref = candidate, acc_cand
            log.append((x_cost, True))
        else:
            radius *= 0.5
            reuse = True
            log.append((c_cost, False))
    return x_p, x_l, log


# ---- config 3 (stereo + intensity + normal residual blocks, landmark block = [position | normal]) ----
def phong_lm_step(cam, poses, points, normals, obs_pose, obs_point, obs_uvd, S, ph, radius, pose_const=None,
                  min_diag=1e-6, max_diag=1e32, shared_free=0):
    """One Ceres LM step of the config-3 problem with a dense/sparse direct solve and complex-step
    Jacobians through the reference's Plus operators (slow: small problems only).
    `shared_free` frees the shared blocks (bit 0 light, bit 1 Phong parameters, bit 2 textures); their
    columns follow the landmark columns in the order light, Phong parameters, textures.
    Returns dp (P,6), dl (L,6), model_cost_change, cost [, db when shared_free]."""
    P, L, N = poses.shape[0], points.shape[0], obs_pose.shape[0]
    if pose_const is None:
        pose_const = np.zeros(P, bool)
        pose_const[0] = True
    seen = np.bincount(obs_pose, minlength=P) > 0
    free = (~pose_const) & seen
    fidx = np.full(P, -1)
    fidx[free] = np.arange(free.sum())
    active = np.bincount(obs_point, minlength=L) > 0
    nf = int(free.sum())
    M = len(ph["texture"])
    b_light = b_phong = b_tex = -1
    nb = 0
    if shared_free & 1:
        b_light, nb = nb, nb + 3
    if shared_free & 2:
        b_phong, nb = nb, nb + 3 * M
    if shared_free & 4:
        b_tex, nb = nb, nb + M
    ncol = 6 * nf + 6 * L + nb
    rows, cols, vals, r = [], [], [], np.zeros(7 * N)
    Sn = np.asarray(ph["normal_stiffness"]).reshape(3, 3)
    h = 1e-30
    for i in range(N):
        k, j = int(obs_pose[i]), int(obs_point[i])
        T, p, n = poses[k], points[j], normals[j]
        m = int(ph["material_of_point"][j])
        r[7 * i: 7 * i + 3] = residual_global(cam, T, p, obs_uvd[i], S)
        Jp3, Jl3 = jacobians_complex_step(cam, T, p, obs_uvd[i], S)
        r[7 * i + 3] = intensity_residual(ph["light_type"], T, p, n, ph["phong"][m], ph["texture"][m], ph["light"],
                                          ph["intensity"][i], ph["int_stiffness"]).real
        J19 = intensity_jacobian_complex_step(ph["light_type"], T, p, n, ph["phong"][m], ph["texture"][m], ph["light"],
                                              ph["intensity"][i], ph["int_stiffness"])
        R = T[3:].reshape(3, 3)
        r[7 * i + 4: 7 * i + 7] = Sn @ (R @ n - ph["normal_obs"][i])
        Jnp, Jnn = np.zeros((3, 6)), np.zeros((3, 3))
        for c in range(6):
            e = np.zeros(6, dtype=complex); e[c] = 1j * h
            Tn = se3_plus(T.astype(complex), e)
            Jnp[:, c] = (Sn @ (Tn[3:].reshape(3, 3) @ n - ph["normal_obs"][i])).imag / h
        for c in range(3):
            e = np.zeros(3, dtype=complex); e[c] = 1j * h
            Jnn[:, c] = (Sn @ (R @ unit_vector_plus(n.astype(complex), e) - ph["normal_obs"][i])).imag / h
        Jp = np.vstack([Jp3, J19[None, :6], Jnp])                       # 7 x 6
        Jl = np.zeros((7, 6))
        Jl[:3, :3] = Jl3
        Jl[3, :] = J19[6:12]
        Jl[4:, 3:] = Jnn
        for a in range(7):
            if fidx[k] >= 0:
                for c in range(6):
                    rows.append(7 * i + a); cols.append(6 * fidx[k] + c); vals.append(Jp[a, c])
            for c in range(6):
                rows.append(7 * i + a); cols.append(6 * nf + 6 * j + c); vals.append(Jl[a, c])
        b0 = 6 * nf + 6 * L
        if b_phong >= 0:
            for c in range(3):
                rows.append(7 * i + 3); cols.append(b0 + b_phong + 3 * m + c); vals.append(J19[12 + c])
        if b_tex >= 0:
            rows.append(7 * i + 3); cols.append(b0 + b_tex + m); vals.append(J19[15])
        if b_light >= 0:
            for c in range(3):
                rows.append(7 * i + 3); cols.append(b0 + b_light + c); vals.append(J19[16 + c])
    J = sp.csr_matrix((vals, (rows, cols)), shape=(7 * N, ncol))
    keep = np.concatenate([np.ones(6 * nf, bool), np.repeat(active, 6), np.ones(nb, bool)])
    J = J[:, keep]
    colsq = np.asarray(J.multiply(J).sum(0)).ravel()
    scale = 1.0 / (1.0 + np.sqrt(colsq))
    Js = J @ sp.diags(scale)
    diag = np.clip(np.asarray(Js.multiply(Js).sum(0)).ravel(), min_diag, max_diag)
    H = (Js.T @ Js + sp.diags(diag / radius)).tocsc()
    y = spla.spsolve(H, Js.T @ r)
    delta = -y * scale
    Jd = J @ delta
    mcc = -Jd @ (r + 0.5 * Jd)
    full = np.zeros(ncol)
    full[keep] = delta
    dp = np.zeros((P, 6))
    dp[free] = full[: 6 * nf].reshape(nf, 6)
    if shared_free:
        return dp, full[6 * nf: 6 * nf + 6 * L].reshape(L, 6), mcc, 0.5 * r @ r, full[6 * nf + 6 * L:]
    return dp, full[6 * nf:].reshape(L, 6), mcc, 0.5 * r @ r

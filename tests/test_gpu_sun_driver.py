"""Row N4 end to end: examples/dataset_vo_sun_gpu (the reference's tests/dataset_vo_sun.cpp main written against the
Ceres-shaped shim: sliding windows, GPU RANSAC initial guess, per-point stereo stiffness, sun blocks, pose prior from
the previous window's ceres::Covariance, SUBSPACE_DOGLEG) against a Python restatement of the same pipeline on the
oracle."""
import subprocess

import numpy as np
import pytest

from ceres_slam_amd import frontend, synth
from oracle import oracle as orc
from test_gpu_frontend import _oracle_ransac

pytestmark = pytest.mark.gpu


def _inv_sqrt(A):
    w, V = np.linalg.eigh(0.5 * (A + A.T))
    return (V / np.sqrt(w)) @ V.T


def _pipeline(prob, sun, window, huber, variant, passes):
    """main() of tests/dataset_vo_sun.cpp:258-311 on the oracle."""
    P = prob.num_poses
    poses = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0]), (P, 1))
    poses[0] = prob.poses_gt[0]
    covs = np.zeros((P, 6, 6))
    covs[0] = 1e-12 * np.eye(6)
    S_point = np.array([_inv_sqrt(c) for c in sun["obs_covars"][: prob.num_points]])     # stereo_obs_covars[j], j = point id (:58)
    out = []
    for use_sun in passes:
        for k1 in range(0, P - window + 1):
            k2 = k1 + window
            sel = (prob.obs_pose >= k1) & (prob.obs_pose < k2)
            st, pt, uvd = prob.obs_pose[sel] - k1, prob.obs_point[sel], prob.obs_uvd[sel]
            pw, xw, init, _ = frontend.compute_initial_guess(prob.camera, window, prob.num_points, st, pt, uvd, poses[k1], variant=variant,
                                                             ransac=_oracle_ransac)
            use = init[pt]
            factors = []
            for k in range(k1, k2):
                if use_sun and sun["has_sun"][k]:
                    factors.append(dict(pose=k - k1, type=1, data=np.concatenate([sun["sun_obs"][k], sun["sun_dir_g"][k], [1000.0, 1000.0]]),
                                        stiffness=_inv_sqrt(sun["sun_covars"][k]).ravel(), huber=huber))
            factors.append(dict(pose=0, type=0, data=poses[k1].copy(), stiffness=_inv_sqrt(covs[k1])))
            op = orc.OracleProblem(prob.camera, pw, xw, st[use], pt[use], uvd[use], S_point[pt[use]], pose_const=np.zeros(window, np.uint8),
                                   pose_factors=factors)
            op.solve(orc.driver_options(num_threads=2, trust_region_strategy_type=1, dogleg_type=1))
            poses[k1:k2] = op.poses
            Sred, _, free_idx = op.reduced_system(1e300)
            f = int(free_idx[1])
            covs[k1 + 1] = np.linalg.inv(Sred)[6 * f: 6 * f + 6, 6 * f: 6 * f + 6]
        out.append(poses.copy())
    return out


@pytest.mark.parametrize("window,huber", [(2, 0.0), (4, 0.5)])
def test_sun_driver_matches_the_oracle_pipeline(tmp_path, window, huber):
    from ceres_slam_amd import build
    exe = build.build_examples("dataset_vo_sun_gpu")
    prob = synth.make_problem(10, 600, track_len=6, seed=4, obs_var=(0.04, 0.04, 0.04))
    sun = synth.make_sun_data(prob, seed=1)
    track, ref, obs = synth.write_reference_sun_csv(prob, sun, str(tmp_path / "sim.csv"))
    args = [exe, track, ref, obs, "--window", str(window)] + (["--huber-param", str(huber)] if huber else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "WARNING" not in r.stdout
    reports = [l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")]
    assert len(reports) == 2 * (prob.num_poses - window + 1) and all("CONVERGENCE" in l for l in reports)
    variant = 1 if int(subprocess.run(["g++", "-dumpversion"], capture_output=True, text=True).stdout.split(".")[0]) >= 11 else 0
    no_sun, with_sun = _pipeline(prob, sun, window, huber, variant, passes=(False, True))
    out_vo = synth.read_pose_csv(str(tmp_path / "sim_poses.csv"))
    out_sun = synth.read_pose_csv(str(tmp_path / "sim_sun_poses.csv"))
    assert np.abs(out_vo - no_sun).max() < 1e-5
    assert np.abs(out_sun - with_sun).max() < 1e-5
    assert np.abs(out_sun[:, :3] - prob.poses_gt[:, :3]).max() < 0.1          # and it is a sensible trajectory
    assert np.abs(out_sun - out_vo).max() > 1e-6                              # the sun blocks did something

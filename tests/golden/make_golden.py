"""Regenerates tests/golden/*.json from the CPU oracle.

The reference holds no golden values and cannot be built here (Ceres/Eigen are
absent), so these vectors are ORACLE output on the
deterministic C1 problem -- a regression pin, not reference output.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ceres_slam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    prob = synth.make_config("C1")
    op = orc.OracleProblem.from_synth(prob)
    initial = op.cost()
    cost, g_p, g_l, H_pp, H_ll = op.linearize()
    s, log = op.solve(orc.driver_options(num_threads=1))
    out = {
        "config": "C1 (50 poses / 2000 landmarks), seed 42, stereo_obs_var (4,4,4)",
        "num_obs": prob.num_obs,
        "initial_cost": initial,
        "g_p_pose7": g_p[7].tolist(),
        "H_pp_pose7_diag": np.diag(H_pp[7]).tolist(),
        "g_l_point100": g_l[100].tolist(),
        "termination_type": s.termination_type,
        "num_iterations": s.num_iterations,
        "final_cost": s.final_cost,
        "cost": log["cost"].tolist(),
        "trust_region_radius": log["trust_region_radius"].tolist(),
        "step_is_successful": log["step_is_successful"].tolist(),
        "poses_1_25_49": op.poses[[1, 25, 49]].tolist(),
    }
    with open(os.path.join(HERE, "c1_solve.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote c1_solve.json:", s.num_iterations, "iterations, final cost", s.final_cost)

    # BASELINE.json configs[0]: dataset_ba_phong on 50 poses / 2 000 landmarks with the driver's own settings
    # (free light / Phong / texture blocks, their bounds, DOGLEG + SUBSPACE_DOGLEG, the reference's initial materials)
    prob, ph = synth.make_phong_problem(50, 2000)
    d = ph.as_oracle_dict("reference")
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                           prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    initial = op.cost()
    s, log = op.solve(orc.driver_options(num_threads=1, trust_region_strategy_type=1, dogleg_type=1))
    out = {
        "config": "C1 + lighting terms (50 poses / 2000 landmarks, 4 materials, point light), shared blocks free with bounds, "
                  "DOGLEG / SUBSPACE_DOGLEG, non-monotonic steps, initial shared blocks as the reference's front end sets them",
        "initial_cost": initial, "termination_type": s.termination_type, "num_iterations": s.num_iterations, "final_cost": s.final_cost,
        "cost": log["cost"].tolist(), "step_is_successful": log["step_is_successful"].tolist(),
        "trust_region_radius": log["trust_region_radius"].tolist(),
        "poses_1_25_49": op.poses[[1, 25, 49]].tolist(), "light": op.light.tolist(), "phong": op.phong.tolist(), "texture": op.texture.tolist(),
    }
    # the same solve cut at 12 iterations: an end point that is a parity statement (the converged run stops on a flat,
    # rounding-sensitive tail)
    op12 = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                             prob.stiffness(), lighting=d, shared_free=7, use_bounds=True)
    s12, _ = op12.solve(orc.driver_options(num_threads=1, trust_region_strategy_type=1, dogleg_type=1, max_num_iterations=12))
    out["at_12_iterations"] = {"final_cost": s12.final_cost, "poses_1_25_49": op12.poses[[1, 25, 49]].tolist(), "light": op12.light.tolist(),
                               "texture": op12.texture.tolist()}
    with open(os.path.join(HERE, "c1_phong_driver.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote c1_phong_driver.json:", s.num_iterations, "iterations, final cost", s.final_cost)

    # one window of the sun-aided driver (tests/dataset_vo_sun.cpp:28-185): prior + sun blocks (HuberLoss) + stereo blocks
    sys.path.insert(0, os.path.dirname(HERE))
    from test_oracle_pose_factors import _sun_problem  # noqa: E402
    prob, factors = _sun_problem(P=8, L=400, seed=4, huber=0.5)
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(),
                           pose_const=np.zeros(prob.num_poses, np.uint8), pose_factors=factors)
    initial = op.cost()
    s, log = op.solve(orc.driver_options(num_threads=1, trust_region_strategy_type=1, dogleg_type=1))
    Sred, _, free_idx = op.reduced_system(1e300)
    f = int(free_idx[1])
    out = {
        "config": "_sun_problem(P=8, L=400, seed=4, huber=0.5): stereo blocks + pose prior + sun blocks, SUBSPACE_DOGLEG",
        "initial_cost": initial, "termination_type": s.termination_type, "num_iterations": s.num_iterations, "final_cost": s.final_cost,
        "cost": log["cost"].tolist(), "step_is_successful": log["step_is_successful"].tolist(),
        "poses": op.poses.tolist(), "covariance_pose1": np.linalg.inv(Sred)[6 * f: 6 * f + 6, 6 * f: 6 * f + 6].tolist(),
    }
    with open(os.path.join(HERE, "sun_window.json"), "w") as f2:
        json.dump(out, f2, indent=1)
    print("wrote sun_window.json:", s.num_iterations, "iterations, final cost", s.final_cost)


if __name__ == "__main__":
    main()

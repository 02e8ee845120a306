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


if __name__ == "__main__":
    main()

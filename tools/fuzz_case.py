"""One case of tools/fuzz_parity.py in detail: per-iteration cost difference and accept flags.  usage: fuzz_case.py <cases> <seed> <case>"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc

cases, seed0, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed0)
for c in range(cases):
    P = int(rng.integers(3, 70)); T = int(rng.integers(2, 13)); L = int(rng.integers(max(20, 3 * P), 40 * P + 50))
    seed = int(rng.integers(0, 10**6)); huber = float(rng.choice([0.0, 0.0, 1.345])); dog = int(rng.choice([-1, -1, 0, 1]))
    if c % 4 == 3:      # a lighting case: consume its draws (tools/fuzz_parity.py: lighting_case)
        if c == want:
            raise SystemExit("lighting case: run tools/fuzz_parity.py")
        rng.integers(1, 6); rng.integers(0, 2)
        shared_free = int(rng.choice([0, 7, 7, 5]))
        if shared_free:
            rng.random()
        rng.choice([-1, 1])
        continue
    frac = 0.2 if huber > 0 and rng.random() < 0.5 else 0.0
    pose_const = np.zeros(P, dtype=np.uint8); pose_const[0] = 1
    if rng.random() < 0.4:
        pose_const[rng.integers(0, P, size=max(1, P // 6))] = 1
    if c != want:
        continue
    prob = synth.make_problem(P, L, track_len=min(T, P), seed=seed, outlier_fraction=frac)
    kw = dict(max_num_iterations=60, use_nonmonotonic_steps=1); okw = dict(num_threads=4, max_num_iterations=60)
    if dog >= 0:
        kw.update(trust_region_strategy_type=1, dogleg_type=dog); okw.update(trust_region_strategy_type=1, dogleg_type=dog)
    ba = StereoBA(prob.camera, prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(), pose_const=pose_const, huber_a=huber)
    s, log = ba.solve(capi.default_options(**kw))
    op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(), pose_const=pose_const, huber_a=huber)
    s2, log2 = op.solve(orc.driver_options(**okw))
    print("P", P, "L", L, "T", T, "huber", huber, "dog", dog, "outliers", frac)
    for i in range(min(len(log["cost"]), len(log2["cost"]))):
        print(i, int(log["step_is_successful"][i]), int(log2["step_is_successful"][i]), f"{log['cost'][i]:.10e} {log2['cost'][i]:.10e} rel {abs(log['cost'][i]-log2['cost'][i])/abs(log2['cost'][i]):.1e} radius {log['trust_region_radius'][i]:.3e} {log2['trust_region_radius'][i]:.3e}")

import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc
# case 91 of `tools/fuzz_parity.py 150 23`: re-draw the generator's stream
rng = np.random.default_rng(23)
for c in range(92):
    P = int(rng.integers(3, 70)); T = int(rng.integers(2, 13)); L = int(rng.integers(max(20, 3 * P), 40 * P + 50))
    seed = int(rng.integers(0, 10**6)); huber = float(rng.choice([0.0, 0.0, 1.345])); dog = int(rng.choice([-1, -1, 0, 1]))
    if c % 4 == 3:
        M = int(rng.integers(1, 6)); light_type = int(rng.integers(0, 2)); shared_free = int(rng.choice([0, 7, 7, 5]))
        bounds = bool(shared_free and rng.random() < 0.5); dogl = int(rng.choice([-1, 1]))
        continue
    if huber > 0: rng.random()
    if rng.random() < 0.4: rng.integers(0, P, size=max(1, P // 6))
print(c, P, L, T, M, light_type, shared_free, bounds, dogl, seed)
prob, ph = synth.make_phong_problem(P, L, num_materials=M, light_type=light_type, seed=seed, track_len=min(T, P))
ld = ph.as_oracle_dict("perturbed")
kw = dict(max_num_iterations=40, use_nonmonotonic_steps=1)
ba = StereoBA.from_synth(prob, lighting=ld, shared_free=shared_free, use_bounds=bounds)
s, log = ba.solve(capi.default_options(**kw))
op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd, prob.stiffness(), lighting=ld, shared_free=shared_free, use_bounds=bounds)
s2, log2 = op.solve(orc.driver_options(num_threads=4, max_num_iterations=40))
print(log.keys() if hasattr(log, "keys") else log.dtype.names)
for i in range(10, 20):
    print(i, "hip", {k: float(log[k][i]) for k in ("cost", "cost_change", "step_norm", "relative_decrease", "trust_region_radius") if k in (log.keys() if hasattr(log, "keys") else log.dtype.names)}, int(log["step_is_successful"][i]))
    print(i, "orc", {k: float(log2[k][i]) for k in ("cost", "cost_change", "step_norm", "relative_decrease", "trust_region_radius") if k in (log2.keys() if hasattr(log2, "keys") else log2.dtype.names)}, int(log2["step_is_successful"][i]))

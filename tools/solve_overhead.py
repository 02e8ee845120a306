"""Blocking ssba_solve against batched stepping on the same handle (C2): where the per-iteration difference comes from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
prob = synth.make_config("C2")
ba = StereoBA.from_synth(prob)
opts = capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1)
for rep in range(3):
    ba.poses[:] = prob.poses_init; ba.points[:] = prob.points_init
    t0 = time.perf_counter()
    s, log = ba.solve(opts)
    dt = time.perf_counter() - t0
    print(f"solve {rep}: {int(s.num_iterations)} iterations, wall {1e3*dt:.2f} ms, total_time_s {1e3*s.total_time_s:.2f} ms, device {1e3*s.device_time_s:.2f} ms -> {1e3*s.device_time_s/int(s.num_iterations):.4f} ms / iteration")

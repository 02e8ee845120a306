"""Per-handle cost of small problems (the reference's scripts run its drivers with --window 2 .. 10 over whole sequences:
thousands of handles): create + finalize + solve + destroy, repeated in one process."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
for P, L, T in ((2, 150, 2), (10, 400, 6), (50, 2000, 12)):
    prob = synth.make_problem(P, L, track_len=T, seed=3)
    opts = capi.default_options(max_num_iterations=50, use_nonmonotonic_steps=1)
    times, its = [], []
    for rep in range(30):
        t0 = time.perf_counter()
        ba = StereoBA.from_synth(prob)
        t1 = time.perf_counter()
        s, _ = ba.solve(opts)
        t2 = time.perf_counter()
        ba.close()
        t3 = time.perf_counter()
        times.append((t1 - t0, t2 - t1, t3 - t2)); its.append(int(s.num_iterations))
    a = 1e3 * np.median(np.asarray(times[5:]), axis=0)
    print(f"P={P:3d} L={L:5d}: build {a[0]:.3f} ms, solve {a[1]:.3f} ms ({int(np.median(its))} iterations), destroy {a[2]:.3f} ms")

"""Wall time per window of examples/dataset_vo_sun_gpu on a synthetic sequence (the reference's scripts run
dataset_vo_sun with --window 2 over whole KITTI / Devon sequences: thousands of tiny solves)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_slam_amd import build, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
exe = build.build_examples("dataset_vo_sun_gpu")
prob = synth.make_problem(P, 60 * P, track_len=6, seed=4, obs_var=(0.04, 0.04, 0.04))
sun = synth.make_sun_data(prob, seed=1)
d = tempfile.mkdtemp()
files = synth.write_reference_sun_csv(prob, sun, os.path.join(d, "sim.csv"))
t0 = time.perf_counter()
r = subprocess.run([exe, *files, "--window", str(W)], capture_output=True, text=True, env=dict(os.environ, SSBA_DRIVER_TIMING="1", SSBA_API_TIMING="1"))
dt = time.perf_counter() - t0
n = len([l for l in r.stdout.splitlines() if l.startswith("Ceres Solver Report")])
print("\n".join(l[l.index("stage seconds"):] if "stage seconds" in l else l for l in r.stderr.splitlines() if "stage seconds" in l or l.startswith("[ssba]")))
print(f"rc={r.returncode} windows={n} wall={dt:.2f}s per_window={1e3 * dt / max(n, 1):.2f} ms")

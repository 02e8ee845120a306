import os, subprocess, sys, tempfile, re
sys.path.insert(0, os.getcwd())
from ceres_slam_amd import build, synth
exe = build.build_examples("dataset_vo_sun_gpu")
prob = synth.make_problem(200, 12000, track_len=6, seed=4, obs_var=(0.04, 0.04, 0.04))
sun = synth.make_sun_data(prob, seed=1)
d = tempfile.mkdtemp()
files = synth.write_reference_sun_csv(prob, sun, os.path.join(d, "sim.csv"))
r = subprocess.run([exe, *files, "--window", "2"], capture_output=True, text=True)
its = [int(m.group(1)) for m in re.finditer(r"Iterations: (\d+)", r.stdout)]
import collections
print(len(its), sum(its) / len(its), sorted(collections.Counter(its).items()))

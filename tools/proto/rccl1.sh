#!/bin/bash
# reproduces tests/test_gpu_parity.py::test_cpp_driver_with_the_native_rccl_exchange by hand, with a short timeout
python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
from ceres_slam_amd import build, synth
exe = build.build_examples()
prob = synth.make_problem(12, 150, track_len=6, seed=9)
os.makedirs("gpurun_out/rccl1", exist_ok=True)
ds, ip, im = synth.write_reference_csv(prob, "gpurun_out/rccl1/sim.csv")
open("gpurun_out/rccl1/cmd.txt", "w").write(" ".join([exe, ds, ip, im]))
PY
cmd=$(cat gpurun_out/rccl1/cmd.txt)
for i in 1 2; do
  NCCL_DEBUG=INFO timeout -k 5 45 $cmd --gpus 1 > gpurun_out/rccl1/g1_$i.out 2> gpurun_out/rccl1/g1_$i.err; echo "gpus1 run $i rc=$?"
done
echo "---- run 1 stdout tail"; tail -12 gpurun_out/rccl1/g1_1.out | cut -c1-220

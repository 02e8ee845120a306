#!/bin/bash
# sweep of the pose-linearisation launch shape (SSBA_LP: see launch_linearize)
for v in 0 1 2 3 4 5 6 7 8 9; do
  SSBA_LP=$v python bench.py --no-cpu-baseline --steps 200 > gpurun_out/lp_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/lp_$v.log") if x.startswith("{")][-1]; j=json.loads(l)
print("LP=$v", round(j["ms_per_step"],5), j["kernel_ms_per_iter"]["k_linearize_poses"], j["config"]["converged_final_cost"])
PY
done

#!/bin/bash
for v in "2 2" "1 2" "2 1" "1 1"; do
  set -- $v
  SSBA_LMW=$1 SSBA_LMB=$2 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/lm_sw.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/lm_sw.log") if x.startswith("{")][-1]; j=json.loads(l)
k=j["kernel_ms_per_iter"]
print("LMW=$1 LMB=$2", round(j["ms_per_step"],5), k["k_linearize_landmarks"], k["k_backsub_eval"], j["config"]["converged_final_cost"])
PY
done

#!/bin/bash
# diagnostic: run the two-rank GPU worker a few times and show what each rank prints
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  port=$((29500 + RANDOM % 2000))
  out=/tmp/dist_$i
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 \
      timeout -k 5 60 python tests/dist_worker.py gpu $out > $out.$r.log 2>&1 &
  done
  wait
  for r in 0 1; do
    echo "== run $i rank $r: $(tail -c 400 $out.$r.log | tr '\n' ' ')"
    python - << PY
import json
try:
    d=json.load(open("$out.$r.json")); print("   iters", d["num_iterations"], "term", d["termination"], "final", d["final_cost"])
except Exception as e: print("   no result", e)
PY
  done
done

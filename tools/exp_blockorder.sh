#!/bin/bash
# experiment: cost-ordered block maps
set -e
timeout -k 10 120 python tools/block_cost.py gpurun_out/cost.u32 1920 1080
for mode in base cost costsort; do
  unset RT_BLOCK_COST_FILE RT_BLOCK_SORT
  if [ $mode = cost ]; then export RT_BLOCK_COST_FILE=gpurun_out/cost.u32; fi
  if [ $mode = costsort ]; then export RT_BLOCK_COST_FILE=gpurun_out/cost.u32 RT_BLOCK_SORT=1; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 64 --warmup 8 > gpurun_out/bo_$mode.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bo_$mode.json"))
print("$mode", round(d["ms_per_step"],4), {k:round(v["avg_ms"]*1e3) for k,v in d["kernels"].items()})
PY
done

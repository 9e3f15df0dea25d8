#!/bin/bash
# usage: bash tools/ab_env.sh VAR value1 value2 ...   (bench.py once per value of an environment variable)
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 150 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_${var}_$v.json || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${var}_$v.json"))
print("$var=$v", round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_ms"]*1e3) for k,v in d["kernels"].items()})
PY
done

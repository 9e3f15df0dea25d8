#!/bin/bash
# usage: bash tools/ab_lib.sh lib1.so lib2.so ...   (bench.py once per build of the native library; restores the first)
cp ray-tracer_amd/librtnative.so /tmp/librtnative.keep
for lib in "$@"; do
  cp $lib ray-tracer_amd/librtnative.so
  timeout -k 10 150 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_lib.json || { cp /tmp/librtnative.keep ray-tracer_amd/librtnative.so; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_lib.json"))
print("$lib", round(d["ms_per_step"],4), round(d["value"]), {k:(round(v["avg_ms"]*1e3), round(v.get("isolated_avg_ms", 0)*1e3, 1)) for k,v in d["kernels"].items()})
PY
done
cp /tmp/librtnative.keep ray-tracer_amd/librtnative.so

# usage: bash tools/run_variants.sh  -- times every var_build/lib_*.so with tools/stage_timing.py (kernel experiments)
for f in var_build/lib_*.so; do
  echo "== $f"
  RT_NATIVE_LIB=$PWD/$f timeout -k 10 120 python tools/stage_timing.py 2>&1 | grep "stats level 0" || { echo FAILED; exit 1; }
done
echo "== default"; timeout -k 10 120 python tools/stage_timing.py 2>&1 | grep "stats level 0"

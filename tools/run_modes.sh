for m in 0 1 2; do
  echo "== RT_BLOCK_MAP=$m"
  RT_BLOCK_MAP=$m timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/bench_mode$m.json 2> gpurun_out/bench_mode$m.err || { echo FAILED; tail -5 gpurun_out/bench_mode$m.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/bench_mode$m.json'))
print(d['value'], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done

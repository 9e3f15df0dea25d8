# usage: bash tools/sweep_env.sh VAR v1 v2 ...   (bench.py per value of an environment variable; prints kernel times)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 32 > gpurun_out/sweep_$v.json 2>/dev/null || { echo "$var=$v failed"; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sweep_$v.json').read().strip().splitlines()[-1])
print('$var=$v', round(d['value']), round(d['ms_per_step'],4), ' '.join(k+'='+str(round(x['avg_ms']*1e3,1)) for k,x in d['kernels'].items()))"
done

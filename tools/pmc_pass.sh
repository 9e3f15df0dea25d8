# usage: bash tools/pmc_pass.sh <workload tag> <pass> "<counters>" [bench.py arguments]
# One rocprofv3 --pmc pass (a run of its own, never combined with a trace) over a short bench run of that workload;
# leaves gpurun_out/pmc_<tag>/<pass>_summary.txt (per kernel, the counters averaged over its dispatches).
export TMPDIR=/tmp
tag=$1; pass=$2; counters=$3; shift 3
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc $counters --output-format csv -d $out/$pass -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-4k "$@" > $out/$pass.json 2> $out/$pass.err || { echo "pass $pass of $tag failed"; tail -5 $out/$pass.err; exit 1; }
python3 tools/pmc_summary.py $out/$pass > $out/${pass}_summary.txt

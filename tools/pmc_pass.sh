# usage: bash tools/pmc_pass.sh <tag> "<counters>"   (one rocprofv3 --pmc pass over a short bench run)
export TMPDIR=/tmp
tag=$1; shift
timeout -s KILL 240 rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err || { echo "pass $tag failed"; tail -5 gpurun_out/pmc_$tag.err; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_$tag

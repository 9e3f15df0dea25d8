#!/bin/bash
# Collects everything under profiles/ for the current build on the GPU box: kernel trace, PMC passes, bench lines.
# Each rocprofv3 pass is a run of its own (counters are never combined with traces).  Afterwards, here:
#   bash tools/publish_profiles.sh
# usage: bash tools/collect_profiles.sh [default|unique4k|4k|instanced4k|256|ris|benches ...]   (no argument: all of them)
export TMPDIR=/tmp
set -e
what=${@:-default unique4k 4k instanced4k 256 ris benches}
passes() {   # <tag> [bench.py arguments]
    tag=$1; shift
    bash tools/pmc_pass.sh $tag fetch "FETCH_SIZE" "$@"
    bash tools/pmc_pass.sh $tag write "WRITE_SIZE" "$@"
    bash tools/pmc_pass.sh $tag sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "$@"
    bash tools/pmc_pass.sh $tag ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "$@"
    bash tools/pmc_pass.sh $tag tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "$@"
}
for w in $what; do
case $w in
default)
    timeout -k 10 200 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
    timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace -- python3 bench.py --no-cpu-baseline --no-4k > gpurun_out/bench_under_rocprof.json 2> gpurun_out/ktrace.err
    passes restir_80k_1920x1080 ;;
unique4k)
    passes restir_unique_3840x2160 --scene unique --width 3840 --height 2160
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace_unique4k -- python3 bench.py --no-cpu-baseline --scene unique --width 3840 --height 2160 --steps 16 --warmup 4 > gpurun_out/bench_unique4k_under_rocprof.json 2> gpurun_out/ktrace_unique4k.err ;;
256)
    passes restir_80k_256x256_tri_reservoir_tests32 --width 256 --height 256 --tri-reservoir-tests 32 ;;
4k)
    passes restir_80k_3840x2160 --width 3840 --height 2160 ;;
instanced4k)
    passes restir_instanced_3840x2160 --scene instanced --width 3840 --height 2160 ;;
ris)
    passes ris_80k_1920x1080 --project ris
    timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace_ris -- python3 bench.py --no-cpu-baseline --project ris > gpurun_out/bench_ris_under_rocprof.json 2> gpurun_out/ktrace_ris.err ;;
benches)
    timeout -k 10 200 python3 bench.py --width 3840 --height 2160 --no-cpu-baseline > gpurun_out/bench_4k.json 2> gpurun_out/bench_4k.err
    timeout -k 10 200 python3 bench.py --project ris > gpurun_out/bench_ris.json 2> gpurun_out/bench_ris.err
    timeout -k 10 300 python3 bench.py --scene instanced --width 3840 --height 2160 --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/bench_instanced_4k.json 2> gpurun_out/bench_instanced_4k.err
    timeout -k 10 400 python3 bench.py --scene unique --width 3840 --height 2160 --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/bench_unique_4k.json 2> gpurun_out/bench_unique_4k.err
    timeout -k 10 200 python3 bench.py --width 256 --height 256 --tri-reservoir-tests 32 > gpurun_out/bench_256.json 2> gpurun_out/bench_256.err ;;
esac
done
echo done

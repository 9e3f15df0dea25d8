#!/bin/bash
# Collects everything under profiles/ for the current build on the GPU box: kernel trace, PMC passes, bench lines.
# Each rocprofv3 pass is a run of its own (counters are never combined with traces).  Afterwards, here:
#   python tools/make_pmc_profile.py gpurun_out > profiles/r02_pmc.json   and copy the summaries (see README of profiles/).
export TMPDIR=/tmp
set -e
timeout -k 10 200 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/ktrace.err
bash tools/pmc_pass.sh fetch "FETCH_SIZE" > gpurun_out/pmc_fetch_summary.txt
bash tools/pmc_pass.sh write "WRITE_SIZE" > gpurun_out/pmc_write_summary.txt
bash tools/pmc_pass.sh sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" > gpurun_out/pmc_sq_summary.txt
bash tools/pmc_pass.sh ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" > gpurun_out/pmc_ta_summary.txt
bash tools/pmc_pass.sh tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" > gpurun_out/pmc_tcc_summary.txt
timeout -k 10 200 python3 bench.py --width 3840 --height 2160 --no-cpu-baseline > gpurun_out/bench_4k.json 2> gpurun_out/bench_4k.err
timeout -k 10 200 python3 bench.py --project ris > gpurun_out/bench_ris.json 2> gpurun_out/bench_ris.err
timeout -k 10 300 python3 bench.py --scene instanced --width 3840 --height 2160 --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/bench_instanced_4k.json 2> gpurun_out/bench_instanced_4k.err
timeout -k 10 400 python3 bench.py --scene unique --width 3840 --height 2160 --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/bench_unique_4k.json 2> gpurun_out/bench_unique_4k.err
echo done

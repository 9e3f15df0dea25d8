#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/ into profiles/ (tracked), named for this round.
set -e
R=r02
cp "$(ls -t gpurun_out/ktrace/*/*_kernel_stats.csv | head -1)" profiles/${R}_kernel_stats.csv
cat gpurun_out/pmc_fetch_summary.txt gpurun_out/pmc_write_summary.txt gpurun_out/pmc_sq_summary.txt gpurun_out/pmc_ta_summary.txt gpurun_out/pmc_tcc_summary.txt > profiles/${R}_pmc_summary.txt
python3 tools/make_pmc_profile.py gpurun_out > profiles/${R}_pmc.json
cp gpurun_out/bench_final.json profiles/${R}_bench.json
cp gpurun_out/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
for n in 4k ris instanced_4k unique_4k; do cp gpurun_out/bench_$n.json profiles/${R}_bench_$n.json; done
[ -f gpurun_out/bench_256.json ] && cp gpurun_out/bench_256.json profiles/${R}_bench_256.json
ls -la profiles/

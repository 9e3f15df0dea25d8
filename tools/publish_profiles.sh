#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/ into profiles/ (tracked), under the prefix R.
set -e
R=r03
cp "$(ls -t gpurun_out/ktrace/*/*_kernel_stats.csv | head -1)" profiles/${R}_kernel_stats.csv
[ -d gpurun_out/ktrace_unique4k ] && cp "$(ls -t gpurun_out/ktrace_unique4k/*/*_kernel_stats.csv | head -1)" profiles/${R}_kernel_stats_unique_4k.csv
[ -d gpurun_out/ktrace_ris ] && cp "$(ls -t gpurun_out/ktrace_ris/*/*_kernel_stats.csv | head -1)" profiles/${R}_kernel_stats_ris.csv
pub() {   # <tag> [make_pmc_profile arguments]
    tag=$1; shift
    [ -d gpurun_out/pmc_$tag ] || return 0
    cat gpurun_out/pmc_$tag/*_summary.txt > profiles/${R}_pmc_summary_$tag.txt
    python3 tools/make_pmc_profile.py gpurun_out/pmc_$tag "$@" > profiles/${R}_pmc_$tag.json
}
pub restir_80k_1920x1080
pub restir_unique_3840x2160 --scene unique --width 3840 --height 2160
pub restir_80k_3840x2160 --width 3840 --height 2160
pub restir_instanced_3840x2160 --scene instanced --width 3840 --height 2160
pub restir_80k_256x256_tri_reservoir_tests32 --width 256 --height 256 --tri-reservoir-tests 32
pub ris_80k_1920x1080 --project ris
cp gpurun_out/bench_final.json profiles/${R}_bench.json
cp gpurun_out/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
for n in 4k ris instanced_4k unique_4k 256; do [ -f gpurun_out/bench_$n.json ] && cp gpurun_out/bench_$n.json profiles/${R}_bench_$n.json; done
ls -la profiles/

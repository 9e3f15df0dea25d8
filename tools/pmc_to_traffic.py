#!/usr/bin/env python3
"""Turns the FETCH_SIZE and WRITE_SIZE pass summaries (tools/pmc_pass.sh) into per-kernel HBM traffic per launch.

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so fetched
bytes = FETCH_SIZE x 1024 x 2; WRITE_SIZE x 1024 is exact for 16-byte-per-lane stores.  The factor 2 is calibrated
for wide coalesced reads only; scattered 16-byte node fetches may be counted differently, so the raw value is kept.

    python tools/pmc_to_traffic.py gpurun_out/pmc_fetch_summary.txt gpurun_out/pmc_write_summary.txt > profiles/r01_hbm_traffic.json
"""
import json
import re
import sys

names = {"k_primary<false>": "primary", "k_candidates<false>": "candidates", "k_trace_rays<false>": "trace_rays",
         "k_temporal": "temporal", "k_spatial": "spatial", "k_transfer": "transfer", "k_accumulate": "accumulate"}
out = {}
for path, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    for line in open(path):
        m = re.search(r"rtn::(\S+) dispatches=(\d+) " + counter + r"=(\S+)", line)
        if m and m.group(1) in names:
            out.setdefault(names[m.group(1)], {})[counter + "_KiB_per_launch"] = float(m.group(3))
for k, v in out.items():
    v["hbm_bytes_per_launch"] = v.get("FETCH_SIZE_KiB_per_launch", 0) * 1024 * 2 + v.get("WRITE_SIZE_KiB_per_launch", 0) * 1024
json.dump({"workload": "bench.py default (80kScene* 1920x1080), rocprofv3 --pmc, one pass per counter",
           "correction": "bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024", "kernels": out}, sys.stdout, indent=1)
print()

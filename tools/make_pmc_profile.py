#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc pass summaries of tools/collect_profiles.sh into profiles/r02_pmc.json: per kernel of the
benchmark frame, the HBM bytes per launch and the occupancy figures bench.py quotes next to its timings.

    python tools/make_pmc_profile.py gpurun_out/pmc_<tag> [bench.py arguments of the passes] > profiles/r03_pmc_<tag>.json

FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so
bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (checked on k_accumulate, whose 48 bytes per pixel are known).
SQ counters count quad-cycles summed over all waves / SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
The file is stamped with a hash of the library's sources: bench.py quotes it only for the build it describes.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash)

NAMES = {"k_primary": "primary", "k_candidates": "candidates", "k_trace_rays": "trace_rays", "k_temporal": "temporal",
         "k_spatial": "spatial", "k_transfer": "transfer", "k_accumulate": "accumulate",
         "k_ris_primary": "ris_primary", "k_ris_candidates": "ris_candidates", "k_ris_trace_rays": "ris_trace_rays"}
CUS, SIMDS, XCDS = 256, 1024, 8


def read(path):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?rtn::(k_\w+)(?:<([\w, ]+)>)? dispatches=(\d+) (.*)", line.strip())
        if not m or m.group(1) not in NAMES or (m.group(2) and m.group(2).split(",")[0].strip() != "false"):
            continue   # the instrumented (<true>) variants only run in the untimed counter frames
        vals = {k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split())}
        vals["_dispatches"] = int(m.group(3))
        name = NAMES[m.group(1)]
        if name not in out or vals["_dispatches"] > out[name]["_dispatches"]:   # merged and unmerged variants: the one the frames ran
            out[name] = vals
    return out


import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--scene", default="80k")
ap.add_argument("--project", default="restir")
ap.add_argument("--small-scene", action="store_true")
ap.add_argument("--tri-reservoir-tests", type=int, default=None)
args, _ = ap.parse_known_args()
d = args.dir
fetch, write = read(f"{d}/fetch_summary.txt"), read(f"{d}/write_summary.txt")
sq, ta = read(f"{d}/sq_summary.txt"), read(f"{d}/ta_summary.txt")
tcc = read(f"{d}/tcc_summary.txt") if os.path.exists(f"{d}/tcc_summary.txt") else {}
kernels = {}
for name in fetch:
    k = {"FETCH_SIZE_KiB": fetch[name]["FETCH_SIZE"], "WRITE_SIZE_KiB": write.get(name, {}).get("WRITE_SIZE", 0.0)}
    k["hbm_fetch_MB_per_launch"] = k["FETCH_SIZE_KiB"] * 1024 * 2 / 1e6
    k["hbm_write_MB_per_launch"] = k["WRITE_SIZE_KiB"] * 1024 / 1e6
    k["hbm_bytes_per_launch"] = k["FETCH_SIZE_KiB"] * 1024 * 2 + k["WRITE_SIZE_KiB"] * 1024
    if name in sq and name in ta:
        s, t = sq[name], ta[name]
        cycles = t["GRBM_GUI_ACTIVE"] / XCDS                      # shader clocks the launch was resident
        k["valu_busy_frac"] = s["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * SIMDS)
        k["wave_wait_frac"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
        k["valu_insts_per_launch"] = s["SQ_INSTS_VALU"]
        k["ta_busy_frac"] = t["TA_TA_BUSY_sum"] / (cycles * CUS)
        k["td_busy_frac"] = t["TD_TD_BUSY_sum"] / (cycles * CUS)
        k["l1_accesses_per_launch"] = t["TCP_TOTAL_CACHE_ACCESSES_sum"]
        k["l1_accesses_per_cu_clk"] = t["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cycles * CUS)
        k["l1_hit_frac"] = 1.0 - t["TCP_TCC_READ_REQ_sum"] / t["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if name in tcc and tcc[name].get("TCC_REQ_sum"):
        k["l2_hit_frac"] = tcc[name]["TCC_HIT_sum"] / (tcc[name]["TCC_HIT_sum"] + tcc[name]["TCC_MISS_sum"])
        k["l2_requests_per_launch"] = tcc[name]["TCC_REQ_sum"]
    kernels[name] = k
json.dump({"workload": "bench.py with the arguments of `config` under rocprofv3 --pmc, one pass per counter group (tools/collect_profiles.sh)",
           "config": dict({"width": args.width, "height": args.height, "world": 1, "small_scene": args.small_scene, "scene": args.scene, "project": args.project},
                          **({"tri_reservoir_tests": args.tri_reservoir_tests} if args.tri_reservoir_tests is not None else {})),
           "source_hash": bench.source_hash(),
           "correction": "hbm bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes)",
           "kernels": kernels}, sys.stdout, indent=1)
print()

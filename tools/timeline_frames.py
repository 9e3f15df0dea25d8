#!/usr/bin/env python3
"""Prints the kernels of a few consecutive pipelined frames from a rocprofv3 --kernel-trace CSV of a bench.py run:
start, end, duration (us, relative to the first), stream and queue.  usage: timeline_frames.py <csv> [first-mark] [count]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void rtn::", "").replace("rtn::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Stream_Id"]), int(r["Queue_Id"])))
rows.sort()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
marks = [i for i, r in enumerate(rows) if "k_trace_rays<false, true, true>" in r[2] or "k_ris_trace_rays<false, true>" in r[2]]
i0 = marks[first]
t0 = rows[i0][0]
for r in rows[i0 - 2:i0 + count]:
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  s{r[3]} q{r[4]} {r[2]}")

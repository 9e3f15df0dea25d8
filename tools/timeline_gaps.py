#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for the longest run of pipelined frames, the
interval between consecutive launches of the frame's last kernel, how much of the time no kernel at all was running,
and the frames whose interval stands out.  usage: timeline_gaps.py <kernel_trace.csv> [kernel-name-substring]"""
import csv
import sys

path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_trace_rays<false, true, true>"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Stream_Id"])))
rows.sort()
marks = [r for r in rows if key in r[2]]
# the timed region: the longest run of marks whose spacing stays under twice the median
gaps = [b[0] - a[0] for a, b in zip(marks, marks[1:])]
med = sorted(gaps)[len(gaps) // 2]
best, cur = (0, 0), 0
for i, g in enumerate(gaps):
    if g < 2 * med:
        cur += 1
        if cur > best[1] - best[0]:
            best = (i + 1 - cur, i + 1)
    else:
        cur = 0
lo, hi = best
sel = gaps[lo:hi]
print(f"{len(marks)} launches of '{key}', longest steady run {len(sel)} intervals, median {med/1e3:.1f} us, mean {sum(sel)/len(sel)/1e3:.1f} us")
print("intervals (us):", " ".join(f"{g/1e3:.0f}" for g in sel))
t0, t1 = marks[lo][0], marks[hi][0]
inside = [r for r in rows if r[1] > t0 and r[0] < t1]
# time with no kernel running
events = sorted([(max(r[0], t0), 1) for r in inside] + [(min(r[1], t1), -1) for r in inside])
idle, depth, last = 0, 0, t0
for t, d in events:
    if depth == 0:
        idle += t - last
    depth += d
    last = t
print(f"window {(t1-t0)/1e3:.0f} us, no kernel running for {idle/1e3:.0f} us ({100*idle/(t1-t0):.1f} %)")
names = {}
for r in inside:
    n = r[2].split("(")[0][-40:]
    a = names.setdefault(n, [0, 0])
    a[0] += 1; a[1] += r[1] - r[0]
for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:42s} {c:5d} launches, {t/c/1e3:8.1f} us each, {t/(t1-t0):5.2f} of the window")

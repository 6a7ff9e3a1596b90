#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of tools/refmode_latency.py and prints, for the LAST align_reference call, every
pass (a pass starts at its scan launch; a fill or staging launch in front of it is shown with the pass): its kernels with
start offset, gap to the previous kernel's end and duration, then totals per kernel.
    python tools/trace_passes.py <kernel_trace.csv> [passes per call = 17]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 17


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("void mi355::", "").replace("mi355::", "")[:44]


scans = [i for i, r in enumerate(rows) if "nn_scan" in r["Kernel_Name"] or "nn_pruned" in r["Kernel_Name"]]
first = scans[-n_pass]
while first > 0 and ("fillBuffer" in rows[first - 1]["Kernel_Name"] or "stage_jobs" in rows[first - 1]["Kernel_Name"]):
    first -= 1
body = rows[first:]
t0 = int(body[0]["Start_Timestamp"])
prev = t0
tot = defaultdict(lambda: [0, 0.0])
gaps = 0.0
for k, r in enumerate(body):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "nn_scan" in r["Kernel_Name"] or "nn_pruned" in r["Kernel_Name"]:
        print("--- pass")
    print(f"   {(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm(r)}")
    tot[nm(r)][0] += 1
    tot[nm(r)][1] += (e - s) / 1e3
    gaps += max(s - prev, 0) / 1e3
    prev = e
span = (prev - t0) / 1e3
print(f"span {span:.1f} us, kernel time {sum(v[1] for v in tot.values()):.1f} us, gaps {gaps:.1f} us (profiled: gaps are larger than in a free run)")
for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"   {us:8.1f} us  {n:3d} x  {name}")

#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of tools/refmode_latency.py and prints, for the LAST align_reference call, every
pass (a pass starts at stage_jobs_kernel): its kernels with start offset, gap to the previous kernel's end and duration."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("void mi355::", "").replace("mi355::", "")[:40]


starts = [i for i, r in enumerate(rows) if "stage_jobs_kernel" in r["Kernel_Name"]]
n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 17
first = starts[-n_pass]
body = rows[first:]
t0 = int(body[0]["Start_Timestamp"])
prev = t0
pass_start = t0
tot_k = 0
for r in body:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "stage_jobs_kernel" in r["Kernel_Name"]:
        print(f"--- pass at {(s - t0) / 1e3:8.1f} us (previous pass took {(s - pass_start) / 1e3:6.1f} us)")
        pass_start = s
    print(f"   {(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm(r)}")
    tot_k += e - s
    prev = e
print(f"span {(prev - t0) / 1e3:.1f} us, kernel time {tot_k / 1e3:.1f} us")

#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of `estimate_pose_harness ... pipeline` and prints, for the LAST request (the
second, warmed-up one), every GPU command in order with its gap to the previous one, and totals per kernel name."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "filter"


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("void mi355::", "").replace("mi355::", "")[:48] or "(blit)"


# the last request: every request fits one plane (plane_hypotheses_kernel) and ends with an accumulate launch of the ICP
hyp = [i for i, r in enumerate(rows) if "plane_hypotheses_kernel" in r["Kernel_Name"]]
last_hyp = hyp[-1]
prev_end = max(i for i in range(last_hyp) if "ref_accumulate_kernel" in rows[i]["Kernel_Name"]) if len(hyp) > 1 else -1
first = prev_end + 1
body = rows[first:]
t0 = int(body[0]["Start_Timestamp"])
prev = t0
tot = OrderedDict()
for r in body:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "-v" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f}  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {nm(r)}")
    k = nm(r)
    c = tot.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
    prev = e
print(f"request: {len(body)} commands, span {(prev - t0) / 1e3:.1f} us, busy {sum(v[1] for v in tot.values()):.1f} us")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us:9.1f} us  x{n:<4d} {k}")

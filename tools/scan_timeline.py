#!/usr/bin/env python3
"""Workgroup timeline of one scan launch (diagnostic build: make -C pose_estimation_amd/csrc diag).
    MI355_ICP_LIB=build/diag/libmi355_icp.so python tools/scan_timeline.py --n 100000 [--ppl 2]
Prints how many workgroups are resident over time, per-CU occupancy and the spread of workgroup durations."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MI355_ICP_LIB", os.path.join(ROOT, "build", "diag", "libmi355_icp.so"))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--nt", type=int, default=0)
ap.add_argument("--jobs", type=int, default=1)
ap.add_argument("--ppl", type=int, default=0)
ap.add_argument("--target-blocks", type=int, default=0)
ap.add_argument("--min-tiles", type=int, default=0)
ap.add_argument("--iters", type=int, default=1, help="iterations of the stamped align call (stamps keep the last launch)")
a = ap.parse_args()
if a.ppl:
    os.environ["MI355_ICP_SCAN_PPL"] = str(a.ppl)
if a.target_blocks:
    os.environ["MI355_ICP_TARGET_BLOCKS"] = str(a.target_blocks)
if a.min_tiles:
    os.environ["MI355_ICP_MIN_TILES_PER_SPLIT"] = str(a.min_tiles)
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

lib = pa._lib.lib
nt = a.nt or a.n
pairs = [st.make_pair(pa.synth_cloud, a.n, nt, object_index=(k if a.jobs > 1 else -1)) for k in range(a.jobs)]
eng = pa.IcpEngine(0)
eng.set_targets([p[1] for p in pairs])
eng.set_sources([p[0] for p in pairs])
ids = list(range(a.jobs))
init = np.stack([np.eye(4)] * a.jobs)
p = pa.default_params(max_iterations=1, flags=pa._lib.FLAG_PROFILE)
for _ in range(3):
    eng.align(init, p, ids, ids)  # warm-up (clocks)
NS = 1 << 20
lib.mi355_icp_diag_begin.argtypes = [C.c_size_t]
lib.mi355_icp_diag_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.mi355_icp_diag_begin(NS) == 0
eng.reset_profile()
eng.align(init, pa.default_params(max_iterations=a.iters, flags=pa._lib.FLAG_PROFILE), ids, ids)
print(f"stamped call: {a.iters} iteration(s), mean scan event time {eng.profile().nn_scan_ms / a.iters:.3f} ms")
buf = np.zeros((NS, 4), dtype=np.uint64)
assert lib.mi355_icp_diag_read(buf.ctypes.data, NS) == 0
used = buf[:, 1] != 0
b = buf[used]
t0 = b[:, 0].astype(np.int64)
t1 = b[:, 1].astype(np.int64)
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds
hw = b[:, 2]
xcc = (hw >> np.uint64(32)).astype(np.int64)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
dur = e - s
cyc = b[:, 3].astype(np.float64)
ghz = cyc / (dur * 1e3)
print(f"shader clock seen by workgroups (cycles / realtime): median {np.median(ghz):.3f} GHz, p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f}")
print(f"{a.jobs} x {a.n} x {nt}: {len(b)} workgroups, scan event time {eng.profile().nn_scan_ms:.3f} ms, stamped span {e.max():.1f} us")
print(f"workgroup duration us: min {dur.min():.1f} p10 {np.percentile(dur,10):.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f}")
print(f"distinct CUs seen: {len(np.unique(cu_key))}; workgroups per CU min/mean/max: "
      f"{np.bincount(np.unique(cu_key, return_inverse=True)[1]).min()}/{len(b)/len(np.unique(cu_key)):.1f}/{np.bincount(np.unique(cu_key, return_inverse=True)[1]).max()}")
T = e.max()
edges = np.linspace(0, T, 21)
print("time slice (us)      resident workgroups (avg)   busy CUs")
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(e, hi) - np.maximum(s, lo), 0, None)
    resident = ov.sum() / (hi - lo)
    busy = len(np.unique(cu_key[ov > 0]))
    print(f"{lo:8.1f}-{hi:8.1f}   {resident:10.1f}                 {busy}")
# per-CU busy time (union of intervals) vs span
tot = 0.0
for k in np.unique(cu_key):
    m = cu_key == k
    iv = sorted(zip(s[m], e[m]))
    cur_s, cur_e, busy = iv[0][0], iv[0][1], 0.0
    for x, y in iv[1:]:
        if x > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = x, y
        else:
            cur_e = max(cur_e, y)
    busy += cur_e - cur_s
    tot += busy
print(f"mean CU busy fraction of the span: {tot / len(np.unique(cu_key)) / T:.3f}")

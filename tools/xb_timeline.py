#!/usr/bin/env python3
"""Workgroup timeline of the LAST matrix-core scan launch of one reference-shaped request (diagnostic build:
make -C pose_estimation_amd/csrc diag): when workgroups start and end, how long prologue, tile walk and epilogue take, how
evenly the CUs are loaded.
    MI355_ICP_LIB=build/diag/libmi355_icp.so python tools/xb_timeline.py [--model 20000] [--scene 100000] [--candidates 6]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MI355_ICP_LIB", os.path.join(ROOT, "build", "diag", "libmi355_icp.so"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", type=int, default=20_000)
ap.add_argument("--scene", type=int, default=100_000)
ap.add_argument("--candidates", type=int, default=6)
ap.add_argument("--levels", type=int, default=8)
a = ap.parse_args()
lib = pa._lib.lib
T = st.perturbation(-1)
tgt = pa.synth_cloud(st.TARGET_SEED, a.scene, normals=True)
src = pa.synth_cloud(st.SOURCE_SEED, a.model, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
gt = np.linalg.inv(T)
cands = [np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0)]
inits = np.stack(cands[:a.candidates])
NS = 1 << 16
lib.mi355_icp_diag_xb_begin.argtypes = [C.c_size_t]
lib.mi355_icp_diag_xb_read.argtypes = [C.c_void_p, C.c_size_t]
with pa.IcpEngine(0) as eng:
    eng.set_targets([tgt])
    eng.set_sources([src])
    for _ in range(3):
        eng.align_reference(inits, 250, 0.005, 2.5, a.levels)
    assert lib.mi355_icp_diag_xb_begin(NS) == 0
    eng.align_reference(inits, 250, 0.005, 2.5, a.levels)
    buf = np.zeros((NS, 8), dtype=np.uint64)
    assert lib.mi355_icp_diag_xb_read(buf.ctypes.data, NS) == 0
used = buf[:, 3] != 0
# slots keep the last launch that wrote them; a coarser level's launch has fewer workgroups, so the finest level's launch is
# the set of slots with the latest stamps: keep the workgroups that started within 1 ms of the latest start
t0_all = buf[:, 0].astype(np.int64)
keep = used & (t0_all > t0_all[used].max() - 100_000)
linear = np.nonzero(keep)[0]  # slot = blockIdx.y * gridDim.x + blockIdx.x: the order the dispatcher is expected to follow
b = buf[keep]
t0, t1, t2, t3 = (b[:, k].astype(np.int64) for k in range(4))
base = t0.min()
us = lambda t: (t - base) / 100.0
tiles = (b[:, 4] >> np.uint64(48)).astype(np.int64)
job = ((b[:, 4] >> np.uint64(40)) & np.uint64(0xFF)).astype(np.int64)
xcc = ((b[:, 4] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
hwid = (b[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu_key = ((xcc * 8 + ((hwid >> 13) & 7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 0xF)
pro, walk, epi, dur = us(t1) - us(t0), us(t2) - us(t1), us(t3) - us(t2), us(t3) - us(t0)
pct = lambda v: f"min {v.min():.1f} p10 {np.percentile(v, 10):.1f} median {np.median(v):.1f} p90 {np.percentile(v, 90):.1f} max {v.max():.1f}"
print(f"{len(b)} workgroups of the last matrix-core launch, {len(np.unique(job))} jobs, tiles per workgroup {tiles.min()}-{tiles.max()}; span {us(t3).max():.1f} us")
print(f"start us       : {pct(us(t0))}")
print(f"end us         : {pct(us(t3))}")
print(f"prologue us    : {pct(pro)}")
print(f"tile walk us   : {pct(walk)}   per tile: {pct(walk / np.maximum(tiles, 1))}")
print(f"epilogue us    : {pct(epi)}")
print(f"duration us    : {pct(dur)}")
per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
print(f"distinct CUs {len(per_cu)}; workgroups per CU min/mean/max {per_cu.min()}/{per_cu.mean():.2f}/{per_cu.max()}; CUs with 1 / 2 / 3+ workgroups: "
      f"{(per_cu == 1).sum()} / {(per_cu == 2).sum()} / {(per_cu >= 3).sum()}")
for k in sorted(np.unique(per_cu)):
    sel = np.isin(cu_key, np.unique(cu_key)[per_cu == k])
    print(f"  workgroups on CUs holding {k}: per-tile walk median {np.median((walk / np.maximum(tiles, 1))[sel]):.2f} us, end median {np.median(us(t3)[sel]):.1f} max {us(t3)[sel].max():.1f}")
first = linear < 256
if first.any() and (~first).any():
    print(f"workgroups 0-255 in grid order: {len(np.unique(cu_key[first]))} distinct CUs, start median {np.median(us(t0)[first]):.1f} us, per-tile walk median "
          f"{np.median((walk / np.maximum(tiles, 1))[first]):.2f} us, end median {np.median(us(t3)[first]):.1f}; the others: start median {np.median(us(t0)[~first]):.1f}, "
          f"per tile {np.median((walk / np.maximum(tiles, 1))[~first]):.2f}, end median {np.median(us(t3)[~first]):.1f}")
q1, q2, q3 = (us(b[:, k].astype(np.int64)) for k in (5, 6, 7))
quarters = np.stack([q1 - us(t1), q2 - q1, q3 - q2, us(t2) - q3], axis=1) / np.maximum(tiles[:, None] / 4.0, 1)
for name, sel in (("first 256 in grid order", first), ("the others", ~first)):
    if sel.any():
        print(f"  {name}: per-tile walk by quarter of the walk, median us: " + " ".join(f"{np.median(quarters[sel, k]):.2f}" for k in range(4)))
for jb in sorted(np.unique(job)):
    for name, sel in (("first 256", first & (job == jb)), ("later", ~first & (job == jb))):
        if sel.any():
            print(f"  job {jb} {name:9s}: {int(sel.sum()):3d} workgroups, per-tile walk median {np.median((walk / np.maximum(tiles, 1))[sel]):.2f} us")
T_end = us(t3).max()
grid = np.linspace(0, T_end, 21)
res = [(int(((us(t0) <= g) & (us(t3) > g)).sum())) for g in grid]
print("resident workgroups at 5 % steps of the span:", res)

#!/usr/bin/env python3
"""Where a compaction tile's life goes (diagnostic build: make -C pose_estimation_amd/csrc diag).
    MI355_ICP_LIB=build/diag/libmi355_icp.so python tools/scene_timeline.py [--n 64000000] [--keep 0.8]
Eight 100 MHz stamps per tile (thread 0): entry, ticket, tile in LDS, flags ranked, prefix ready, look-back done, stores issued,
stores complete."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355_ICP_LIB", os.path.join(ROOT, "build", "diag", "libmi355_icp.so"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64_000_000)
ap.add_argument("--keep", type=float, default=0.8)
a = ap.parse_args()
lib = pa._lib.lib
rng = np.random.default_rng(0)
pts = rng.standard_normal((a.n, 3)).astype(np.float32)
pts[rng.random(a.n) >= a.keep, 0] = np.nan
TILE = int(os.environ.get("MI355_SCENE_TILE", 12288))  # kSceneTile of the library under test (icp_internal.h)
n_tiles = (a.n + TILE - 1) // TILE
lib.mi355_icp_diag_scene_begin.argtypes = [C.c_size_t]
lib.mi355_icp_diag_scene_read.argtypes = [C.c_void_p, C.c_size_t]
with pa.IcpEngine(0) as eng:
    eng.set_profiling(True)
    for _ in range(2):
        eng.scene_load(pts)
        eng.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
    assert lib.mi355_icp_diag_scene_begin(n_tiles) == 0
    eng.scene_load(pts)
    eng.reset_profile()
    kept = eng.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
    ms = eng.profile().scene_filter_ms
    buf = np.zeros((n_tiles, 8), dtype=np.uint64)
    assert lib.mi355_icp_diag_scene_read(buf.ctypes.data, n_tiles) == 0
t = buf.astype(np.int64)
base = t[:, 0].min()
us = (t - base) / 100.0
names = ["ticket", "loads -> LDS", "rank flags", "prefix", "look-back", "issue stores", "stores complete"]
print(f"{a.n} points, {n_tiles} tiles, kept {kept}, kernel {ms * 1e3:.1f} us (with stamps), span of stamps {us[:, 7].max():.1f} us")
d = np.diff(us, axis=1)
print("phase              median    p10     p90     max   (us)")
for k, nm in enumerate(names):
    x = d[:, k]
    print(f"{nm:16s} {np.median(x):8.2f} {np.percentile(x, 10):7.2f} {np.percentile(x, 90):7.2f} {x.max():7.2f}")
life = us[:, 7] - us[:, 0]
print(f"tile life        {np.median(life):8.2f} {np.percentile(life, 10):7.2f} {np.percentile(life, 90):7.2f} {life.max():7.2f}")
# concurrency: how many tiles are alive over time
T = us[:, 7].max()
edges = np.linspace(0, T, 11)
print("time slice (us)     tiles alive (avg)   in loads   in look-back")
for i in range(10):
    lo, hi = edges[i], edges[i + 1]
    def ov(s, e):
        return np.clip(np.minimum(e, hi) - np.maximum(s, lo), 0, None).sum() / (hi - lo)
    print(f"{lo:8.1f}-{hi:8.1f}   {ov(us[:, 0], us[:, 7]):10.1f} {ov(us[:, 1], us[:, 2]):10.1f} {ov(us[:, 4], us[:, 5]):10.1f}")
# start order vs tile number: do tiles start in ticket order, and how far ahead of its predecessor's prefix does a tile look back?
order = np.argsort(us[:, 1])
print(f"tiles whose look-back ended before the previous tile's did: {(np.diff(us[:, 5]) < 0).mean() * 100:.1f} %")
print(f"prefix frontier: tile i's look-back ends {np.median(us[1:, 5] - us[:-1, 5]) * 1e3:.1f} ns after tile i-1's (median) -> {1.0 / max(np.median(us[1:, 5] - us[:-1, 5]), 1e-9):.1f} tiles/us; needed for 6 TB/s: {6e6 / (TILE * (12 + 12 * a.keep)):.1f} tiles/us")

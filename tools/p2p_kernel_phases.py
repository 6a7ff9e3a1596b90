#!/usr/bin/env python3
"""Phase stamps of the small-pass kernels (diagnostic build) over a few point-to-point iterations at 1k<->1k and over one
reference-shaped request.  MI355_ICP_LIB=build/diag/libmi355_icp.so python tools/p2p_kernel_phases.py"""
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

lib = pa._lib.lib
lib.mi355_icp_diag_p2p_read.argtypes = [C.c_void_p, C.c_void_p]
names = {1: ("resolve_accumulate<sums>", ["descr + key/src loads issued", "chunks + products", "block sum", "end"]),
         2: ("resolve_accumulate<idx>", ["descr + key/src loads issued", "chunk", "end"]),
         3: ("nn_scan_kernel (block 0)", ["descr, transforms, tile 0", "tiles scanned", "keys out"]),
         4: ("resolve_xfilter (wave 0 of every 16th block)", ["descr, key rows, transform", "first granules", "second granules", "end"]),
         5: ("overflow_tiles (block 0 of a job, nothing listed)", ["descr + count", "end"]),
         6: ("overflow_tiles (block 0 of a job, first listed point; n = listed)", ["entry, point, bound", "tile table", "tiles scanned", "reduce + store"])}


def dump(title):
    buf = np.zeros((4096, 16), dtype=np.uint64)
    n = C.c_uint(0)
    assert lib.mi355_icp_diag_p2p_read(buf.ctypes.data, C.byref(n)) == 0
    print(f"--- {title}: {n.value} records")
    groups = {}
    for r in buf[:min(n.value, 4096)]:
        kid, npts = int(r[0]) & 0xFF, (int(r[0]) >> 8) & 0xFFFFFFFF
        extra = (int(r[0]) >> 40) & 0xFF, int(r[0]) >> 48
        cnt = int(r[1])
        t = r[2:2 + cnt].astype(np.int64)
        groups.setdefault((kid, npts, extra[1]), []).append((np.diff(t) / 100.0, extra[0]))
    for (kid, npts, rows), recs in sorted(groups.items()):
        nm, ph = names[kid]
        d = np.median(np.stack([x for x, _ in recs]), axis=0)
        mx = np.max(np.stack([x for x, _ in recs]), axis=0)
        more = f"  key rows {rows}, second-granule points per wave median {np.median([e for _, e in recs]):.0f} max {max(e for _, e in recs)}" if kid == 4 else (f"  tiles {rows}, reachable median {np.median([e for _, e in recs]):.0f} max {max(e for _, e in recs)}" if kid == 6 else "")
        print(f"{nm:26s} n={npts:6d} x{len(recs):4d}  total {d.sum():6.2f} us (max {mx.sum():6.2f}) | " + "  ".join(f"{p} {x:5.2f} (max {m:5.2f})" for p, x, m in zip(ph, d, mx)) + more)


src, tgt, gt = st.make_pair(pa.synth_cloud, 1000, 1000)
with pa.IcpEngine(0) as eng:
    eng.set_targets([tgt])
    eng.set_sources([src])
    eng.align(np.eye(4)[None], pa.default_params(max_iterations=20))
    assert lib.mi355_icp_diag_p2p_begin() == 0
    eng.align(np.eye(4)[None], pa.default_params(max_iterations=4))
    dump("1k<->1k, 4 point-to-point iterations")
T = st.perturbation(-1)
tgt = pa.synth_cloud(st.TARGET_SEED, 100_000, normals=True)
src = pa.synth_cloud(st.SOURCE_SEED, 20_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
gt = np.linalg.inv(T)
inits = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
with pa.IcpEngine(0) as eng:
    eng.set_targets([tgt])
    eng.set_sources([src])
    for _ in range(3):
        eng.align_reference(inits, 250, 0.005, 2.5, 8)
    assert lib.mi355_icp_diag_p2p_begin() == 0
    eng.align_reference(inits, 250, 0.005, 2.5, 8)
    dump("reference request, coarse levels")

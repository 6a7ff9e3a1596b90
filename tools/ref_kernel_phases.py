#!/usr/bin/env python3
"""Phase stamps of the reference-mode kernels (diagnostic build: make -C pose_estimation_amd/csrc diag) over one request of the
reference's shape.  MI355_ICP_LIB=build/diag/libmi355_icp.so python tools/ref_kernel_phases.py"""
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
    assert lib.mi355_icp_diag_ref_begin() == 0
    eng.align_reference(inits, 250, 0.005, 2.5, 8)
    buf = np.zeros((4096, 16), dtype=np.uint64)
    n = C.c_uint(0)
    lib.mi355_icp_diag_ref_read.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.mi355_icp_diag_ref_read(buf.ctypes.data, C.byref(n)) == 0
names = {1: ("rejection_select", ["cache", "median", "MAD", "end"]), 2: ("rejection_select+scatter", ["cache", "median", "MAD", "scatter"]),
         3: ("ref_accumulate (block 0)", ["descr, idx, d2 arrive", "owner + rows requested", "owner + rows arrive", "products", "block sum", "end"])}
for r in buf[:min(n.value, 4096)]:
    kid, npts, nblk = int(r[0]) & 0xFF, (int(r[0]) >> 8) & 0xFFFFFFFF, int(r[0]) >> 40
    cnt = int(r[1])
    t = r[2:2 + cnt].astype(np.int64)
    d = np.diff(t) / 100.0
    nm, ph = names[kid]
    if len(d) != len(ph):  # the select kernels stamp inside their radix passes too (histogram walk, bin scan, candidates): print them raw
        print(f"{nm:26s} n={npts:6d} blocks={nblk:3d}  total {((t[-1] - t[0]) / 100.0):6.2f} us | stamps " + " ".join(f"{x:5.2f}" for x in d))
        continue
    print(f"{nm:26s} n={npts:6d} blocks={nblk:3d}  total {((t[-1] - t[0]) / 100.0):6.2f} us | " + "  ".join(f"{p} {x:5.2f}" for p, x in zip(ph, d)))

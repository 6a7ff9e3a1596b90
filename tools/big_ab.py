#!/usr/bin/env python3
"""One-off large A/B: every scan path against the all-exact scan, bit for bit, at sizes beyond the benchmark's
(3M targets = 2930 tiles, more than the pruned scan's candidate list holds; 700k sources)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

ns, nt = 700_000, 3_000_000
tgt = pa.synth_cloud(st.TARGET_SEED, nt)
src = pa.synth_cloud(st.SOURCE_SEED, ns, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=st.perturbation(-1))
ref = None
for mode in ("exact", "xbf16", "xfilter", "fused", "pruned"):
    os.environ["MI355_ICP_SCAN_MODE"] = mode
    with pa.IcpEngine(0) as e:
        e.set_targets([tgt])
        e.set_sources([src])
        e.correspond(np.eye(4)[None])
        t0 = time.perf_counter()
        idx, d2, sums = e.correspond(np.eye(4)[None])
        dt = time.perf_counter() - t0
    if ref is None:
        ref = (idx, d2, sums)
    same = np.array_equal(idx, ref[0]) and np.array_equal(d2.view(np.uint32), ref[1].view(np.uint32))
    print(f"{mode:8s} {dt * 1e3:9.2f} ms  identical to exact: {same}  sums rel diff {np.abs(sums - ref[2]).max() / np.abs(ref[2]).max():.1e}", flush=True)
    assert same

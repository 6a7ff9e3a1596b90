#!/usr/bin/env python3
"""Where a large correspondence pass spends its time, by scan path: 700k <-> 3M (tools/big_ab.py's size) or --ns/--nt.
    python tools/big_phases.py [--ns 700000 --nt 3000000] [modes ...]      (MI355_ICP_LIB picks the library)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=700_000)
ap.add_argument("--nt", type=int, default=3_000_000)
ap.add_argument("--no-profile", action="store_true", help="wall time only (the engine records no events)")
ap.add_argument("modes", nargs="*", default=["xbf16", "pruned"])
a = ap.parse_args()
tgt = pa.synth_cloud(st.TARGET_SEED, a.nt)
src = pa.synth_cloud(st.SOURCE_SEED, a.ns, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=st.perturbation(-1))
for mode in a.modes:
    os.environ["MI355_ICP_SCAN_MODE"] = mode
    with pa.IcpEngine(0) as e:
        e.set_targets([tgt])
        e.set_sources([src])
        e.correspond(np.eye(4)[None])
        if not a.no_profile:
            e.set_profiling(True)
        e.reset_profile()
        t0 = time.perf_counter()
        e.correspond(np.eye(4)[None])
        dt = time.perf_counter() - t0
        pr = e.profile()
        print(f"{mode:8s} wall {dt * 1e3:8.2f} ms | scan {pr.nn_scan_ms:8.3f} resolve {pr.resolve_ms:8.3f} reduce {pr.reduce_ms:7.3f} | overflow points {pr.overflow_points} "
              f"| launches {pr.nn_scan_launches}")

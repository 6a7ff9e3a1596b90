#!/usr/bin/env python3
"""A/B of scan-kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24).
Variants are engine-creation-time settings read from the environment by csrc/icp_engine.hip.
    python tools/scan_ab.py [--n 1048576] [--rounds 4] "PPL=4" "PPL=8" "PPL=4,TARGET_BLOCKS=12288" ...
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

KEYS = {"PPL": "MI355_ICP_SCAN_PPL", "TARGET_BLOCKS": "MI355_ICP_TARGET_BLOCKS", "MIN_TILES": "MI355_ICP_MIN_TILES_PER_SPLIT"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    nt = a.nt or a.n
    src, tgt, gt = st.make_pair(pa.synth_cloud, a.n, nt)
    engines = []
    for v in a.variants:
        for k in KEYS.values():
            os.environ.pop(k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            os.environ[KEYS[k]] = val
        e = pa.IcpEngine(0)
        e.set_targets([tgt])
        e.set_sources([src])
        engines.append(e)
    p = pa.default_params(max_iterations=1, flags=pa._lib.FLAG_PROFILE)
    times = [[] for _ in engines]
    ref = None
    for r in range(a.rounds + 1):
        for k, e in enumerate(engines):
            e.reset_profile()
            res = e.align(np.eye(4)[None], p)
            if ref is None:
                ref = res.poses[0]
            assert np.array_equal(res.poses[0], ref), "variants disagree"
            if r:  # round 0 = warm-up
                times[k].append(e.profile().nn_scan_ms)
    for v, t in zip(a.variants, times):
        med = statistics.median(t)
        print(f"{v:40s} scan median {med:9.3f} ms  min {min(t):9.3f}  -> {a.n * nt / med / 1e9:8.1f} Gpairs/s  ({len(t)} rounds)")


if __name__ == "__main__":
    main()

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

KEYS = {"MODE": "MI355_ICP_SCAN_MODE", "PPL": "MI355_ICP_SCAN_PPL", "TARGET_BLOCKS": "MI355_ICP_TARGET_BLOCKS", "MIN_TILES": "MI355_ICP_MIN_TILES_PER_SPLIT", "SHAPE": "MI355_ICP_XBF16_SHAPE", "SORT": "MI355_ICP_SORT_SOURCES"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--jobs", type=int, default=1, help="batched: this many objects, each n x nt (BASELINE config[3] style)")
    ap.add_argument("--sorted-source", action="store_true", help="upload the source clouds in Morton order (what-if: lanes of a wave hold neighbouring points)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    nt = a.nt or a.n
    pairs = [st.make_pair(pa.synth_cloud, a.n, nt, object_index=(k if a.jobs > 1 else -1)) for k in range(a.jobs)]
    if a.sorted_source:
        def morton(p):
            q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.uint64)
            def spread(v):
                v = (v | (v << 16)) & 0x030000FF
                v = (v | (v << 8)) & 0x0300F00F
                v = (v | (v << 4)) & 0x030C30C3
                v = (v | (v << 2)) & 0x09249249
                return v
            return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
        pairs = [(np.ascontiguousarray(s[morton(s)]), t, g) for s, t, g in pairs]
    engines = []
    for v in a.variants:
        for k in KEYS.values():
            os.environ.pop(k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            os.environ[KEYS[k]] = val
        e = pa.IcpEngine(0)
        e.set_targets([p[1] for p in pairs])
        e.set_sources([p[0] for p in pairs])
        engines.append(e)
    p = pa.default_params(max_iterations=1, flags=pa._lib.FLAG_PROFILE)
    times = [[] for _ in engines]
    other = [[] for _ in engines]
    over = [0 for _ in engines]
    ref = None
    for r in range(a.rounds + 1):
        for k, e in enumerate(engines):
            e.reset_profile()
            ids = list(range(a.jobs))
            res = e.align(np.stack([np.eye(4)] * a.jobs), p, source_ids=ids, target_ids=ids)
            if ref is None:
                ref = res.poses.copy()
            assert np.array_equal(res.poses, ref), "variants disagree"
            if r:  # round 0 = warm-up
                pr = e.profile()
                times[k].append(pr.nn_scan_ms)
                other[k].append(pr.resolve_ms + pr.reduce_ms)
                over[k] = pr.overflow_points
    print(f"# {a.jobs} job(s) x {a.n} source x {nt} target")
    for v, t, o, ov in zip(a.variants, times, other, over):
        med = statistics.median(t)
        print(f"{v:44s} scan median {med:9.3f} ms  min {min(t):9.3f}  -> {a.jobs * a.n * nt / med / 1e9:8.3f} Tpairs/s"
              f"  resolve+reduce {statistics.median(o):7.3f} ms  overflow points/pass {ov}  ({len(t)} rounds)")


if __name__ == "__main__":
    main()

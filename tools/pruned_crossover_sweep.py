#!/usr/bin/env python3
"""Where does the exact PRUNED scan beat the brute-force scan for a request's passes?  One correspondence pass of 6 candidate
poses (1-5 degrees off the alignment) of an n_model-point model against an n_scene-point scene, scan kernel time from the
engine's HIP events, planner default (brute force: all-exact / matrix-core filter) against MI355_ICP_SCAN_MODE=pruned.
Feeds the per-level choice of mi355_icp_align_reference (icp_engine.hip: service_prune)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402


def engine(mode):
    old = os.environ.get("MI355_ICP_SCAN_MODE")
    if mode:
        os.environ["MI355_ICP_SCAN_MODE"] = mode
    try:
        return pa.IcpEngine(0)
    finally:
        if mode:
            if old is None:
                os.environ.pop("MI355_ICP_SCAN_MODE")
            else:
                os.environ["MI355_ICP_SCAN_MODE"] = old


def main():
    T = st.perturbation(-1)
    gt = np.linalg.inv(T)
    cands = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    jobs = int(os.environ.get("JOBS", "6"))
    with engine(None) as brute, engine("pruned") as pruned:
        for nt in (15_000, 30_000, 60_000, 120_000, 240_000, 475_000, 950_000):
            tgt = pa.synth_cloud(st.TARGET_SEED, nt)
            for e in (brute, pruned):
                e.set_targets([tgt])
            for ns in (1_500, 3_000, 6_000, 12_500, 25_000, 50_000, 100_000):
                if float(ns) * nt * jobs > 1.3e12:
                    continue
                src = pa.synth_cloud(st.SOURCE_SEED, ns, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
                row = {"jobs": jobs, "n_model": ns, "n_scene": nt, "pairs": float(ns) * nt * jobs}
                for name, e in (("brute", brute), ("pruned", pruned)):
                    e.set_sources([src])
                    e.set_profiling(True)
                    for rep in range(4):
                        if rep == 1:
                            e.reset_profile()
                        e.correspond(cands[:jobs], 0.0, [0] * jobs, [0] * jobs, want_pairs=False)
                    p = e.profile()
                    row[name + "_scan_ms"] = p.nn_scan_ms / max(p.nn_scan_launches, 1)
                    row[name + "_rest_ms"] = (p.resolve_ms + p.reduce_ms) / max(p.nn_scan_launches, 1)
                row["pruned_wins"] = row["pruned_scan_ms"] + row["pruned_rest_ms"] < row["brute_scan_ms"] + row["brute_rest_ms"]
                print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

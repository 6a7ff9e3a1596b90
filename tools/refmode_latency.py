#!/usr/bin/env python3
"""Steady-state latency of reference mode at the reference's own request shape: <= 6 PPF candidates of one model against
one scene, ICP(250, 0.005f, 2.5f, 8) (opencv_surface_match.cpp:85-94).  Un-profiled wall clock (median of --reps calls on
resident clouds), then one profiled call for the split between the NN scan and everything else.
    python tools/refmode_latency.py [--model 20000] [--scene 100000] [--candidates 6] [--reps 20]"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=int, default=20_000)
    ap.add_argument("--scene", type=int, default=100_000)
    ap.add_argument("--candidates", type=int, default=6)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--legacy", action="store_true", help="the round 1-3 form of reference mode (flags = 0) instead of MI355_ICP_FLAG_UPSTREAM")
    ap.add_argument("--shift", type=float, nargs=3, default=(0.0, 0.0, 0.0), help="move the scene (and the candidates) away from the camera origin")
    a = ap.parse_args()
    flags = 0 if a.legacy else pa._lib.FLAG_UPSTREAM
    T = st.perturbation(-1)
    tgt = pa.synth_cloud(st.TARGET_SEED, a.scene, normals=True)
    src = pa.synth_cloud(st.SOURCE_SEED, a.model, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)  # N x 6, as the reference's models
    S = np.eye(4)
    S[:3, 3] = a.shift
    tgt[:, :3] += np.asarray(a.shift, dtype=np.float32)
    gt = S @ np.linalg.inv(T)
    cands = [S] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0)]
    inits = np.stack(cands[:a.candidates])
    with pa.IcpEngine(0) as eng:
        eng.set_targets([tgt])
        eng.set_sources([src])
        for _ in range(3):
            res = eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
        times = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            res = eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
            times.append(time.perf_counter() - t0)
        eng.set_profiling(True)
        eng.reset_profile()
        eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
        prof = eng.profile()
    gpu = prof.nn_scan_ms + prof.resolve_ms
    print(json.dumps({"workload": f"{a.candidates} candidates, {a.model} model vs {a.scene} scene, ICP(250, 0.005, 2.5, 8)",
                      "form": "round 1-3 (flags 0)" if a.legacy else "upstream-faithful (MI355_ICP_FLAG_UPSTREAM)",
                      "wall_ms_median": 1e3 * statistics.median(times), "wall_ms_min": 1e3 * min(times), "reps": a.reps,
                      "passes": res.passes.tolist(), "passes_max": int(res.passes.max()),
                      "profiled_scan_ms_total": prof.nn_scan_ms, "profiled_other_kernels_ms_total": prof.resolve_ms,
                      "other_share_of_gpu_time": prof.resolve_ms / gpu if gpu else None,
                      "pose_error_vs_truth_max": float(max(np.linalg.norm(p - gt) for p in res.poses))}), flush=True)


if __name__ == "__main__":
    main()

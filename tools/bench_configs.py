#!/usr/bin/env python3
"""Runs every single-GPU BASELINE.json config through the engine and prints one JSON line each:
C1 1k<->1k (bunny stand-in), C2 100k<->100k x 50 iterations, C3 1M<->1M, C4 64 x (50k<->50k) batched.
Each line carries time per iteration, evaluated pairs/s and the Frobenius distance of the final pose(s)
to the CPU oracle's on the same inputs (the parity gate of SURVEY.md 8d).  (C5 8M<->1M across 8 GPUs
is bench.py --gpus 8.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import oracle_lib as ora  # noqa: E402
import synth_twin as st  # noqa: E402

THREADS = min(ora.max_threads(), len(os.sched_getaffinity(0)), 16)


def run(name, n_jobs, ns, nt, iters, oracle_iters=None):
    pairs = [st.make_pair(pa.synth_cloud, ns, nt, object_index=(k if n_jobs > 1 else -1)) for k in range(n_jobs)]
    with pa.IcpEngine(0) as eng:
        t0 = time.perf_counter()
        eng.set_targets([p[1] for p in pairs])
        eng.set_sources([p[0] for p in pairs])
        upload = time.perf_counter() - t0
        ids = list(range(n_jobs))
        init = np.stack([np.eye(4)] * n_jobs)
        eng.align(init, pa.default_params(max_iterations=2), ids, ids)  # warm-up
        t0 = time.perf_counter()
        res = eng.align(init, pa.default_params(max_iterations=iters), ids, ids)   # the timed run: no event records in the loop
        dt = time.perf_counter() - t0
        eng.reset_profile()
        eng.align(init, pa.default_params(max_iterations=iters, flags=pa._lib.FLAG_PROFILE), ids, ids)  # the kernel split
        prof = eng.profile()
    out = {"config": name, "jobs": n_jobs, "n_source": ns, "n_target": nt, "iterations": iters,
           "ms_per_iteration": 1e3 * dt / iters, "pairs_per_s": n_jobs * float(ns) * nt * iters / dt,
           "iters_per_s": iters / dt, "scan_ms": prof.nn_scan_ms / iters, "resolve_ms": prof.resolve_ms / iters,
           "reduce_ms": prof.reduce_ms / iters, "host_and_launch_ms": 1e3 * dt / iters - (prof.nn_scan_ms + prof.resolve_ms + prof.reduce_ms) / iters,
           "upload_s": upload, "upload_inclusive_pairs_per_s": n_jobs * float(ns) * nt * iters / (dt + upload)}
    oi = oracle_iters if oracle_iters is not None else iters
    if oi:
        po = pa.default_params(max_iterations=oi)
        worst = 0.0
        t0 = time.perf_counter()
        if oi != iters:
            with pa.IcpEngine(0) as eng:
                eng.set_targets([p[1] for p in pairs])
                eng.set_sources([p[0] for p in pairs])
                res = eng.align(init, po, ids, ids)
        for k in range(n_jobs):
            o = ora.icp(pairs[k][0], pairs[k][1], np.eye(4), ora.params_from(po), nthreads=THREADS)
            worst = max(worst, float(np.linalg.norm(res.poses[k] - o["pose"])))
        out["oracle_iterations"] = oi
        out["pose_vs_oracle_frobenius_max"] = worst
        out["oracle_s"] = time.perf_counter() - t0
        out["oracle_threads"] = THREADS
        assert worst <= 1e-5, worst
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run("C1 1k<->1k", 1, 1000, 1000, 50)
    run("C2 100k<->100k x50", 1, 100_000, 100_000, 50)
    run("C3 1M<->1M", 1, 1 << 20, 1 << 20, 5, oracle_iters=5)
    run("C4 64x(50k<->50k)", 64, 50_000, 50_000, 20, oracle_iters=20)

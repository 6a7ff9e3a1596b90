#!/usr/bin/env python3
"""Scene normals (k = 20 PCA, computeNormalsPC3d semantics) on the GPU beside the CPU oracle (exact KD-tree k-NN)."""
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

if __name__ == "__main__":
    with pa.IcpEngine(0) as eng:
        eng.set_profiling(True)
        for n in (20_000, 100_000, 500_000, 1 << 20):
            xyz = pa.synth_cloud(st.TARGET_SEED, n)
            eng.scene_load(xyz)
            eng.scene_normals(20)  # warm-up
            eng.reset_profile()
            t0 = time.perf_counter()
            nrm = eng.scene_normals(20, True, (0.0, 0.0, 0.0))
            dt = time.perf_counter() - t0
            p = eng.profile()
            out = {"points": n, "k": 20, "gpu_total_s": dt, "knn_scan_ms": p.knn_scan_ms, "pca_ms": p.normals_ms,
                   "pairs_per_s": float(n) * n / (p.knn_scan_ms * 1e-3)}
            if n <= 500_000:
                t0 = time.perf_counter()
                ref = ora.normals(xyz, 20, True, (0, 0, 0), nthreads=THREADS)
                out["cpu_oracle_s"] = time.perf_counter() - t0
                out["cpu_threads"] = THREADS
                out["max_abs_diff"] = float(np.abs(ref - nrm).max())
                out["bitwise_equal_fraction"] = float((ref.view(np.uint32) == nrm.view(np.uint32)).all(1).mean())
            print(json.dumps(out), flush=True)

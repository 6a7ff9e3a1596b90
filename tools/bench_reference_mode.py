#!/usr/bin/env python3
"""Reference mode (cv::ppf_match_3d::ICP(250, 0.005f, 2.5f, 8) semantics: pyramid + MAD rejection + picky +
point-to-plane) on the GPU next to the CPU oracle, one JSON line per workload.  The first workload is the shape
of the reference's own request: <= 6 PPF candidates of one model against one scene
(opencv_surface_match.cpp:85-94)."""
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


def run(name, ns, nt, inits, check=True):
    T = st.perturbation(-1)
    tgt = pa.synth_cloud(st.TARGET_SEED, nt, normals=True)
    src = pa.synth_cloud(st.SOURCE_SEED, ns, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    gt = np.linalg.inv(T)
    inits = np.stack([i @ gt if k else np.eye(4) for k, i in enumerate(inits)])
    with pa.IcpEngine(0) as eng:
        eng.set_targets([tgt])
        eng.set_sources([src])
        eng.align_reference(inits, 8, 0.005, 2.5, 8)  # warm-up with the same batch: builds the pyramid, sizes the work buffers
        eng.set_profiling(True)
        eng.reset_profile()
        t0 = time.perf_counter()
        res = eng.align_reference(inits, 250, 0.005, 2.5, 8)
        dt = time.perf_counter() - t0
        prof = eng.profile()
    out = {"workload": name, "n_model": ns, "n_scene": nt, "candidates": len(inits), "gpu_s": dt,
           "passes": res.passes.tolist(), "pairs_evaluated": prof.pairs_evaluated,
           "scan_ms_total": prof.nn_scan_ms, "other_kernels_ms_total": prof.resolve_ms,
           "pose_error_vs_truth": [float(np.linalg.norm(p - gt)) for p in res.poses]}
    if check:
        t0 = time.perf_counter()
        worst = 0.0
        for k in range(len(inits)):
            o = ora.icp_reference(src, tgt, inits[k], 250, 0.005, 2.5, 8, nthreads=THREADS)
            assert o["passes"] == res.passes[k], (k, o["passes"], res.passes[k])
            worst = max(worst, float(np.linalg.norm(o["pose"] - res.poses[k])))
        out["cpu_oracle_s"] = time.perf_counter() - t0
        out["cpu_threads"] = THREADS
        out["pose_vs_oracle_frobenius_max"] = worst
        assert worst <= 1e-5
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    cands = [np.eye(4)] + [st.rigid_transform(a, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) for a in (1.0, 2.0, 3.0, 4.0, 5.0)]
    run("6 candidates, 20k model vs 100k scene", 20_000, 100_000, cands)
    run("1 pose, 100k vs 100k", 100_000, 100_000, cands[:1])
    run("6 candidates, 100k model vs 500k scene", 100_000, 500_000, cands)
    run("1 pose, 1M vs 1M", 1 << 20, 1 << 20, cands[:1])

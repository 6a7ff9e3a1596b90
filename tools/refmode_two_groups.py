#!/usr/bin/env python3
"""Experiment: the 6 candidates of one request split over G engines driven by G host threads (each its own stream), to
see what overlapping one group's small dependent kernels with the other group's scans buys at the request shape."""
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = st.perturbation(-1)
    tgt = pa.synth_cloud(st.TARGET_SEED, 100_000, normals=True)
    src = pa.synth_cloud(st.SOURCE_SEED, 20_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    gt = np.linalg.inv(T)
    cands = [np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)]
    inits = np.stack(cands)
    parts = np.array_split(np.arange(6), groups)
    engines = [pa.IcpEngine(0) for _ in range(groups)]
    for e in engines:
        e.set_targets([tgt])
        e.set_sources([src])
    out = [None] * groups

    def work(g):
        out[g] = engines[g].align_reference(inits[parts[g]], 250, 0.005, 2.5, 8)

    def once():
        th = [threading.Thread(target=work, args=(g,)) for g in range(1, groups)]
        for t in th:
            t.start()
        work(0)
        for t in th:
            t.join()

    for _ in range(3):
        once()
    times = []
    for _ in range(30):
        t0 = time.perf_counter()
        once()
        times.append(time.perf_counter() - t0)
    print(json.dumps({"groups": groups, "wall_ms_median": 1e3 * statistics.median(times), "wall_ms_min": 1e3 * min(times),
                      "passes": [o.passes.tolist() for o in out]}), flush=True)


if __name__ == "__main__":
    main()

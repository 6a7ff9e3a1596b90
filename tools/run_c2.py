#!/usr/bin/env python3
"""BASELINE config 2 alone (100k<->100k, 50 iterations), for per-kernel traces: rocprofv3 --kernel-trace --stats -- python3 tools/run_c2.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402

src, tgt, gt = st.make_pair(pa.synth_cloud, 100_000, 100_000)
with pa.IcpEngine(0) as eng:
    eng.set_targets([tgt])
    eng.set_sources([src])
    for _ in range(3):
        res = eng.align(np.eye(4)[None], pa.default_params(max_iterations=50))
    print(res.iterations, np.linalg.norm(res.poses[0] - gt))

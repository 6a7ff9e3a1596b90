#!/usr/bin/env python3
"""Reference-mode ICP of one pose at 1M<->1M, for per-kernel traces:
rocprofv3 --kernel-trace --stats -- python3 tools/run_ref1m.py [pruned]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402
from pose_estimation_amd import _lib  # noqa: E402

n = 1 << 20
T = st.perturbation(-1)
tgt = pa.synth_cloud(st.TARGET_SEED, n, normals=True)
src = pa.synth_cloud(st.SOURCE_SEED, n, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
flags = _lib.FLAG_PRUNED_SCAN if len(sys.argv) > 1 and sys.argv[1] == "pruned" else 0
with pa.IcpEngine(0) as eng:
    eng.set_targets([tgt])
    eng.set_sources([src])
    for _ in range(2):
        res = eng.align_reference(np.eye(4)[None], 250, 0.005, 2.5, 8, flags=flags)
    print(res.passes, np.linalg.norm(res.poses[0] - np.linalg.inv(T)))

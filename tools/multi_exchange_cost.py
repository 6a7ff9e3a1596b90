#!/usr/bin/env python3
"""What the in-process exchange of a multi-device handle costs per pass ("8 device->host copies + host sum, to be
measured against RCCL"): the same slab-sharded point-to-point alignment on a single engine and on handles over {0,0} and
{0,0,0} (this pool's boxes have one GPU: the slots share it, so the scan work is the same and the difference per iteration is
the exchange + the second engine's launches), at a size where a pass is short."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pose_estimation_amd as pa  # noqa: E402
import synth_twin as st  # noqa: E402


def run(eng, src, tgt, iters, ref_flags=None, reps=5):
    eng.set_targets([tgt])
    eng.set_sources([src])
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        if ref_flags is None:
            r = eng.align(np.eye(4)[None], pa.default_params(max_iterations=iters))
            n = int(r.iterations[0])
        else:
            r = eng.align_reference(np.eye(4)[None], 250, 0.005, 2.5, 8, flags=ref_flags)
            n = int(r.passes[0])
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return 1e3 * best / n, n


def main():
    for ns, nt, iters in ((20_000, 20_000, 50), (100_000, 100_000, 50), (1 << 20, 1 << 20, 3)):
        src, tgt, gt = st.make_pair(pa.synth_cloud, ns, nt)
        tgt6 = pa.synth_cloud(st.TARGET_SEED, nt, normals=True)
        src6 = pa.synth_cloud(st.SOURCE_SEED, ns, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=st.perturbation(-1), normals=True)
        row = {"n_source": ns, "n_target": nt}
        for name, ids in (("single", None), ("multi_0_0", [0, 0]), ("multi_0_0_0", [0, 0, 0])):
            with (pa.IcpEngine(0) if ids is None else pa.IcpEngine(device_ids=ids)) as eng:
                row[name + "_p2p_ms_per_iteration"], _ = run(eng, src, tgt, iters)
                if ns <= 100_000:
                    row[name + "_reference_ms_per_pass"], row[name + "_reference_passes"] = run(eng, src6, tgt6, 0, ref_flags=pa._lib.FLAG_UPSTREAM)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

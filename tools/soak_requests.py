#!/usr/bin/env python3
"""Soak: thousands of requests with scenes of changing size through one engine - device memory must level off (the buffer
pool reuses, nothing leaks), every request must succeed, and a repeated scene must give the same bits at the end as at the
start.  python tools/soak_requests.py [requests] [device slots of the handle, default 1: a plain context; n > 1: mi355_icp_create_multi
over n slots of device 0, requests sharded by candidate] [idle ms before each request, default 0: back to back]"""
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


def free_bytes():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # the device left idle this long before every request (a camera's frame period)
    rng = np.random.default_rng(7)
    T = st.perturbation(-1)
    model = pa.synth_cloud(st.SOURCE_SEED, 20_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    gt = np.linalg.inv(T)
    inits = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    big = pa.synth_cloud(st.TARGET_SEED, 130_000)
    mem, first, t0, lat = [], None, time.perf_counter(), []
    with (pa.IcpEngine(0) if slots == 1 else pa.IcpEngine(device_ids=[0] * slots)) as eng:
        eng.set_sources([model])
        for r in range(n_req):
            n = 100_000 if r in (0, n_req - 1) else int(rng.integers(70_000, 130_000))
            if idle_ms > 0:
                time.sleep(idle_ms * 1e-3)
            t_req = time.perf_counter()
            eng.scene_load(big[:n])
            eng.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
            eng.scene_normals(20, True, (0.0, 0.0, 0.0))
            eng.scene_commit_target()
            res = eng.align_reference(inits, 250, 0.005, 2.5, 8)
            assert (np.asarray(res.status) == 0).all(), r
            lat.append((time.perf_counter() - t_req) * 1e3)
            if r == 0:
                first = res
            if r % 250 == 0 or r == n_req - 1:
                mem.append(free_bytes())
        same = np.array_equal(first.poses.view(np.uint64), res.poses.view(np.uint64)) and np.array_equal(first.passes, res.passes)
    print(json.dumps({"requests": n_req, "device_slots": slots, "idle_ms_before_each_request": idle_ms, "seconds": time.perf_counter() - t0, "first_equals_last_bitwise": bool(same),
                      "request_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(np.max(lat)),
                                     "slowest_requests": [[int(i), round(float(lat[i]), 2)] for i in np.argsort(lat)[::-1][:8]]},
                      "free_mib_every_250": [m >> 20 for m in mem],
                      "free_mib_spread_after_warmup": (max(mem[2:]) - min(mem[2:])) >> 20 if len(mem) > 3 else None}))
    assert same


if __name__ == "__main__":
    main()

"""The matrix-core filter's INVARIANT, not its consequence: for every source point and every run of
128 consecutive targets of the Morton-sorted copy, the lower bound L the scan computes must not exceed the exact float32
minimum of that run.  L comes from mi355_icp_debug_filter_bounds - the production scan kernel instantiated with a dump
switch (csrc/icp_fused_kernels.hip, kDump) - and the exact minima from the contract formula in float32 (the oracle's
dist2: every product and sum rounded separately), evaluated here for EVERY (point, run).  Clouds: the adversarial sets of
test_gpu_fused_filter.py / test_gpu_filter_stress.py (near-ties, duplicates, far from the origin, flattened, two distant
clusters, subnormal and overflowing distances, random geometry)."""
import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu


def exact_run_minima(moved, tgt, runs):
    """[n_runs, n_points] float32: min over the run's targets of fl(fl(fl(dx*dx) + fl(dy*dy)) + fl(dz*dz))."""
    out = np.full((runs.shape[0], moved.shape[0]), np.inf, dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        for g in range(runs.shape[0]):
            members = runs[g][runs[g] >= 0]
            if members.size == 0:
                continue
            t = tgt[members]
            dx = moved[:, None, 0] - t[None, :, 0]
            dy = moved[:, None, 1] - t[None, :, 1]
            dz = moved[:, None, 2] - t[None, :, 2]
            a = dx * dx
            a = a + dy * dy
            a = a + dz * dz
            a = np.where(np.isnan(a), np.float32(np.inf), a)
            out[g] = a.min(axis=1)
    return out


def check(engine, src, tgt, pose, expect_tight=False):
    src = np.ascontiguousarray(src, dtype=np.float32)
    tgt = np.ascontiguousarray(tgt, dtype=np.float32)
    engine.set_targets([tgt])
    engine.set_sources([src])
    lower, runs = engine.debug_filter_bounds(pose)
    assert lower.shape == (runs.shape[0], src.shape[0]) and runs.shape[1] == 128
    members = runs[runs >= 0]
    assert np.array_equal(np.sort(members), np.arange(tgt.shape[0]))            # the runs partition the targets
    assert not np.isnan(lower).any() and (lower >= 0).all()   # +inf only where a run is all padding (checked below: L <= exact)
    moved = ora.transform(pose, src)
    exact = exact_run_minima(moved, tgt, runs)
    # the restatement above IS the contract formula: its minimum over the runs must be the oracle's answer, bit for bit
    oi, od = ora.nn_brute(moved, tgt, nthreads=ora.max_threads())
    assert np.array_equal(exact.min(axis=0).view(np.uint32), od.view(np.uint32))
    bad = lower > exact
    assert not bad.any(), (f"{int(bad.sum())} (point, run) bounds exceed the exact run minimum; "
                           f"worst excess {float((lower - exact)[bad].max())}")
    if expect_tight:   # not vacuous: on a benign surface the bound sits within a per cent of the exact value
        finite = np.isfinite(exact) & (exact > 0)
        ratio = lower[finite] / exact[finite]
        assert np.median(ratio) > 0.99 and (ratio > 0.5).mean() > 0.99
    return lower, exact


def test_bound_on_the_benchmark_surface_is_valid_and_tight(engine):
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 3000, 12000)
    for pose in (np.eye(4), gt, st.rigid_transform(33.0, (0.3, -1, 0.2), (0.5, 0.1, -0.4))):
        check(engine, src, tgt, pose, expect_tight=True)


def test_bound_on_near_ties_duplicates_and_mirrored_targets(engine):
    rng = np.random.default_rng(0)
    u = rng.standard_normal((20000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    tgt = (u * (1.0 + rng.integers(-3, 4, (20000, 1)) * 6e-8)).astype(np.float32)
    src = (rng.standard_normal((500, 3)) * 1e-4).astype(np.float32)
    check(engine, src, tgt, np.eye(4))
    rng = np.random.default_rng(1)
    tgt = rng.standard_normal((9000, 3)).astype(np.float32) * 5
    src = rng.standard_normal((3000, 3)).astype(np.float32)
    off = (rng.standard_normal((3000, 3)) * 0.01).astype(np.float32)
    tgt[:3000] = src + off
    tgt[6000:9000] = src - off
    check(engine, src, tgt, np.eye(4))
    tgt[6000:9000] = tgt[:3000]
    tgt[3000:6000] = tgt[:3000]
    check(engine, src, tgt, np.eye(4))


def test_bound_far_from_the_origin_flattened_and_clustered(engine):
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 3000, 9000)
    shift = np.float32([50.0, -30.0, 20.0])
    check(engine, src + shift, tgt + shift, np.eye(4))
    flat = np.float32([1.0, 1.0, 1e-3])
    check(engine, src * flat, tgt * flat, np.eye(4))
    two = np.concatenate([tgt, tgt + np.float32([1000.0, 0, 0])]).astype(np.float32)
    two = two[np.random.default_rng(3).permutation(len(two))]
    check(engine, np.concatenate([src, src + np.float32([1000.0, 0, 0])]), two, np.eye(4))
    check(engine, src, tgt, st.rigid_transform(25.0, (0.3, 1, -0.2), (40.0, -10.0, 5.0)))


def test_bound_with_subnormal_and_overflowing_distances(engine):
    rng = np.random.default_rng(5)
    for scale in (1e-20, 1e-12, 1e18, 3e19):
        cloud = (rng.standard_normal((3000, 3)) * scale).astype(np.float32)
        q = (cloud[rng.permutation(3000)[:777]] * np.float32(1.001)).astype(np.float32)
        check(engine, q, cloud, np.eye(4))


def test_bound_on_random_geometry(engine):
    from test_gpu_filter_stress import random_cloud
    rng = np.random.default_rng(20261005)
    for case in range(8):
        ns, nt = int(rng.integers(100, 2500)), int(rng.integers(100, 15000))
        tgt = random_cloud(rng, nt)
        if rng.random() < 0.5:
            src = (tgt[rng.integers(0, nt, ns)] * np.float32(1.0 + 1e-3 * rng.standard_normal())).astype(np.float32)
            pose = st.rigid_transform(rng.uniform(0, 0.5), rng.standard_normal(3), rng.standard_normal(3) * 1e-3 * np.abs(tgt).max())
        else:
            src = random_cloud(rng, ns)
            pose = st.rigid_transform(rng.uniform(0, 180), rng.standard_normal(3), rng.standard_normal(3))
        check(engine, src, tgt, pose)


def test_bound_where_the_dropped_low_products_are_largest(engine):
    """Term (h) of the filter's error budget (csrc/icp_fused_kernels.hip): the three low x low products the matrix-core
    contraction leaves out are largest - 2^-16 (|tc|^2 + |qc|^2) - when every coordinate of both points sits exactly half-way
    between two bfloat16 values just above a power of two (the low piece is then a full half ulp) and all low pieces agree in
    sign.  One group of four tiles, mirrored so that its centre is exactly 0: coordinates 1 + a/256 with a = 1 (mod 4)
    round DOWN to even (low piece +2^-8), sources from the same lattice (q' = -2 q: low piece -2^-7), so s overshoots the
    squared distance of the quantised points by the whole term on every pair."""
    a = 1.0 + (4.0 * np.arange(16) + 1.0) / 256.0
    b = 1.0 + (4.0 * np.arange(8) + 1.0) / 256.0
    pos = np.stack(np.meshgrid(a, a, b, indexing="ij"), axis=-1).reshape(-1, 3)
    tgt = np.concatenate([pos, -pos]).astype(np.float32)                      # 4096 targets, box symmetric about 0
    assert np.array_equal(tgt.astype(np.float64), np.concatenate([pos, -pos]))  # the lattice is exact in float32
    rng = np.random.default_rng(11)
    src = np.concatenate([pos[rng.permutation(len(pos))[:700]], -pos[rng.permutation(len(pos))[:324]]]).astype(np.float32)
    lower, exact = check(engine, src, tgt, np.eye(4))
    # not vacuous here either: away from the point's own lattice site (exact 0) the bound keeps most of the distance
    far = exact > 1e-3
    assert (lower[far] > 0.5 * exact[far]).mean() > 0.99
    # the same lattice at the other scales of a float32's exponent the tests above use
    for scale in (2.0 ** -20, 2.0 ** 12):
        check(engine, src * np.float32(scale), tgt * np.float32(scale), np.eye(4))

"""Randomised battery for the matrix-core filter path, the all-exact scan and the optional pruned scan: clouds of random
size, scale (1e-4 .. 1e4), offset from the origin, anisotropy and clustering, under random rigid poses.  Every case is
checked against the ORACLE (exact KD-tree, every point) and the three HIP paths against each other, bit for bit; this
battery is about the filter's error band (icp_fused_kernels.hip) on geometry the named tests do not cover.
MI355_ICP_STRESS_CASES raises the case count."""
import os

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("MI355_ICP_STRESS_CASES", "24"))


from gpu_util import engine_with_mode  # noqa: E402


def random_cloud(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:      # volume
        p = rng.standard_normal((n, 3))
    elif kind == 1:    # surface of a bumpy sphere
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        p = u * (1.0 + 0.1 * np.sin(5 * u[:, :1]) * np.cos(3 * u[:, 1:2]))
    elif kind == 2:    # a few tight clusters far apart
        centres = rng.standard_normal((rng.integers(2, 9), 3)) * 10.0
        p = centres[rng.integers(0, len(centres), n)] + rng.standard_normal((n, 3)) * 0.01
    else:              # jittered lattice: many near-ties
        side = int(np.ceil(n ** (1 / 3)))
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float64)
        p = g + rng.standard_normal((n, 3)) * 1e-4
    p = p * rng.uniform(0.2, 1.0, 3)                                   # anisotropy
    p = p * 10.0 ** rng.uniform(-4, 4) + rng.standard_normal(3) * 10.0 ** rng.uniform(-2, 3)
    return p.astype(np.float32)


def test_matrix_core_filter_equals_the_exact_scan_on_random_geometry():
    rng = np.random.default_rng(20261004)
    with engine_with_mode("xbf16") as fast, engine_with_mode("exact") as exact, engine_with_mode("pruned") as pruned:
        for case in range(CASES):
            ns, nt = int(rng.integers(100, 30000)), int(rng.integers(100, 60000))
            tgt = random_cloud(rng, nt)
            if rng.random() < 0.5:     # sources near the targets (the ICP situation) ...
                src = (tgt[rng.integers(0, nt, ns)] * np.float32(1.0 + 1e-3 * rng.standard_normal())).astype(np.float32)
                pose = st.rigid_transform(rng.uniform(0, 0.5), rng.standard_normal(3), rng.standard_normal(3) * 1e-3 * np.abs(tgt).max())
            else:                       # ... or anywhere
                src = random_cloud(rng, ns)
                pose = st.rigid_transform(rng.uniform(0, 180), rng.standard_normal(3), rng.standard_normal(3))
            out = []
            for eng in (fast, exact, pruned):
                eng.set_targets([tgt])
                eng.set_sources([src])
                out.append(eng.correspond(pose[None]))
            oi, od, osums = ora.correspond(pose, src, tgt, nthreads=ora.max_threads())       # the oracle, every point
            assert np.array_equal(out[1][0], oi), f"case {case}: all-exact scan differs from the oracle ({ns} x {nt})"
            assert np.array_equal(out[1][1].view(np.uint32), od.view(np.uint32)), f"case {case}: all-exact distances differ from the oracle"
            assert np.array_equal(out[0][0], out[1][0]), f"case {case}: indices differ ({ns} x {nt})"
            assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32)), f"case {case}: distances differ"
            np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-11, atol=0)
            assert np.array_equal(out[2][0], out[1][0]), f"case {case}: pruned-scan indices differ ({ns} x {nt})"
            assert np.array_equal(out[2][1].view(np.uint32), out[1][1].view(np.uint32)), f"case {case}: pruned-scan distances differ"


def test_pruned_scan_flag_gives_the_same_alignment(engine):
    """MI355_ICP_FLAG_PRUNED_SCAN per call (no environment involved): point-to-point and reference mode."""
    import pose_estimation_amd as pa
    from pose_estimation_amd import _lib
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 30000, 50000)
    tgt6 = st.synth_cloud_np(st.TARGET_SEED, 50000, normals=True)
    engine.set_targets([tgt6])
    engine.set_sources([src])
    a = engine.align(np.eye(4)[None], pa.default_params(max_iterations=12))
    b = engine.align(np.eye(4)[None], pa.default_params(max_iterations=12, flags=_lib.FLAG_PRUNED_SCAN))
    assert np.array_equal(a.poses, b.poses) and np.array_equal(a.iterations, b.iterations)
    ra = engine.align_reference(np.eye(4)[None], 60, 0.005, 2.5, 5)
    rb = engine.align_reference(np.eye(4)[None], 60, 0.005, 2.5, 5, flags=_lib.FLAG_PRUNED_SCAN)
    assert np.array_equal(ra.poses, rb.poses) and np.array_equal(ra.passes, rb.passes)


def test_icp_mirror_with_pruned_scan(engine):
    import pose_estimation_amd as pa
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 20000, 30000)
    poses_a, poses_b = [pa.Pose3D()], [pa.Pose3D()]
    pa.ICP(40, 0.005, 2.5, 8, engine=engine, mode="point_to_point").registerModelToScene(src, tgt, poses_a)
    pa.ICP(40, 0.005, 2.5, 8, engine=engine, mode="point_to_point", pruned_scan=True).registerModelToScene(src, tgt, poses_b)
    assert np.array_equal(poses_a[0].pose, poses_b[0].pose) and poses_a[0].residual == poses_b[0].residual


def test_reference_mode_is_the_same_through_every_scan_path():
    """align_reference (pyramid, rejection, picky, point-to-plane) on random surfaces under random initial poses: passes,
    residuals and poses identical whichever scan answers the nearest-neighbour queries."""
    rng = np.random.default_rng(77)
    cases = max(4, CASES // 6)
    with engine_with_mode("exact") as exact, engine_with_mode("xbf16") as fast, engine_with_mode("pruned") as pruned:
        for case in range(cases):
            ns, nt = int(rng.integers(3000, 40000)), int(rng.integers(5000, 80000))
            tgt6 = st.synth_cloud_np(st.TARGET_SEED + case, nt, normals=True)
            T = st.rigid_transform(rng.uniform(0.5, 4.0), rng.standard_normal(3), rng.standard_normal(3) * 0.02)
            src = st.synth_cloud_np(st.SOURCE_SEED + case, ns, noise_seed=st.NOISE_SEED + case, sigma=st.SIGMA, transform=T)
            init = np.stack([np.eye(4), st.rigid_transform(1.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015))])
            out = []
            for eng in (exact, fast, pruned):
                eng.set_targets([tgt6])
                eng.set_sources([src])
                out.append(eng.align_reference(init, 60, 0.005, 2.5, 6, source_ids=[0, 0], target_ids=[0, 0]))
            for other in out[1:]:
                assert np.array_equal(other.passes, out[0].passes), case
                assert np.array_equal(other.poses, out[0].poses) and np.array_equal(other.residual, out[0].residual), case

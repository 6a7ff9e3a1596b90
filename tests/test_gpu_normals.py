"""Scene normals on the GPU against the oracle: identical neighbour lists ((d2, index) order) and normals."""
import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu


def check(engine, xyz, k=20, flip=True, vp=(0.0, 0.0, 0.0), kd=True):
    engine.scene_load(xyz)
    n_gpu, nb = engine.scene_normals(k, flip, vp, want_neighbors=True)
    n_cpu, idx = ora.normals(xyz, k, flip, vp, use_kdtree=kd, nthreads=ora.max_threads(), want_idx=True)
    assert np.array_equal(nb, idx), "neighbour lists differ"
    # DESIGN.md 3.7 claims bit-identical normals (the device Jacobi is written operation for operation like the oracle's,
    # double +,-,*,/ and sqrt are correctly rounded on both sides): enforce exactly that
    same = np.array_equal(n_gpu.view(np.uint32), n_cpu.view(np.uint32))
    if not same:
        bad = np.flatnonzero((n_gpu.view(np.uint32) != n_cpu.view(np.uint32)).any(axis=1))
        raise AssertionError(f"{bad.size} of {len(xyz)} normals differ in bits from the oracle; first {bad[:5]}, "
                             f"max abs difference {np.abs(n_gpu - n_cpu).max()}")
    return same


@pytest.mark.parametrize("n", [1, 2, 19, 20, 21, 255, 257, 1025, 5000, 40000])
def test_normals_match_oracle(engine, n):
    xyz = st.synth_cloud_np(st.TARGET_SEED, n)
    check(engine, xyz, kd=n > 3000)


def test_other_k_and_viewpoints(engine):
    xyz = st.synth_cloud_np(st.TARGET_SEED + 3, 6000)
    for k, flip, vp in ((5, True, (0, 0, 10.0)), (8, False, (0, 0, 0)), (12, True, (3.0, -2.0, 1.0)), (32, True, (0, 0, 0))):
        check(engine, xyz, k, flip, vp)


def test_lattice_ties(engine):
    g = np.arange(8, dtype=np.float32)
    t = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    t = np.ascontiguousarray(t[np.random.default_rng(0).permutation(len(t))])
    engine.scene_load(t)
    _, nb = engine.scene_normals(7, False, (0, 0, 0), want_neighbors=True)
    idx, _ = ora.knn_self(t, 7, use_kdtree=False)
    assert np.array_equal(nb, idx)


def test_normals_feed_reference_mode_through_commit_target(engine):
    """camera frame -> filters -> normals -> commit -> reference-mode ICP, never leaving the device for the scene."""
    import pose_estimation_amd as pa
    T = st.perturbation(-1)
    scene = st.synth_cloud_np(st.TARGET_SEED, 8000)
    model = st.synth_cloud_np(st.SOURCE_SEED, 6000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    raw = np.concatenate([scene, np.full((300, 3), np.nan, np.float32)])
    raw = raw[np.random.default_rng(5).permutation(len(raw))]
    engine.scene_load(raw)
    engine.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
    clean = engine.scene_download()
    normals = engine.scene_normals(20, True, (0.0, 0.0, 0.0))
    engine.scene_commit_target()
    engine.set_sources([model])
    res = engine.align_reference(np.eye(4)[None], 100, 0.005, 2.5, 6)
    tgt6 = np.concatenate([clean, ora.normals(clean, 20, True, (0, 0, 0), nthreads=ora.max_threads())], axis=1)
    np.testing.assert_allclose(normals, tgt6[:, 3:], atol=3e-6)
    o = ora.icp_reference(model, np.concatenate([clean, normals], axis=1), np.eye(4), 100, 0.005, 2.5, 6)
    assert res.passes[0] == o["passes"] and np.linalg.norm(res.poses[0] - o["pose"]) <= 1e-5
    assert np.linalg.norm(res.poses[0] - np.linalg.inv(T)) < 5e-3


def test_normals_need_a_finite_scene(engine):
    import pose_estimation_amd as pa
    bad = st.synth_cloud_np(st.TARGET_SEED, 100)
    bad[7, 1] = np.nan
    engine.scene_load(bad)
    with pytest.raises(pa.IcpError) as e:
        engine.scene_normals(20)
    assert e.value.status == pa._lib.ERR_NONFINITE_INPUT


def test_cpp_harness_full_pipeline_on_device(engine):
    """The C++ host mirror drives raw frame -> NaN removal + plane band -> k=20 normals -> commit -> reference-mode
    refine, and lands on the known alignment."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "estimate_pose_harness")
    out = subprocess.run([exe, "4000", "6000", "3", "pipeline"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    kept = [l for l in lines if l.startswith("pipeline pass 1:")][0].split()   # "pipeline pass 1: 9000 raw points -> 5700 kept | ..."
    assert int(kept[3]) == 9000 and int(kept[7]) == 6000 - 300       # 3000 plane points and 300 NaN pixels removed
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]).reshape(4, 4) for l in lines if l.startswith("pose ")}
    gt = np.linalg.inv(st.perturbation(-1))
    for k in range(3):
        assert np.linalg.norm(poses[k] - gt) < 6e-3, k
    # A service's SECOND request must not meet first-use costs of the runtime any more (round 4: a 20 KB read-back that went to a
    # DMA engine cost 7-8 ms there, a request's whole time three times over: profiles/r04t_second_request_stall.txt).  Loose on
    # purpose - the three passes of this small frame are ~1 ms each and a box may hiccup; the stall was 7x.
    total = {int(l.split()[2].rstrip(":")): float(l.split("total ")[1].split()[0]) for l in lines if l.startswith("pipeline pass")}
    assert total[1] < 4.0 * total[2] + 1.0, total


def engine_with_knn_mode(mode):
    import os
    import pose_estimation_amd as pa
    old = os.environ.get("MI355_ICP_KNN_MODE")
    os.environ["MI355_ICP_KNN_MODE"] = mode
    try:
        return pa.IcpEngine(0)
    finally:
        if old is None:
            os.environ.pop("MI355_ICP_KNN_MODE")
        else:
            os.environ["MI355_ICP_KNN_MODE"] = old


def test_exhaustive_knn_scan_still_matches_the_oracle():
    """The default is the pruned scan on the Morton-sorted copy; MI355_ICP_KNN_MODE=brute keeps the exhaustive one alive."""
    with engine_with_knn_mode("brute") as eng:
        for n in (21, 1025, 5000):
            check(eng, st.synth_cloud_np(st.TARGET_SEED, n), kd=n > 3000)


def test_pruned_knn_on_geometry_that_stresses_the_tile_bounds(engine):
    """Distant tight clusters (huge tile spheres in Morton order), duplicated points (ties at distance 0 across tiles),
    a flat sheet far from the origin, and 300k points: neighbour lists must still be the oracle's."""
    rng = np.random.default_rng(11)
    centres = rng.standard_normal((7, 3)) * 50.0
    clusters = (centres[rng.integers(0, 7, 30000)] + rng.standard_normal((30000, 3)) * 0.02).astype(np.float32)
    check(engine, clusters)
    dup = st.synth_cloud_np(st.TARGET_SEED + 1, 4000)
    dup = np.concatenate([dup] * 6)[rng.permutation(24000)]              # every point six times: k = 20 ties everywhere
    engine.scene_load(dup)
    _, nb = engine.scene_normals(20, False, (0, 0, 0), want_neighbors=True)
    idx, _ = ora.knn_self(dup, 20, use_kdtree=True, nthreads=ora.max_threads())
    assert np.array_equal(nb, idx)
    sheet = (rng.uniform(-1, 1, (50000, 3)) * np.float32([1, 1, 1e-4]) + np.float32([1000.0, -500.0, 250.0])).astype(np.float32)
    check(engine, sheet)
    big = st.synth_cloud_np(st.TARGET_SEED + 2, 300_000)
    engine.scene_load(big)
    _, nb = engine.scene_normals(20, False, (0, 0, 0), want_neighbors=True)
    idx, _ = ora.knn_self(big, 20, use_kdtree=True, nthreads=ora.max_threads())
    assert np.array_equal(nb, idx)


def test_pruned_knn_when_every_sphere_is_reachable(engine):
    """135k copies of one point: every run, tile and super-tile ties with every query, so the k-NN kernel's LDS queues
    (192 runs, 128 tiles per wave) overflow and are drained many times per query.  The answer is known without an oracle:
    all distances are 0 and the (d2, index) order picks the k lowest indices for every query."""
    n = 135_000
    xyz = np.tile(np.float32([1.5, -2.0, 0.25]), (n, 1))
    engine.scene_load(xyz)
    _, nb = engine.scene_normals(20, False, (0, 0, 0), want_neighbors=True)
    assert nb.shape == (n, 20)
    assert np.array_equal(nb, np.broadcast_to(np.arange(20, dtype=nb.dtype), (n, 20)))


def test_a_service_of_requests_with_scenes_of_changing_size(engine):
    """Five requests in a row through one engine, the scene a little larger or smaller every time (what a camera delivers):
    device buffers come back from the engine's pool, the committed target takes over the sorted copy the k-NN built, and
    every request must still give the oracle's normals (bit for bit) and the oracle's reference-mode result.  The last scene
    is committed twice: the second commit has no sorted copy to inherit and builds its own."""
    T = st.perturbation(-1)
    model = st.synth_cloud_np(st.SOURCE_SEED, 3000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    engine.set_sources([model])
    sizes = (9000, 9500, 8100, 20000, 9001)
    for rep, n in enumerate(sizes):
        scene = st.synth_cloud_np(st.TARGET_SEED + rep, n)
        engine.scene_load(scene)
        normals = engine.scene_normals(20, True, (0.0, 0.0, 0.0))
        want = ora.normals(scene, 20, True, (0, 0, 0), nthreads=ora.max_threads())
        assert np.array_equal(normals.view(np.uint32), want.view(np.uint32)), rep
        for commit in range(2 if rep == len(sizes) - 1 else 1):
            engine.scene_commit_target()
            res = engine.align_reference(np.eye(4)[None], 60, 0.005, 2.5, 5)
            o = ora.icp_reference(model, np.concatenate([scene, want], axis=1), np.eye(4), 60, 0.005, 2.5, 5)
            assert res.passes[0] == o["passes"], (rep, commit)
            assert np.linalg.norm(res.poses[0] - o["pose"]) <= 1e-5, (rep, commit)

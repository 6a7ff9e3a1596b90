"""Scene-prep compaction on the GPU against the oracle: kept points identical and in the same order (bit-exact),
including NaN/Inf frames, empty results, ragged sizes, and multi-million-point frames (look-back across many tiles)."""
import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st
from test_scene_filters_cpu import frame

pytestmark = pytest.mark.gpu

FILTERS = [
    [(0, [])],
    [(0, []), (1, [0.1, -0.05, 0.7, 0.12])],
    [(0, []), (2, [0.1, -0.05, 0.7, 0.12])],
    [(0, []), (3, [0.0, 0.0, 1.0, -0.8, 0.005])],
    [(0, []), (2, [0.3, 0.3, 0.8, 0.1]), (3, [0.0, 0.0, 1.0, -0.8, 0.005]), (3, [1.0, 0.0, 0.0, 0.4, 0.005])],
    [],
]


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 1024, 1025, 20000, 300001])
def test_filter_matches_oracle(engine, n):
    pts = frame(max(n, 2), seed=n)[:n]
    for f in FILTERS:
        engine.scene_load(pts)
        kept = engine.scene_filter(f)
        got = engine.scene_download()
        want = ora.scene_filter(pts, f)
        assert kept == len(want) == len(got)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))   # NaN-safe bit comparison, same order


def test_boundary_points_exactly_on_the_sphere_and_band(engine):
    """sqrt and the division must round exactly like the host's: points placed on and next to the thresholds."""
    rng = np.random.default_rng(1)
    base = rng.standard_normal((50000, 3)).astype(np.float32)
    r = np.float32(1.25)
    on = (base / np.linalg.norm(base, axis=1, keepdims=True) * r).astype(np.float32)
    pts = np.concatenate([on, np.nextafter(on, np.float32(np.inf)), np.nextafter(on, np.float32(-np.inf))])
    for f in ([(1, [0, 0, 0, r])], [(2, [0, 0, 0, r])], [(3, [0.3, -0.2, 0.9, 0.0, 0.005])], [(3, [0.0, 0.0, 1.0, -1.0, 0.25])]):
        engine.scene_load(pts)
        engine.scene_filter(f)
        assert np.array_equal(engine.scene_download(), ora.scene_filter(pts, f))


def test_sequential_passes_and_all_removed(engine):
    pts = frame(100000, seed=9)
    engine.scene_load(pts)
    n1 = engine.scene_filter([(0, [])])
    n2 = engine.scene_filter([(3, [0.0, 0.0, 1.0, -0.8, 0.005])])       # second pass works on the compacted scene
    want = ora.scene_filter(ora.scene_filter(pts, [(0, [])]), [(3, [0.0, 0.0, 1.0, -0.8, 0.005])])
    assert n1 > n2 == len(want) and np.array_equal(engine.scene_download(), want)
    assert engine.scene_filter([(1, [100.0, 100.0, 100.0, 0.001])]) == 0  # nothing inside that sphere
    assert engine.scene_download().shape == (0, 3)
    assert engine.scene_filter([(0, [])]) == 0                             # filtering an empty scene is fine


def test_large_frame_many_tiles(engine):
    """8M points = 8192 tiles: the decoupled look-back crosses many windows and all 8 XCDs."""
    n = 1 << 23
    rng = np.random.default_rng(2)
    pts = rng.standard_normal((n, 3)).astype(np.float32)
    pts[rng.random(n) < 0.3, 1] = np.nan
    for f in ([(0, [])], [(0, []), (2, [0, 0, 0, 1.0])], [(0, []), (1, [0, 0, 0, 0.2])]):
        engine.scene_load(pts)
        kept = engine.scene_filter(f)
        want = ora.scene_filter(pts, f)
        assert kept == len(want)
        assert np.array_equal(engine.scene_download(), want)


def test_commit_target_feeds_the_icp_without_leaving_the_device(engine):
    import pose_estimation_amd as pa
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 4000, 6000)
    rng = np.random.default_rng(4)
    raw = np.concatenate([tgt, np.full((500, 3), np.nan, np.float32), (rng.standard_normal((800, 3)) * 0.01 + [5, 5, 5]).astype(np.float32)])
    raw = raw[rng.permutation(len(raw))]
    engine.scene_load(raw)
    with pytest.raises(pa.IcpError) as e:
        engine.scene_commit_target()                                     # still holds NaN
    assert e.value.status == pa._lib.ERR_NONFINITE_INPUT
    f = [(0, []), (2, [5.0, 5.0, 5.0, 0.5])]
    engine.scene_filter(f)
    engine.scene_commit_target()
    engine.set_sources([src])
    scene = ora.scene_filter(raw, f)
    idx, d2, sums = engine.correspond(gt[None])
    oi, od, os_ = ora.correspond(gt, src, scene, use_kdtree=False)
    assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))

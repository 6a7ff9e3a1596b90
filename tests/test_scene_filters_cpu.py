"""The scene-prep filter oracle against plain numpy restatements of the reference's lines
(pose_estimation/src/pose_estimation.cpp:248, :310-333, :347-372)."""
import numpy as np

import oracle_lib as ora


def frame(n=20000, seed=0):
    """A camera-like frame: points on a table plane + an object blob + NaN dropouts."""
    rng = np.random.default_rng(seed)
    plane = np.c_[rng.uniform(-0.5, 0.5, (n // 2, 2)), 0.8 + 0.001 * rng.standard_normal(n // 2)]
    blob = rng.standard_normal((n - n // 2, 3)) * 0.05 + [0.1, -0.05, 0.7]
    pts = np.concatenate([plane, blob]).astype(np.float32)
    pts = pts[rng.permutation(n)]
    bad = rng.random(n) < 0.1
    pts[bad, rng.integers(0, 3, bad.sum())] = np.nan
    pts[rng.random(n) < 0.01] = np.inf
    return pts


def test_nan_removal_keeps_order():
    pts = frame()
    out = ora.scene_filter(pts, [(0, [])])
    want = pts[np.isfinite(pts).all(1)]
    assert np.array_equal(out, want) and 0 < len(out) < len(pts)


def test_sphere_filter_matches_reference_arithmetic():
    pts = frame()
    pts = pts[np.isfinite(pts).all(1)]
    c, r = np.float32([0.1, -0.05, 0.7]), np.float32(0.12)
    d = c - pts                                     # float dx = point[0] - cloud->points[i].x  (:353-355)
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])   # float32 throughout (:357)
    inside = dist <= r
    assert np.array_equal(ora.scene_filter(pts, [(1, [*c, r])]), pts[inside])      # ExtractIndices, negative = false
    assert np.array_equal(ora.scene_filter(pts, [(2, [*c, r])]), pts[~inside])     # filter_out == "inliers": negative = true
    assert 0 < inside.sum() < len(pts)


def test_plane_band_divides_by_point_norm_like_the_reference():
    pts = frame()
    pts = pts[np.isfinite(pts).all(1)]
    a, b, c, dd = np.float32([0.0, 0.0, 1.0, -0.8])
    num = ((pts[:, 0] * a + pts[:, 1] * b) + pts[:, 2] * c) + dd                   # :313-316
    den = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2])   # :317-319, |p| not |(a,b,c)|
    d = num / den
    near = np.abs(d).astype(np.float64) <= np.float64(np.float32(0.005))           # std::abs(d) <= 0.005 (:320)
    out = ora.scene_filter(pts, [(3, [a, b, c, dd, 0.005])])
    assert np.array_equal(out, pts[~near]) and 0 < near.sum() < len(pts)
    # a point at the origin divides by zero -> inf/nan -> not "near" -> kept, as in the reference
    z = np.zeros((1, 3), np.float32)
    assert len(ora.scene_filter(z, [(3, [a, b, c, dd, 0.005])])) == 1


def test_filters_compose_in_one_pass():
    pts = frame(seed=3)
    f = [(0, []), (2, [0.3, 0.3, 0.8, 0.1]), (3, [0, 0, 1, -0.8, 0.005])]
    step = pts
    for one in f:
        step = ora.scene_filter(step, [one])
    assert np.array_equal(ora.scene_filter(pts, f), step)
    assert len(ora.scene_filter(np.zeros((0, 3), np.float32), f)) == 0

"""The oracle's RANSAC plane fit (oracle/icp_oracle.c, the checker of mi355_icp_scene_fit_plane; restates what the reference
asks pcl::SACSegmentation for at pose_estimation.cpp:285-297 - parity unpinned, PCL is absent): it must recover a known
plane from a cluttered scene, spend fewer iterations than the cap when the plane dominates, be deterministic in its seed,
ignore NaN pixels, and report failure on a scene no three points of which span a plane."""
import numpy as np

import oracle_lib as ora


def table_scene(n_plane=60000, n_clutter=30000, noise=5e-5, seed=1):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-0.5, 0.5, (n_plane, 2))
    plane = np.c_[xy, 0.3 * xy[:, 0] + 0.1 * xy[:, 1] + 1.0 + rng.normal(0, noise, n_plane)]
    clutter = rng.normal([0, 0, 0.9], 0.05, (n_clutter, 3))
    pts = np.concatenate([plane, clutter]).astype(np.float32)
    return pts[rng.permutation(len(pts))], np.array([0.3, 0.1, -1.0]) / np.linalg.norm([0.3, 0.1, -1.0]), 1.0 / np.linalg.norm([0.3, 0.1, -1.0])


def test_recovers_the_table_plane_and_stops_early():
    pts, n_true, d_true = table_scene()
    st, c, inliers, it, planes, counts = ora.plane_fit(pts)
    assert st == 0 and abs(np.linalg.norm(c[:3]) - 1.0) < 1e-6
    s = np.sign(c[:3] @ n_true)
    assert np.arccos(min(1.0, abs(c[:3] @ n_true))) < 2e-4 and abs(s * c[3] - d_true) < 2e-4
    assert inliers > 0.9 * 60000 * 0.95 and it < 100                      # threshold 0.0001 = 2 sigma of the noise; early stop
    assert counts.max() >= 3 and (counts[~np.isnan(planes[:, 3])] >= 3).all()   # a hypothesis supports at least its own triple
    again = ora.plane_fit(pts)
    assert np.array_equal(again[1], c) and again[2] == inliers and again[3] == it
    other = ora.plane_fit(pts, seed=7)
    assert np.arccos(min(1.0, abs(other[1][:3] @ n_true))) < 2e-4              # another sample sequence, the same plane


def test_nan_pixels_and_degenerate_scenes():
    pts, n_true, _ = table_scene(20000, 8000)
    pts[::13] = np.nan
    st, c, inliers, it, _, _ = ora.plane_fit(pts)
    assert st == 0 and np.arccos(min(1.0, abs(c[:3] @ n_true))) < 3e-4
    line = np.outer(np.arange(500.0), [1.0, 2.0, 3.0]).astype(np.float32)   # integers: exactly collinear in float32      # all points collinear: no plane
    assert ora.plane_fit(line)[0] == 1
    no_opt = ora.plane_fit(pts, optimize=False)
    assert no_opt[0] == 0 and no_opt[2] <= inliers          # a raw hypothesis never has more support than its refinement here

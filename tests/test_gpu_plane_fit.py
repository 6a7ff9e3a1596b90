"""mi355_icp_scene_fit_plane - the pcl::SACSegmentation call of PoseEstimation::remove_planes (reference
pose_estimation.cpp:285-297) as three small kernels - against the oracle's restatement: same hypotheses (hence the same
iteration count), the same integer support, coefficients equal to float32 rounding; then the reference's own chain
(NaN removal -> fit -> distance band around the FITTED plane, :248, :285-333) against the oracle's chain, point for point."""
import numpy as np
import pytest

import oracle_lib as ora
from test_plane_fit_cpu import table_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_plane,n_clutter,noise", [(60000, 30000, 5e-5), (5000, 20000, 2e-5), (300, 40, 0.0), (700000, 300000, 5e-5)])
def test_fit_matches_the_oracle(engine, n_plane, n_clutter, noise):
    pts, n_true, d_true = table_scene(n_plane, n_clutter, noise, seed=n_plane)
    engine.scene_load(pts)
    for seed, optimize in ((11, False), (11, True), (0x5EEDC0DE, True)):
        c, inliers, it = engine.scene_fit_plane(seed=seed, optimize=optimize)
        st, oc, oin, oit, planes, counts = ora.plane_fit(pts, seed=seed, optimize=optimize)
        assert st == 0 and it == oit, (seed, optimize, it, oit)
        if optimize:       # moments are summed in another order: the refined plane agrees to float32 rounding
            np.testing.assert_allclose(c, oc, rtol=0, atol=2e-7)
            assert abs(int(inliers) - int(oin)) <= max(2, oin // 100000)   # a point exactly on the threshold may flip with the last bit
        else:              # a raw hypothesis: float32 arithmetic and integer counts, bit for bit
            assert np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and inliers == oin
    if n_plane > n_clutter:      # the table dominates: the default fit must land on it (a minority plane needs more than 100 draws)
        assert np.arccos(min(1.0, abs(c[:3] @ n_true))) < 1e-3


def test_reference_chain_with_the_fitted_plane(engine):
    """create_surface_match_pc's chain with num_planes = 1: a camera-like frame with NaN pixels, NaN removal, RANSAC plane,
    distance band 0.005 around the fitted plane - the kept points equal the oracle's chain."""
    from pose_estimation_amd import _lib
    pts, n_true, _ = table_scene(80000, 40000, 5e-5, seed=5)
    pts[::17] = np.nan
    engine.scene_load(pts)
    engine.scene_filter([(_lib.SCENE_FILTER_FINITE, [])])
    c, inliers, it = engine.scene_fit_plane(optimize=False)          # raw hypothesis: bit-identical to the oracle's, so the band is too
    clean = ora.scene_filter(pts, [(0, [])])
    st, oc, oin, oit, _, _ = ora.plane_fit(clean, optimize=False)
    assert st == 0 and np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and inliers == oin and it == oit
    kept = engine.scene_filter([(_lib.SCENE_FILTER_PLANE_BAND_REMOVE, list(c) + [0.005])])
    want = ora.scene_filter(clean, [(3, list(oc) + [0.005])])
    assert kept == len(want) and np.array_equal(engine.scene_download().view(np.uint32), want.view(np.uint32))
    assert kept < 40000 * 1.02          # the table is gone, the clutter stays


def test_fit_argument_checks(engine):
    import pose_estimation_amd as pa
    engine.scene_load(np.zeros((2, 3), dtype=np.float32))
    with pytest.raises(pa.IcpError):
        engine.scene_fit_plane()
    line = np.outer(np.arange(500.0), [1.0, 2.0, 3.0]).astype(np.float32)   # integers: exactly collinear in float32
    engine.scene_load(line)
    with pytest.raises(pa.IcpError) as e:
        engine.scene_fit_plane()
    assert e.value.status == pa._lib.ERR_NUMERIC
    engine.scene_load(table_scene(3000, 100)[0])
    with pytest.raises(pa.IcpError):
        engine.scene_fit_plane(max_iterations=5000)

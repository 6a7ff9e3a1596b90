"""Scene-normals oracle (k-NN PCA, computeNormalsPC3d semantics) against scipy / numpy and the analytic normals of
the synthetic surface.  PARITY UNPINNED (upstream source absent; oracle/icp_oracle.c "Scene normals")."""
import numpy as np
from scipy.spatial import cKDTree

import oracle_lib as ora
import synth_twin as st


def test_knn_kdtree_equals_brute_and_scipy():
    t = st.synth_cloud_np(st.TARGET_SEED, 6000)
    ib, db = ora.knn_self(t, 20, use_kdtree=False)
    ik, dk = ora.knn_self(t, 20, use_kdtree=True, nthreads=2)
    assert np.array_equal(ib, ik) and np.array_equal(db.view(np.uint32), dk.view(np.uint32))
    assert np.array_equal(ib[:, 0], np.arange(len(t))) and np.all(db[:, 0] == 0)     # the point itself comes first
    assert np.all(np.diff(db, axis=1) >= 0)
    dd, ii = cKDTree(t.astype(np.float64)).query(t.astype(np.float64), k=20)
    assert (ii == ib).mean() > 0.9999                                                  # float32 ties aside


def test_ties_resolve_by_index_on_a_lattice():
    g = np.arange(6, dtype=np.float32)
    t = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    t = t[np.random.default_rng(0).permutation(len(t))]
    ib, db = ora.knn_self(t, 7, use_kdtree=False)
    ik, dk = ora.knn_self(t, 7, use_kdtree=True)
    assert np.array_equal(ib, ik)
    D = ((t[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(-1)
    want = np.lexsort((np.broadcast_to(np.arange(len(t)), D.shape), D), axis=1)[:, :7]  # (d2, index) order
    assert np.array_equal(ib, want.astype(np.int32))


def test_normals_match_numpy_eigh_and_the_analytic_surface_normal():
    t6 = st.synth_cloud_np(st.TARGET_SEED, 40000, normals=True)
    xyz = np.ascontiguousarray(t6[:, :3])
    n, idx = ora.normals(xyz, 20, True, (0, 0, 0), nthreads=4, want_idx=True)
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-6
    # flipped towards the viewpoint at the origin = against the outward analytic normal
    cosv = (n * t6[:, 3:]).sum(1)
    assert cosv.max() < 0 and np.abs(cosv).min() > 0.998
    # independent PCA on the same neighbour sets
    for q in range(0, 40000, 997):
        c = xyz[idx[q]].astype(np.float64)
        w, v = np.linalg.eigh(np.cov(c.T, bias=True))
        assert abs(abs(v[:, 0] @ n[q]) - 1) < 1e-5
    n2 = ora.normals(xyz, 20, False, (0, 0, 0), nthreads=4)
    assert np.allclose(np.abs((n * n2).sum(1)), 1, atol=1e-6)                         # flip only changes the sign
    far = ora.normals(xyz, 20, True, (0, 0, 100.0), nthreads=4)
    assert ((far * (np.float32([0, 0, 100.0]) - xyz)).sum(1) >= 0).all()


def test_small_clouds_clamp_k():
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.001]], dtype=np.float32)
    n = ora.normals(pts, 20, True, (0, 0, 5))
    assert np.abs(np.abs(n[:, 2]) - 1).max() < 1e-3 and (n[:, 2] > 0).all()
    one = ora.normals(pts[:1], 20, False, (0, 0, 0))
    assert np.isfinite(one).all()

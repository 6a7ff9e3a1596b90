"""Reference mode (pyramid + MAD rejection + picky + point-to-plane): the oracle restatement against numpy,
and the product's host solver against both.  PARITY UNPINNED: upstream's source is not available offline
(oracle/icp_oracle.c, second half, states what is restated and what deviates)."""
import numpy as np

import oracle_lib as ora
import pose_estimation_amd as pa
import synth_twin as st


def pair6(ns, nt, k=-1):
    T = st.perturbation(k)
    off = 0 if k < 0 else 16 * k
    tgt = st.synth_cloud_np(st.TARGET_SEED + off, nt, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED + off, ns, noise_seed=st.NOISE_SEED + off, sigma=st.SIGMA, transform=T)
    return src, tgt, np.linalg.inv(T)


def test_rejection_threshold_is_median_plus_scaled_mad():
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 1000, 1001):
        d2 = (rng.standard_normal(n) ** 2).astype(np.float32)
        med = np.sort(d2)[(n - 1) // 2]
        dev = np.abs(d2.astype(np.float64) - np.float64(med)).astype(np.float32)
        mad = np.sort(dev)[(n - 1) // 2]
        want = np.float32(2.5) * (np.float32(1.48257968) * mad) + med
        assert ora.rejection_threshold(d2, 2.5) == want


def test_pass_selection_is_one_to_one_and_matches_numpy():
    src, tgt, gt = pair6(3000, 2000)
    P0 = st.rigid_transform(1.0)
    n, sums, idx, d2, sel, thr = ora.ref_pass(src, tgt, P0, P0, 2.5)
    assert n == sel.sum() == sums[28] and 6 < n < 3000
    assert thr == ora.rejection_threshold(d2, 2.5)
    assert np.all(d2[sel] < thr)
    assert len(np.unique(idx[sel])) == n                      # picky: one source per target
    for j in np.unique(idx[sel])[:200]:                       # ... the closest kept one, lowest index on ties
        cand = np.nonzero((idx == j) & (d2 < thr))[0]
        best = cand[np.lexsort((cand, d2[cand]))[0]]
        assert sel[best] and sel[cand].sum() == 1
    # normal equations against an explicit A, b
    p = ora.transform(P0, src)[sel].astype(np.float64)
    q = tgt[idx[sel], :3].astype(np.float64)
    nn = tgt[idx[sel], 3:].astype(np.float64)
    A = np.concatenate([np.cross(p, nn), nn], axis=1)
    b = ((q - p) * nn).sum(1)
    N = A.T @ A
    np.testing.assert_allclose(sums[:21], N[np.triu_indices(6)], rtol=1e-12)
    np.testing.assert_allclose(sums[21:27], A.T @ b, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sums[27], ((p - q) ** 2).sum(), rtol=1e-12)
    x_np = np.linalg.lstsq(A, b, rcond=None)[0]               # what cv::solve(..., DECOMP_SVD) computes
    st_o, x_o = ora.solve_point_to_plane(sums)
    assert st_o == 0
    np.testing.assert_allclose(x_o, x_np, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(pa.solve_point_to_plane(sums), x_np, rtol=1e-8, atol=1e-12)


def test_rank_deficient_normal_equations_give_the_minimum_norm_solution():
    rng = np.random.default_rng(3)
    p = np.c_[rng.uniform(-1, 1, (50, 2)), np.zeros(50)]      # a plane with one normal: only 3 of 6 dof observable
    nn = np.tile([0.0, 0.0, 1.0], (50, 1))
    q = p + [0, 0, 0.01]
    A = np.concatenate([np.cross(p, nn), nn], axis=1)
    b = ((q - p) * nn).sum(1)
    N = A.T @ A
    sums = np.concatenate([N[np.triu_indices(6)], A.T @ b, [((p - q) ** 2).sum(), 50.0]])
    want = np.linalg.lstsq(A, b, rcond=None)[0]
    np.testing.assert_allclose(ora.solve_point_to_plane(sums)[1], want, atol=1e-12)
    np.testing.assert_allclose(pa.solve_point_to_plane(sums), want, atol=1e-12)


def test_euler_convention():
    x = np.array([0.3, -0.2, 0.5, 1.0, 2.0, 3.0])
    P = ora.euler_pose(x)
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(x[0]), -s(x[0])], [0, s(x[0]), c(x[0])]])
    Ry = np.array([[c(x[1]), 0, s(x[1])], [0, 1, 0], [-s(x[1]), 0, c(x[1])]])
    Rz = np.array([[c(x[2]), -s(x[2]), 0], [s(x[2]), c(x[2]), 0], [0, 0, 1]])
    np.testing.assert_allclose(P[:3, :3], Rx @ (Ry @ Rz), atol=1e-15)
    np.testing.assert_allclose(P[:3, 3], x[3:])


def test_reference_mode_recovers_the_known_perturbation():
    """With the reference's own parameters (250, 0.005, 2.5, 8) the pyramid + point-to-plane loop lands two orders
    of magnitude closer to the truth than 50 point-to-point iterations on the same clouds."""
    src, tgt, gt = pair6(20000, 20000)
    r = ora.icp_reference(src, tgt, np.eye(4), 250, 0.005, 2.5, 8, nthreads=4)
    assert r["status"] == 0 and 8 <= r["passes"] <= 60
    assert np.linalg.norm(r["pose"] - gt) < 2e-3
    assert abs(np.linalg.det(r["pose"][:3, :3]) - 1) < 1e-9
    p2p = ora.icp(src, tgt[:, :3], np.eye(4), ora.params(max_iterations=50), nthreads=4)
    assert np.linalg.norm(r["pose"] - gt) < 0.1 * np.linalg.norm(p2p["pose"] - gt)


def test_tiny_clouds_skip_levels_instead_of_failing():
    src, tgt, gt = pair6(40, 60)
    r = ora.icp_reference(src, tgt, np.eye(4), 50, 0.005, 2.5, 8)
    assert r["status"] == 0 and np.isfinite(r["pose"]).all()   # levels with < 6 source points are skipped


def test_golden_reference_mode_fixture_reproduced():
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "refmode_2k.npz"))
    meta = json.load(open(os.path.join(gold, "golden.json")))["refmode_2k"]
    it, tol, scale, levels = meta["params"]
    r = ora.icp_reference(g["src"], g["tgt6"], np.eye(4), it, tol, scale, levels)
    assert r["passes"] == meta["passes"] and r["residual"] == meta["residual"]
    assert np.array_equal(r["pose"], g["final_pose"])
    n, sums, idx, d2, sel, thr = ora.ref_pass(g["src"][::2][:1000], g["tgt6"][::2][:1250], g["start"], g["cur"], scale)
    assert n == meta["pass_selected"] and np.array_equal(idx, g["pass_idx"]) and np.array_equal(d2, g["pass_d2"])
    assert np.array_equal(sel, g["pass_sel"]) and np.float32(thr) == g["pass_thr"] and np.array_equal(sums, g["pass_sums"])


def _random_normal_equations(rng, n_rows, rank_deficient=False, scale=1.0):
    p = rng.uniform(-1, 1, (n_rows, 3)) * scale
    nn = rng.standard_normal((n_rows, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    if rank_deficient:
        p[:, 2] = 0.0
        nn[:] = [0.0, 0.0, 1.0]
    q = p + rng.standard_normal((n_rows, 3)) * 1e-2 * scale
    A = np.concatenate([np.cross(p, nn), nn], axis=1)
    b = ((q - p) * nn).sum(1)
    N = A.T @ A
    return np.concatenate([N[np.triu_indices(6)], A.T @ b, [((p - q) ** 2).sum(), float(n_rows)]])


def test_batched_solve_gives_every_system_the_bits_of_the_single_solve():
    """The SIMD batch (8 systems per pass; what a reference-mode pass runs for its candidates) against the scalar solve, bit for
    bit: well-conditioned, rank-deficient (masked eigen-directions), badly scaled, all-zero and non-finite systems, in batches
    whose lanes converge after different numbers of sweeps, and batch sizes around the SIMD width."""
    rng = np.random.default_rng(20261004)
    rows = []
    for k in range(45):
        kind = k % 5
        if kind == 0:
            rows.append(_random_normal_equations(rng, 200))
        elif kind == 1:
            rows.append(_random_normal_equations(rng, 50, rank_deficient=True))
        elif kind == 2:
            rows.append(_random_normal_equations(rng, 30, scale=10.0 ** rng.integers(-6, 7)))
        elif kind == 3:
            rows.append(np.zeros(29))
        else:
            bad = _random_normal_equations(rng, 100)
            bad[rng.integers(0, 29)] = [np.nan, np.inf, -np.inf][k % 3]
            rows.append(bad)
    sums = np.stack(rows)
    for n in (1, 2, 6, 7, 8, 9, 16, 17, 45):
        x, status = pa.solve_point_to_plane_batch(sums[:n])
        for k in range(n):
            try:
                want = pa.solve_point_to_plane(sums[k])
                assert status[k] == 0, (n, k)
                assert np.array_equal(x[k].view(np.uint64), want.view(np.uint64)), (n, k, x[k], want)
            except pa.IcpError:
                assert status[k] != 0, (n, k)
    x, status = pa.solve_point_to_plane_batch(np.zeros((0, 29)))
    assert x.shape == (0, 6) and status.shape == (0,)

"""Pins the oracle (oracle/icp_oracle.c): against the committed golden vectors, and against
implementations that share no code with it (scipy cKDTree, numpy SVD, float64 brute force).
PARITY UNPINNED by the reference: it holds no tests or vectors for this path."""
import hashlib
import json
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle_lib as ora
import synth_twin as st

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kabsch_numpy(sums):
    n = sums[0]
    sb, tb = sums[1:4] / n, sums[4:7] / n
    H = sums[7:16].reshape(3, 3) - np.outer(sums[1:4], sums[4:7]) / n
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = tb - R @ sb
    return P


def test_golden_c1_reproduced():
    g = np.load(os.path.join(GOLD, "c1_1k.npz"))
    cl = np.load(os.path.join(GOLD, "c1_1k_clouds.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden.json")))["c1_1k"]
    src, tgt = cl["src"], cl["tgt"]
    assert sha(src) == meta["src_sha"] and sha(tgt) == meta["tgt_sha"]
    for kd in (True, False):
        r = ora.icp(src, tgt, np.eye(4), ora.params(max_iterations=50), use_kdtree=kd, trace=True)
        assert r["status"] == 0 and r["iterations"] == meta["iterations"]
        assert np.array_equal(r["pose"], g["final_pose"])
        assert np.array_equal(r["trace_sums"], g["trace_sums"])
    idx, d2, sums = ora.correspond(np.eye(4), src, tgt, use_kdtree=False)
    assert np.array_equal(idx, g["idx0"]) and np.array_equal(d2.view(np.uint32), g["d2_0"].view(np.uint32))


def test_golden_c2_100k_first_pass_and_hashes():
    g = np.load(os.path.join(GOLD, "c2_100k.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden.json")))["c2_100k"]
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 100_000, 100_000)
    assert sha(src) == meta["src_sha"] and sha(tgt) == meta["tgt_sha"]
    for it in (0, 49):
        idx, d2, sums = ora.correspond(g["trace_pose"][it], src, tgt, nthreads=ora.max_threads())
        assert sha(idx) == meta["idx_sha"][str(it)]["idx"] and sha(d2) == meta["idx_sha"][str(it)]["d2"]
        assert np.array_equal(sums, g["trace_sums"][it])
    assert np.linalg.norm(g["final_pose"] - gt) < 0.6 * np.linalg.norm(np.eye(4) - gt)


@pytest.mark.parametrize("ns,nt,seed", [(500, 700, 0), (4000, 3000, 1), (1, 50, 2), (300, 1, 3), (2000, 13, 4)])
def test_kdtree_equals_brute_bitwise(ns, nt, seed):
    rng = np.random.default_rng(seed)
    t = rng.standard_normal((nt, 3)).astype(np.float32)
    q = rng.standard_normal((ns, 3)).astype(np.float32)
    ib, db = ora.nn_brute(q, t)
    ik, dk = ora.KdTree(t).query(q, nthreads=2)
    assert np.array_equal(ib, ik) and np.array_equal(db.view(np.uint32), dk.view(np.uint32))


def test_ties_fixture_lowest_index_rule():
    g = np.load(os.path.join(GOLD, "ties.npz"))
    q, tgt = g["q"], g["tgt"]
    D = ((q[:, None, :].astype(np.float64) - tgt[None, :, :].astype(np.float64)) ** 2).sum(-1)   # exact here
    want = D.argmin(1).astype(np.int32)  # numpy argmin = first (lowest) index of the minimum
    assert ((D == D.min(1, keepdims=True)).sum(1) > 1).sum() > 500       # the fixture really is tie-heavy
    for idx, d2 in (ora.nn_brute(q, tgt), ora.KdTree(tgt).query(q)):
        assert np.array_equal(idx, want) and np.array_equal(idx, g["idx"]) and np.array_equal(d2, g["d2"])


def test_duplicate_targets_and_coincident_queries():
    t = np.repeat(np.random.default_rng(0).standard_normal((50, 3)).astype(np.float32), 7, axis=0)
    perm = np.random.default_rng(1).permutation(t.shape[0])
    t = t[perm]
    ib, db = ora.nn_brute(t, t)
    ik, dk = ora.KdTree(t).query(t)
    assert np.array_equal(ib, ik) and np.all(db == 0)
    assert np.all(ib <= np.arange(t.shape[0]))  # the lowest-index copy wins


def test_nn_against_scipy_ckdtree():
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 20000, 15000)
    moved = ora.transform(st.rigid_transform(2.0), src)
    idx, d2 = ora.KdTree(tgt).query(moved, nthreads=ora.max_threads())
    dd, ii = cKDTree(tgt.astype(np.float64)).query(moved.astype(np.float64))
    bad = np.nonzero(ii != idx)[0]
    for i in bad:  # only float32 ties may differ from the float64 answer
        assert ora.nn_brute(moved[i:i + 1], tgt[ii[i]:ii[i] + 1])[1][0] == d2[i]
    assert len(bad) < 5
    np.testing.assert_allclose(np.sqrt(d2.astype(np.float64)), dd, rtol=2e-6, atol=1e-9)


def test_distance_and_transform_rounding_order():
    """d2 = ((dx*dx)+(dy*dy))+(dz*dz) and s' = ((r0*x+r1*y)+r2*z)+t with float32 rounding after every op."""
    rng = np.random.default_rng(9)
    q = rng.standard_normal((2000, 3)).astype(np.float32)
    t = rng.standard_normal((1, 3)).astype(np.float32)
    _, d2 = ora.nn_brute(q, t)
    d = q - t[0]
    want = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]   # numpy float32 ops round each step
    assert np.array_equal(d2.view(np.uint32), want.view(np.uint32))
    P = st.rigid_transform(17.0, (0.2, 0.9, -0.4), (0.3, -0.7, 1.1))
    m = ora.pose_to_f32(P).reshape(3, 4)
    got = ora.transform(P, q)
    for r in range(3):
        want = ((m[r, 0] * q[:, 0] + m[r, 1] * q[:, 1]) + m[r, 2] * q[:, 2]) + m[r, 3]
        assert np.array_equal(got[:, r].view(np.uint32), want.view(np.uint32))


def test_accumulate_and_rejection():
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 3000, 2000)
    idx, d2, sums = ora.correspond(np.eye(4), src, tgt, use_kdtree=False)
    s, t = src.astype(np.float64), tgt[idx].astype(np.float64)
    want = np.concatenate([[len(s)], s.sum(0), t.sum(0), (s[:, :, None] * t[:, None, :]).sum(0).ravel(), [d2.astype(np.float64).sum()]])
    np.testing.assert_allclose(sums, want, rtol=1e-12, atol=1e-12)
    thr = np.float32(np.sqrt(np.median(d2)))
    sums_r = ora.accumulate(src, tgt, idx, d2, thr)
    keep = d2 <= thr * thr
    assert sums_r[0] == keep.sum() and 0 < keep.sum() < len(s)
    np.testing.assert_allclose(sums_r[1:4], s[keep].sum(0), rtol=1e-12)


def test_kabsch_against_numpy_svd():
    rng = np.random.default_rng(4)
    for _ in range(50):
        s = rng.standard_normal((40, 3)).astype(np.float32)
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        if np.linalg.det(R) < 0:
            R[:, 0] = -R[:, 0]
        d = (s.astype(np.float64) @ R.T + rng.standard_normal(3) + 0.01 * rng.standard_normal((40, 3))).astype(np.float32)
        sums = ora.accumulate(s, d, np.arange(40, dtype=np.int32), np.zeros(40, np.float32))
        stt, P = ora.solve_rigid(sums)
        assert stt == 0
        np.testing.assert_allclose(P, kabsch_numpy(sums), atol=1e-10)
        assert abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12


def test_degenerate_fixture():
    g = np.load(os.path.join(GOLD, "degenerate.npz"))
    for nm in ("coplanar", "collinear", "mirror"):
        stt, P = ora.solve_rigid(g[nm + "_sums"])
        assert stt == 0 and abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12
        if nm != "collinear":       # the collinear optimum is a one-parameter family: compare fit, not pose
            np.testing.assert_allclose(P, g[nm + "_pose"], atol=1e-12)
            np.testing.assert_allclose(P, kabsch_numpy(g[nm + "_sums"]), atol=1e-9)
        else:
            s, d = g["collinear_src"].astype(np.float64), g["collinear_dst"].astype(np.float64)
            assert np.abs(s @ P[:3, :3].T + P[:3, 3] - d).max() < 1e-5
    assert ora.solve_rigid(np.r_[2.0, np.zeros(16)])[0] == 5      # < 3 pairs
    assert ora.solve_rigid(np.r_[10.0, np.nan, np.zeros(15)])[0] == 6


def test_identity_alignment_converges_immediately():
    tgt = st.synth_cloud_np(st.TARGET_SEED, 2000)
    p = ora.params(max_iterations=30, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    r = ora.icp(tgt, tgt, np.eye(4), p)
    assert r["status"] == 0 and r["iterations"] == 1 and r["converged"] == 1 and r["residual"] == 0.0
    np.testing.assert_allclose(r["pose"], np.eye(4), atol=1e-12)


def test_pose7_fixture():
    g = np.load(os.path.join(GOLD, "pose7.npz"))
    for P, a, b in zip(g["poses"], g["pose7"], g["pose7_ref"]):
        assert np.array_equal(ora.pose_to_pose7(P, False), a)
        assert np.array_equal(ora.pose_to_pose7(P, True), b)


def test_committed_fixtures_are_what_the_generator_writes():
    """tests/golden/make_golden.py --check regenerates every fixture into a scratch directory (cross-checking each against
    scipy / numpy / the literal upstream restatement on the way) and fails if an array or golden.json differs from the
    committed files: a fixture can neither drift from the oracle nor be edited by hand unnoticed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

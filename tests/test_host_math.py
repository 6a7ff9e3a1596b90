"""The product's host arithmetic (csrc/host_math.cpp, via the C-ABI) against numpy/scipy and the oracle."""
import os

import numpy as np
from scipy.spatial.transform import Rotation

import oracle_lib as ora
import pose_estimation_amd as pa
import synth_twin as st
from test_oracle import kabsch_numpy

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def random_sums(rng, n=60, noise=0.02):
    s = rng.standard_normal((n, 3)).astype(np.float32)
    R = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
    d = (s.astype(np.float64) @ R.T + rng.standard_normal(3) + noise * rng.standard_normal((n, 3))).astype(np.float32)
    d2 = ((s - d) ** 2).sum(1).astype(np.float32)
    return ora.accumulate(s, d, np.arange(n, dtype=np.int32), d2)


def test_solve_rigid_matches_numpy_svd_and_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        sums = random_sums(rng)
        P = pa.solve_rigid(sums)
        np.testing.assert_allclose(P, kabsch_numpy(sums), atol=1e-10)
        np.testing.assert_allclose(P, ora.solve_rigid(sums)[1], atol=1e-10)
        assert abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12 and np.array_equal(P[3], [0, 0, 0, 1])


def test_solve_rigid_degenerate_and_errors():
    g = np.load(os.path.join(GOLD, "degenerate.npz"))
    for nm in ("coplanar", "mirror"):
        np.testing.assert_allclose(pa.solve_rigid(g[nm + "_sums"]), g[nm + "_pose"], atol=1e-10)
    P = pa.solve_rigid(g["collinear_sums"])
    s, d = g["collinear_src"].astype(np.float64), g["collinear_dst"].astype(np.float64)
    assert abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12 and np.abs(s @ P[:3, :3].T + P[:3, 3] - d).max() < 1e-5
    for bad, code in ((np.r_[2.0, np.zeros(16)], pa._lib.ERR_TOO_FEW_PAIRS), (np.r_[9.0, np.inf, np.zeros(15)], pa._lib.ERR_NUMERIC)):
        try:
            pa.solve_rigid(bad)
            raise AssertionError("expected failure")
        except pa.IcpError as e:
            assert e.status == code
    # all points identical (rank 0): still a proper rotation, translation maps centroid to centroid
    sums = np.zeros(17)
    sums[0] = 5
    sums[4:7] = 5 * np.array([1.0, 2.0, 3.0])
    P = pa.solve_rigid(sums)
    np.testing.assert_allclose(P[:3, :3], np.eye(3), atol=1e-15)
    np.testing.assert_allclose(P[:3, 3], [1, 2, 3])


def test_host_update_follows_oracle_rule():
    rng = np.random.default_rng(1)
    cfgs = [dict(max_iterations=5), dict(max_iterations=50, transformation_epsilon=1e-6, rotation_epsilon=1e-6),
            dict(max_iterations=50, relative_mse_epsilon=1e-3), dict(max_iterations=1)]
    for cfg in cfgs:
        p = pa.default_params(**cfg)
        op = ora.params_from(p)
        pose_a = pose_b = st.rigid_transform(3.0)
        mse_a = mse_b = 0.0
        for it in range(8):
            sums = random_sums(rng, noise=0.02 / (1 + it) ** 3)
            sa, pose_a, mse_a, done_a, conv_a = pa.host_update(p, sums, it, pose_a, mse_a)
            sb, pose_b, mse_b, done_b, conv_b = ora.host_update(op, sums, it, pose_b, mse_b)
            assert (sa, done_a, conv_a) == (sb, done_b, conv_b)
            np.testing.assert_allclose(pose_a, pose_b, atol=1e-10)
            assert abs(mse_a - mse_b) <= 1e-15
            if done_a:
                break
        assert done_a or cfg.get("max_iterations", 50) > 8
    # a failing solve leaves the pose untouched
    p = pa.default_params(max_iterations=5)
    pose = st.rigid_transform(3.0)
    stt, pose2, _, _, _ = pa.host_update(p, np.r_[1.0, np.zeros(16)], 0, pose, 0.0)
    assert stt == pa._lib.ERR_TOO_FEW_PAIRS and np.array_equal(pose2, pose)


def test_convergence_criteria_semantics():
    ident = np.zeros(17)
    pts = np.random.default_rng(2).standard_normal((30, 3)).astype(np.float32)
    ident = ora.accumulate(pts, pts, np.arange(30, dtype=np.int32), np.zeros(30, np.float32))
    # criterion A: zero incremental transform -> converged at once
    p = pa.default_params(max_iterations=10, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    stt, pose, mse, done, conv = pa.host_update(p, ident, 0, np.eye(4), 0.0)
    assert stt == 0 and done and conv
    # criterion B needs a previous iteration
    p = pa.default_params(max_iterations=10, relative_mse_epsilon=0.5)
    sums = ident.copy()
    sums[16] = 3.0
    _, _, mse, done, conv = pa.host_update(p, sums, 0, np.eye(4), 0.0)
    assert not done and not conv and mse == 0.1
    sums[16] = 2.9
    _, _, mse, done, conv = pa.host_update(p, sums, 1, np.eye(4), mse)
    assert done and conv
    # max_iterations ends the loop without "converged"
    p = pa.default_params(max_iterations=2)
    _, _, _, done, conv = pa.host_update(p, sums, 1, np.eye(4), 0.1)
    assert done and not conv


def test_pose7_matches_reference_formula_and_scipy():
    g = np.load(os.path.join(GOLD, "pose7.npz"))
    for P, a, b in zip(g["poses"], g["pose7"], g["pose7_ref"]):
        assert np.array_equal(pa.pose_to_pose7(P, False), a)
        assert np.array_equal(pa.pose_to_pose7(P, True), b)      # orientation.z dropped, as opencv_surface_match.cpp:138-141
    rng = np.random.default_rng(3)
    for _ in range(100):
        Rm = Rotation.from_rotvec(rng.standard_normal(3) * 0.8)
        P = np.eye(4)
        P[:3, :3] = Rm.as_matrix()
        P[:3, 3] = rng.standard_normal(3)
        p7 = pa.pose_to_pose7(P)
        q = Rm.as_quat()
        q = q if q[3] >= 0 else -q
        # the reference evaluates sqrt(a-b-c+1) in float32: a component near 0 carries ~sqrt(eps_f32) absolute error
        np.testing.assert_allclose(p7[3:], q, atol=5e-4)
        np.testing.assert_allclose(np.linalg.norm(p7[3:]), 1.0, atol=1e-5)
        np.testing.assert_allclose(p7[:3], P[:3, 3].astype(np.float32))
        assert np.array_equal(p7, ora.pose_to_pose7(P))


def test_params_from_reference_call_site():
    p = pa.params_from_reference(250, 0.005, 2.5, 8)   # opencv_surface_match.cpp:85
    assert p.max_iterations == 250 and abs(p.relative_mse_epsilon - 0.005) < 1e-9
    assert p.max_correspondence_distance == 0.0 and p.transformation_epsilon == 0.0
    d = pa.default_params()
    assert d.max_iterations == 50 and d.relative_mse_epsilon == 0.0 and d.flags == 0

"""The UPSTREAM-FAITHFUL variant of reference mode (oracle/icp_oracle.c, second half; MI355_ICP_FLAG_UPSTREAM in the product),
CPU side: the oracle's ALGEBRAIC restatement of upstream's pre-normalisation against a LITERAL numpy restatement that
materialises upstream's clouds step by step (tests/upstream_literal.py; independent code: numpy + scipy cKDTree), the
sampling-step and pass-limit rules, and what the variant changes against the round 1-3 form.

PARITY UNPINNED: both restatements are written from upstream's published algorithm as recalled; its source is absent from
/root/reference and from this image, the reference holds no vectors for this path."""
import numpy as np
import pytest

import oracle_lib as ora
import pose_estimation_amd as pa
import synth_twin as st
import upstream_literal as ul


def pair66(ns, nt, k=-1, shift=(0.0, 0.0, 0.0)):
    """model (N x 6), scene (N x 6) moved by `shift` (a camera looks at the scene from a distance), ground truth."""
    T = st.perturbation(k)
    off = 0 if k < 0 else 16 * k
    tgt = st.synth_cloud_np(st.TARGET_SEED + off, nt, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED + off, ns, noise_seed=st.NOISE_SEED + off, sigma=st.SIGMA, transform=T, normals=True)
    tgt[:, :3] += np.asarray(shift, dtype=np.float32)
    S = np.eye(4)
    S[:3, 3] = shift
    return src, tgt, S @ np.linalg.inv(T)


def test_sampling_step_is_cvround_of_the_model_size():
    # n > 64 * 2^l: exactly 2^l (every level of the reference's 8-level call from 16,385 model points up)
    for n in (16385, 20000, 100000, 1 << 20):
        assert [ora.upstream_step(n, l) for l in range(8)] == [1 << l for l in range(8)]
    # small models: a neighbouring integer; the product's host function agrees with the oracle
    assert [ora.upstream_step(1000, l) for l in range(8)] == [1, 2, 4, 8, 16, 32, 62, 125]  # 62: 62.5 rounds half to even
    for n in (6, 50, 777, 1000, 1337, 2500, 5000, 16384, 16385):
        for l in range(12):
            want = ora.upstream_step(n, l)
            num = int(np.rint(n / 2.0 ** l))  # half to even, as cvRound
            assert want == (0 if num < 1 else max(1, int(np.rint(n / num)))), (n, l)
            assert pa.upstream_step(n, l) == want, (n, l)
    assert ora.upstream_step(50, 7) == 0  # 50 / 128 rounds to 0 samples: the level does not exist


def test_pass_limit_rounds_half_to_even():
    assert [ora.upstream_max_iterations(250, l) for l in range(8)] == [250, 125, 83, 62, 50, 42, 36, 31]  # 62.5 -> 62


@pytest.mark.parametrize("ns,nt,shift", [(20000, 20000, (0, 0, 0)), (6000, 30000, (0.3, -0.2, 1.5)), (1000, 3000, (0, 0, 1.0)),
                                         (2500, 2500, (0.3, -0.2, 1.5)), (777, 1337, (-0.4, 0.1, 0.8))])
def test_algebraic_normalisation_matches_the_materialised_clouds(ns, nt, shift):
    """The oracle keeps the clouds as they are and carries upstream's centre and scale through the formulas; the literal
    restatement builds upstream's float32 clouds (candidate-transformed, centred, scaled, level-start) and runs from the
    identity.  Same passes on every level, poses equal up to the float32 roundings of those intermediate clouds, the
    residual - upstream's number, in ITS units - equal to 1e-6 relative."""
    src, tgt, gt = pair66(ns, nt, shift=shift)
    init = np.eye(4)
    init[:3, 3] = shift
    init = st.rigid_transform(1.5, (0.2, 1, -0.4), (0.004, -0.003, 0.002)) @ init
    o = ora.icp_reference_upstream(src, tgt, init, nthreads=ora.max_threads())
    lit = ul.register_model_to_scene(src, tgt, init)
    assert o["status"] == 0
    assert list(o["level_passes"]) == list(lit["level_passes"])
    assert np.linalg.norm(o["pose"] - lit["pose"]) <= 2e-6
    assert abs(o["residual"] - lit["residual"]) <= 1e-6 * lit["residual"]
    assert abs(o["scale"] - lit["scale"]) <= 1e-7 * lit["scale"]
    np.testing.assert_allclose(o["centre"], lit["centre"], atol=2e-7)
    assert np.linalg.norm(o["pose"] - gt) < 3e-3


def test_from_centred_is_the_conjugation_by_the_centre():
    rng = np.random.default_rng(5)
    X = st.rigid_transform(7.0, (0.3, -1, 0.2), (0.05, 0.01, -0.02))
    T0 = st.rigid_transform(33.0, (1, 0.5, 0.25), (0.3, -0.2, 1.5))
    c = rng.uniform(-1, 1, 3).astype(np.float32)
    C = np.eye(4)
    C[:3, 3] = -c.astype(np.float64)
    np.testing.assert_allclose(ora.from_centred(X, c, T0), np.linalg.inv(C) @ X @ C @ T0, atol=1e-15)


def test_pass_about_the_centre_matches_numpy():
    """Normal equations about the centre and the six-column residual, against explicit numpy on the oracle's own winners."""
    src, tgt, gt = pair66(3000, 2000, shift=(0.2, 0.1, 0.9))
    init = np.eye(4)
    init[:3, 3] = (0.2, 0.1, 0.9)
    start = st.rigid_transform(1.0) @ init
    c, scale = ora.upstream_normalisation(src, tgt, init)
    n, sums, idx, d2, sel, thr = ora.ref_pass_ex(src, tgt, start, start, 2.5, centre=c)
    n0, sums0, idx0, d20, sel0, thr0 = ora.ref_pass_ex(src, tgt, start, start, 2.5, centre=None)
    # correspondences, threshold and winners do not depend on the form
    assert np.array_equal(idx, idx0) and np.array_equal(d2, d20) and np.array_equal(sel, sel0) and thr == thr0 and n == n0
    assert sums0[29] == 0.0
    p = ora.transform(start, src[:, :3])[sel].astype(np.float64)
    q = tgt[idx[sel], :3].astype(np.float64)
    nq = tgt[idx[sel], 3:].astype(np.float64)
    pc = p - c.astype(np.float64)
    A = np.concatenate([np.cross(pc, nq), nq], axis=1)
    b = ((q - p) * nq).sum(1)
    N = A.T @ A
    np.testing.assert_allclose(sums[:21], N[np.triu_indices(6)], rtol=1e-11)
    np.testing.assert_allclose(sums[21:27], A.T @ b, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sums[27], ((p - q) ** 2).sum(), rtol=1e-12)
    R32 = ora.pose_to_f32(start).reshape(3, 4)[:, :3].astype(np.float64)
    n2 = src[sel, 3:].astype(np.float64) @ R32.T
    npf = (n2 / np.linalg.norm(n2, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(sums[29], ((npf - nq) ** 2).sum(), rtol=1e-9)
    # the legacy sums are the same equations about the origin
    A0 = np.concatenate([np.cross(p, nq), nq], axis=1)
    np.testing.assert_allclose(sums0[:21], (A0.T @ A0)[np.triu_indices(6)], rtol=1e-11)


def test_what_the_variant_changes_against_the_round_1_to_3_form():
    """(i) A large model near the cloud origin: both forms land on the same pose (the stop rule fires on the same passes
    there).  (ii) A 1000-point model a metre from the camera: the coarse levels hold 8-16 points; the round 1-3 form
    linearises the rotation about the CAMERA origin and its minimum-norm solves throw the pose away, upstream's form - about
    the joint mean - converges.  This is why the adapter's default follows upstream."""
    src, tgt, gt = pair66(20000, 20000)
    up = ora.icp_reference_upstream(src, tgt, np.eye(4), nthreads=ora.max_threads())
    leg = ora.icp_reference(src[:, :3], tgt, np.eye(4), nthreads=ora.max_threads())
    assert np.linalg.norm(up["pose"] - leg["pose"]) < 1e-6 and up["passes"] == leg["passes"]
    assert up["residual"] > leg["residual"]  # six columns, normalised units: not the same number
    src, tgt, gt = pair66(1000, 3000, shift=(0, 0, 1.0))
    init = np.eye(4)
    init[2, 3] = 1.0
    up = ora.icp_reference_upstream(src, tgt, init)
    leg = ora.icp_reference(src[:, :3], tgt, init)
    assert np.linalg.norm(up["pose"] - gt) < 3e-3
    assert np.linalg.norm(leg["pose"] - gt) > 1.0


def test_golden_upstream_fixture_reproduced():
    """tests/golden/refmode_upstream_1k.npz (make_golden.py (8): written only after the literal restatement agreed)."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "refmode_upstream_1k.npz"))
    meta = json.load(open(os.path.join(gold, "golden.json")))["refmode_upstream_1k"]
    it, tol, scale, levels = meta["params"]
    r = ora.icp_reference_upstream(g["src6"], g["tgt6"], g["init"], it, tol, scale, levels)
    assert r["passes"] == meta["passes"] and r["residual"] == meta["residual"] and r["scale"] == meta["scale"]
    assert np.array_equal(r["pose"], g["final_pose"]) and np.array_equal(r["centre"], g["centre"]) and list(r["level_passes"]) == list(g["level_passes"])
    assert [ora.upstream_step(1000, l) for l in range(8)] == meta["steps"]
    step = meta["pass_step"]
    n, sums, idx, d2, sel, thr = ora.ref_pass_ex(g["src6"][::step][:1000 // step], g["tgt6"][::step][:3000 // step], g["start"], g["cur"], scale,
                                                 centre=g["centre"])
    assert n == meta["pass_selected"] and np.array_equal(idx, g["pass_idx"]) and np.array_equal(d2, g["pass_d2"])
    assert np.array_equal(sel, g["pass_sel"]) and np.float32(thr) == g["pass_thr"] and np.array_equal(sums, g["pass_sums"])

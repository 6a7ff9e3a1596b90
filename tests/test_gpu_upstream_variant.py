"""MI355_ICP_FLAG_UPSTREAM on the GPU against the oracle's upstream-faithful variant (oracle/icp_oracle.c): the pre-normalisation
(centre bit for bit, scale to rounding), per-pass results on levels whose step is NOT a power of two (index, d2, threshold,
winners bit-exact; the 30 sums 1e-10), whole alignments (passes per job equal, pose <= 1e-5 Frobenius - observed 1e-12 -, the
residual in upstream's units), the request shape of the reference, mixed models in one call, and the host mirrors' default.
PARITY UNPINNED (see the oracle's header): upstream's source is not available offline."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5
FLAG_UPSTREAM = 4


def pair66(ns, nt, k=-1, shift=(0.0, 0.0, 0.0)):
    T = st.perturbation(k)
    off = 0 if k < 0 else 16 * k
    tgt = st.synth_cloud_np(st.TARGET_SEED + off, nt, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED + off, ns, noise_seed=st.NOISE_SEED + off, sigma=st.SIGMA, transform=T, normals=True)
    tgt[:, :3] += np.asarray(shift, dtype=np.float32)
    S = np.eye(4)
    S[:3, 3] = shift
    return src, tgt, S @ np.linalg.inv(T)


def shifted(shift, extra=None):
    init = np.eye(4)
    init[:3, 3] = shift
    return init if extra is None else extra @ init


@pytest.mark.parametrize("ns,nt,shift", [(3000, 2500, (0.2, 0.1, 0.9)), (1000, 3000, (0, 0, 1.0)), (20000, 100000, (0.3, -0.2, 1.5)),
                                         (40000, 30000, (0, 0, 0))])
def test_normalisation_matches_oracle(engine, ns, nt, shift):
    src, tgt, gt = pair66(ns, nt, shift=shift)
    engine.set_targets([tgt])
    engine.set_sources([src])
    for init in (shifted(shift), shifted(shift, st.rigid_transform(4.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015)))):
        c, s = engine.upstream_normalisation(init)
        co, so = ora.upstream_normalisation(src, tgt, init)
        # the centre is a float32 rounding of sums that differ in their association only: equal, or one ulp apart at a tie
        assert np.all(np.abs(c.view(np.int32) - co.view(np.int32)) <= 1), (c, co)
        assert abs(s - so) <= 1e-10 * so


@pytest.mark.parametrize("ns,nt", [(1000, 3000), (1337, 777), (5000, 2500), (25000, 9000)])
def test_pass_about_the_centre_on_upstream_steps(engine, ns, nt):
    """Levels sampled with upstream's step for this model size (for 1000 points: 125, 62, 32 - not 2^l), both forms."""
    shift = (0.2, 0.1, 0.9)
    src, tgt, gt = pair66(ns, nt, shift=shift)
    engine.set_targets([tgt])
    engine.set_sources([src])
    start = st.rigid_transform(0.7, (0.1, 1, 0.3), (0.004, 0.001, -0.002)) @ shifted(shift)
    cur = st.rigid_transform(0.2, (1, 0, 0), (0.001, 0, 0)) @ start
    centre, _ = ora.upstream_normalisation(src, tgt, shifted(shift))
    steps = sorted({ora.upstream_step(ns, l) for l in range(8)} - {0})
    assert ns > 1400 or any(s & (s - 1) for s in steps)  # the small models: a step that is not a power of two is among them
    for step in steps:
        if ns // step < 6 or nt // step < 1:
            continue
        s_l, t_l = src[::step][:ns // step], tgt[::step][:nt // step]
        for scale in (2.5, 0.0):
            for c in (centre, None):
                idx, d2, sel, thr, sums = engine.reference_pass_ex(start, cur, step, scale, centre=c)
                n_o, s_o, idx_o, d2_o, sel_o, thr_o = ora.ref_pass_ex(s_l, t_l, start, cur, scale, centre=c, nthreads=ora.max_threads())
                assert np.array_equal(idx, idx_o) and np.array_equal(d2.view(np.uint32), d2_o.view(np.uint32)), (step, scale)
                assert (thr == thr_o) or (np.isinf(thr) and np.isinf(thr_o))
                assert np.array_equal(sel, sel_o)
                assert sums[28] == n_o
                np.testing.assert_allclose(sums, s_o, rtol=1e-10, atol=1e-12)
                assert (sums[29] > 0) == (c is not None)


@pytest.mark.parametrize("ns,nt,shift", [(20000, 20000, (0, 0, 0)), (1000, 3000, (0, 0, 1.0)), (777, 1337, (-0.4, 0.1, 0.8)),
                                         (6000, 30000, (0.3, -0.2, 1.5))])
def test_align_upstream_matches_oracle(engine, ns, nt, shift):
    src, tgt, gt = pair66(ns, nt, shift=shift)
    engine.set_targets([tgt])
    engine.set_sources([src])
    init = shifted(shift, st.rigid_transform(1.5, (0.2, 1, -0.4), (0.004, -0.003, 0.002)))
    res = engine.align_reference(init[None], 250, 0.005, 2.5, 8, flags=FLAG_UPSTREAM)
    o = ora.icp_reference_upstream(src, tgt, init, 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
    assert res.status[0] == 0 and res.passes[0] == o["passes"]
    assert np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL
    assert abs(res.residual[0] - o["residual"]) <= 1e-9 * max(1.0, o["residual"])
    assert np.linalg.norm(res.poses[0] - gt) < 3e-3
    # the round 1-3 form is still there behind flags = 0 and is a different computation on the same clouds
    leg = engine.align_reference(init[None], 250, 0.005, 2.5, 8)
    o_leg = ora.icp_reference(src[:, :3], tgt, init, 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
    assert leg.passes[0] == o_leg["passes"] and np.linalg.norm(leg.poses[0] - o_leg["pose"]) <= POSE_TOL
    assert abs(leg.residual[0] - o_leg["residual"]) <= 1e-9
    assert res.residual[0] != leg.residual[0]


def test_request_shape_six_candidates_upstream(engine):
    """The reference's own request: 6 candidates of one 20k-point model (N x 6) against a 100k-point scene a metre and a half
    from the camera; every candidate has its own centre and scale."""
    import pose_estimation_amd as pa
    shift = (0.3, -0.2, 1.5)
    src, tgt, gt = pair66(20000, 100000, shift=shift)
    inits = np.stack([shifted(shift)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    engine.set_targets([tgt])
    engine.set_sources([src])
    res = engine.align_reference(inits, 250, 0.005, 2.5, 8, flags=FLAG_UPSTREAM)
    for k in range(6):
        o = ora.icp_reference_upstream(src, tgt, inits[k], 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
        assert res.status[k] == 0 and res.passes[k] == o["passes"], k
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL, k
        assert abs(res.residual[k] - o["residual"]) <= 1e-9 * max(1.0, o["residual"]), k
    # the Python mirror of cv::ppf_match_3d::ICP runs this form by default
    icp = pa.ICP(250, 0.005, 2.5, 8, engine=engine)
    assert icp.upstream and icp.mode == "reference"
    poses = [pa.Pose3D() for _ in range(6)]
    for p, init in zip(poses, inits):
        p.updatePose(init)
    icp.registerModelToScene(src, tgt, poses)
    for k in range(6):
        assert np.array_equal(poses[k].pose, res.poses[k]) and poses[k].residual == res.residual[k]
    # and says so when the model comes without normals
    with pytest.raises(pa.IcpError) as e:
        icp.registerModelToScene(src[:, :3].copy(), tgt, [pa.Pose3D()])
    assert e.value.status == pa._lib.ERR_INVALID_ARGUMENT and "normals" in str(e.value)
    legacy = pa.ICP(250, 0.005, 2.5, 8, engine=engine, upstream=False)
    p = pa.Pose3D()
    p.updatePose(inits[0])
    legacy.registerModelToScene(src[:, :3].copy(), tgt, [p])
    o = ora.icp_reference(src[:, :3], tgt, inits[0], 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
    assert np.linalg.norm(p.pose - o["pose"]) <= POSE_TOL


def test_mixed_models_sample_with_their_own_steps(engine):
    """Three models of different sizes in one call: below 16,385 points their levels sample with different steps, so the
    level loop runs them in groups by step; targets are sampled with the step of the job's MODEL."""
    pairs = [pair66(1000 + 2333 * k, 4000 + 1300 * k, k, shift=(0.1 * k, -0.05, 0.7 + 0.2 * k)) for k in range(3)]
    engine.set_targets([p[1] for p in pairs])
    engine.set_sources([p[0] for p in pairs])
    sid = [0, 1, 2, 1, 0, 2]
    tid = [0, 1, 2, 1, 0, 2]
    inits = []
    for j, s in enumerate(sid):
        base = shifted((0.1 * s, -0.05, 0.7 + 0.2 * s))
        inits.append(base if j < 3 else st.rigid_transform(1.0 + j, (0.2, 1, -0.4), (0.004, -0.003, 0.002)) @ base)
    res = engine.align_reference(np.stack(inits), 120, 0.005, 2.5, 7, source_ids=sid, target_ids=tid, flags=FLAG_UPSTREAM)
    for k in range(len(sid)):
        o = ora.icp_reference_upstream(pairs[sid[k]][0], pairs[tid[k]][1], inits[k], 120, 0.005, 2.5, 7)
        assert res.passes[k] == o["passes"], k
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL, k
        assert abs(res.residual[k] - o["residual"]) <= 1e-9 * max(1.0, o["residual"]), k


def test_pass_limit_and_tiny_models(engine):
    """A model of 40 points: levels whose sample count rounds to zero do not exist, levels with fewer than six model points
    are skipped; and a tolerance of 0 runs every level to upstream's pass limit (cvRound: 250 / 4 -> 62)."""
    src, tgt, gt = pair66(40, 60, shift=(0, 0, 0.5))
    engine.set_targets([tgt])
    engine.set_sources([src])
    res = engine.align_reference(shifted((0, 0, 0.5))[None], 50, 0.005, 2.5, 8, flags=FLAG_UPSTREAM)
    o = ora.icp_reference_upstream(src, tgt, shifted((0, 0, 0.5)), 50, 0.005, 2.5, 8)
    assert res.passes[0] == o["passes"] and np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL
    src, tgt, gt = pair66(3000, 3000)
    engine.set_targets([tgt])
    engine.set_sources([src])
    res = engine.align_reference(np.eye(4)[None], 12, 0.0, 2.5, 4, flags=FLAG_UPSTREAM)
    o = ora.icp_reference_upstream(src, tgt, np.eye(4), 12, 0.0, 2.5, 4, nthreads=ora.max_threads())
    assert res.passes[0] == o["passes"] == sum(ora.upstream_max_iterations(12, l) for l in range(4))
    assert np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL


def test_sharded_code_path_with_one_rank_is_bit_identical_upstream(engine):
    """A one-rank external communicator sends the upstream variant through the collective layer too (model sums, distance
    sums, histograms, owner table, rank-ordered rows): the single-GPU bits."""
    shift = (0.1, 0.2, 0.8)
    src, tgt, gt = pair66(30011, 41000, shift=shift)
    engine.set_targets([tgt])
    engine.set_sources([src])
    init = np.stack([shifted(shift), st.rigid_transform(1.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt])
    ref = engine.align_reference(init, 120, 0.005, 2.5, 7, source_ids=[0, 0], target_ids=[0, 0], flags=FLAG_UPSTREAM)
    engine.comm_attach_external(0, 1, lambda send: bytes(send))
    try:
        got = engine.align_reference(init, 120, 0.005, 2.5, 7, source_ids=[0, 0], target_ids=[0, 0], flags=FLAG_UPSTREAM)
    finally:
        engine.comm_detach()
    assert np.array_equal(got.passes, ref.passes) and np.array_equal(got.poses, ref.poses) and np.array_equal(got.residual, ref.residual)


def test_scene_prepared_on_the_device_feeds_the_upstream_variant(engine):
    """Camera frame -> filters -> normals -> commit (the scene never leaves the GPU; its coordinate sums for upstream's joint
    mean ride on the normals' synchronisation) -> upstream-variant refine, against the oracle on the downloaded scene."""
    import pose_estimation_amd as pa
    shift = (0.1, -0.1, 1.2)
    src, tgt, gt = pair66(5000, 20000, shift=shift)
    raw = tgt[:, :3].copy()
    raw[::17] = np.nan
    engine.set_sources([src])
    engine.scene_load(raw)
    engine.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
    normals = engine.scene_normals(20, True, (0.0, 0.0, 0.0))
    scene = engine.scene_download()
    engine.scene_commit_target()
    scene6 = np.concatenate([scene, normals], axis=1)
    init = shifted(shift)
    res = engine.align_reference(init[None], 250, 0.005, 2.5, 8, flags=FLAG_UPSTREAM)
    o = ora.icp_reference_upstream(src, scene6, init, 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
    assert res.passes[0] == o["passes"] and np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL
    # committed WITHOUT normals first (no fresh sums), then again: the commit computes the sums itself
    engine.scene_load(raw)
    engine.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
    engine.scene_commit_target()
    engine.set_targets([scene6])
    res2 = engine.align_reference(init[None], 250, 0.005, 2.5, 8, flags=FLAG_UPSTREAM)
    assert res2.passes[0] == o["passes"] and np.linalg.norm(res2.poses[0] - o["pose"]) <= POSE_TOL


def test_cpp_harness_runs_the_upstream_variant_by_default(engine):
    import pose_estimation_amd as pa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "estimate_pose_harness")
    out = subprocess.run([exe, "2500", "3000", "4", "reference"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]).reshape(4, 4)
             for l in out.stdout.splitlines() if l.startswith("pose ")}
    T = st.perturbation(-1)
    scene = pa.synth_cloud(st.TARGET_SEED, 3000, normals=True)
    model = pa.synth_cloud(st.SOURCE_SEED, 2500, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)
    engine.set_targets([scene])
    engine.set_sources([model])
    inits = np.stack([st.rigid_transform(1.0 + k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) for k in range(4)])
    res = engine.align_reference(inits, 250, 0.005, 2.5, 8, source_ids=[0] * 4, target_ids=[0] * 4, flags=FLAG_UPSTREAM)
    for k in range(4):
        assert np.linalg.norm(poses[k] - res.poses[k]) <= 1e-7, k   # candidate builders differ in the last ulp of cos/sin
        o = ora.icp_reference_upstream(model, scene, inits[k], 250, 0.005, 2.5, 8)
        assert np.linalg.norm(poses[k] - o["pose"]) <= POSE_TOL


def test_golden_upstream_fixture(engine):
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "refmode_upstream_1k.npz"))
    meta = json.load(open(os.path.join(gold, "golden.json")))["refmode_upstream_1k"]
    it, tol, scale, levels = meta["params"]
    engine.set_targets([g["tgt6"]])
    engine.set_sources([g["src6"]])
    idx, d2, sel, thr, sums = engine.reference_pass_ex(g["start"], g["cur"], meta["pass_step"], scale, centre=g["centre"])
    assert np.array_equal(idx, g["pass_idx"]) and np.array_equal(d2.view(np.uint32), g["pass_d2"].view(np.uint32))
    assert np.array_equal(sel, g["pass_sel"]) and np.float32(thr) == g["pass_thr"]
    np.testing.assert_allclose(sums, g["pass_sums"], rtol=1e-10, atol=1e-12)
    res = engine.align_reference(g["init"][None], it, tol, scale, levels, flags=FLAG_UPSTREAM)
    assert res.passes[0] == meta["passes"] and abs(res.residual[0] - meta["residual"]) <= 1e-9 * max(1.0, meta["residual"])
    assert np.linalg.norm(res.poses[0] - g["final_pose"]) <= POSE_TOL
    c, s_ = engine.upstream_normalisation(g["init"])
    assert np.all(np.abs(c.view(np.int32) - g["centre"].view(np.int32)) <= 1) and abs(s_ - meta["scale"]) <= 1e-10 * meta["scale"]


@pytest.mark.parametrize("upstream", [False, True], ids=["round1to3_form", "upstream_form"])
def test_reference_mode_without_the_pass_ending_launch_against_the_oracle(upstream):
    """MI355_ICP_PASS_TAIL=0: the separate final-reduce launch + device-to-host copy per pass - the A/B partner of the fence-free
    pass-ending launch (icp_internal.h: finish_pass), which every other reference-mode test in this suite runs through - against
    the oracle, both forms, multi-workgroup and single-workgroup jobs."""
    import pose_estimation_amd as pa
    shift = (0.2, 0.1, 0.9)
    src, tgt, gt = pair66(9000, 12000, shift=shift)
    inits = np.stack([shifted(shift)] + [st.rigid_transform(a, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for a in (1.0, 3.0, 5.0)])
    old = os.environ.get("MI355_ICP_PASS_TAIL")
    os.environ["MI355_ICP_PASS_TAIL"] = "0"
    try:
        plain = pa.IcpEngine(0)
    finally:
        if old is None:
            os.environ.pop("MI355_ICP_PASS_TAIL")
        else:
            os.environ["MI355_ICP_PASS_TAIL"] = old
    with plain:
        plain.set_targets([tgt])
        plain.set_sources([src])
        res = plain.align_reference(inits, 250, 0.005, 2.5, 8, flags=FLAG_UPSTREAM if upstream else 0)
    for k in range(len(inits)):
        o = (ora.icp_reference_upstream if upstream else ora.icp_reference)(src if upstream else src[:, :3], tgt, inits[k], 250, 0.005, 2.5, 8,
                                                                            nthreads=ora.max_threads())
        assert res.passes[k] == o["passes"], k
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL, k
        assert abs(res.residual[k] - o["residual"]) <= 1e-9 * max(1.0, o["residual"]), k

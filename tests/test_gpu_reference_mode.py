"""Reference mode on the GPU against the oracle: per-pass (index, d2, threshold, winners bit-exact; sums 1e-10),
and whole alignments (pose <= 1e-5 Frobenius, same number of passes)."""
import os

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5


def pair6(ns, nt, k=-1):
    T = st.perturbation(k)
    off = 0 if k < 0 else 16 * k
    tgt = st.synth_cloud_np(st.TARGET_SEED + off, nt, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED + off, ns, noise_seed=st.NOISE_SEED + off, sigma=st.SIGMA, transform=T)
    return src, tgt, np.linalg.inv(T)


# sizes: every register blocking of the one-workgroup selection (4, 8, 20, 32 values per thread) and the LDS kernel above them
@pytest.mark.parametrize("ns,nt", [(3000, 2500), (1337, 777), (40000, 30000), (5000, 100), (25000, 9000), (18000, 4000)])
def test_reference_pass_matches_oracle(engine, ns, nt):
    src, tgt, gt = pair6(ns, nt)
    engine.set_targets([tgt])
    engine.set_sources([src])
    start = st.rigid_transform(0.7, (0.1, 1, 0.3), (0.004, 0.001, -0.002))
    cur = st.rigid_transform(0.2, (1, 0, 0), (0.001, 0, 0)) @ start
    for level in (0, 2, 4):
        step = 1 << level
        if ns // step < 6 or nt // step < 1:
            continue
        for scale in (2.5, 0.0):
            idx, d2, sel, thr, sums = engine.reference_pass(start, cur, level, scale)
            n_o, s_o, idx_o, d2_o, sel_o, thr_o = ora.ref_pass(src[::step][:ns // step], tgt[::step][:nt // step], start, cur,
                                                               scale, nthreads=ora.max_threads())
            assert np.array_equal(idx, idx_o) and np.array_equal(d2.view(np.uint32), d2_o.view(np.uint32))
            assert (thr == thr_o) or (np.isinf(thr) and np.isinf(thr_o))
            assert np.array_equal(sel, sel_o)
            assert sums[28] == n_o
            np.testing.assert_allclose(sums, s_o, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("distinct,n", [(1, 3000), (7, 5000), (300, 9000), (2, 64), (5000, 21000)])
def test_rejection_threshold_with_repeated_distances(engine, distinct, n):
    """The rank selection's corner cases: one distinct distance (nothing to select), a handful of values with hundreds of copies
    each (the chosen radix bin never gets small: the selection runs to the last bit), a few hundred (ranked by counting), all
    against the oracle's sort-based median."""
    src, tgt, gt = pair6(distinct, 2000)
    src = np.ascontiguousarray(np.tile(src, (n // distinct + 1, 1))[:n])
    engine.set_targets([tgt])
    engine.set_sources([src])
    start = st.rigid_transform(0.7, (0.1, 1, 0.3), (0.004, 0.001, -0.002))
    for scale in (2.5, 0.0):
        idx, d2, sel, thr, sums = engine.reference_pass(start, start, 0, scale)
        n_o, s_o, idx_o, d2_o, sel_o, thr_o = ora.ref_pass(src, tgt, start, start, scale, nthreads=ora.max_threads())
        assert np.array_equal(idx, idx_o) and np.array_equal(d2.view(np.uint32), d2_o.view(np.uint32))
        assert np.float32(thr).view(np.uint32) == np.float32(thr_o).view(np.uint32)
        assert np.array_equal(sel, sel_o) and sums[28] == n_o


def test_align_reference_matches_oracle_with_the_reference_parameters(engine):
    import pose_estimation_amd as pa
    src, tgt, gt = pair6(20000, 20000)
    engine.set_targets([tgt])
    engine.set_sources([src])
    res = engine.align_reference(np.eye(4)[None], 250, 0.005, 2.5, 8)
    o = ora.icp_reference(src, tgt, np.eye(4), 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
    assert res.status[0] == 0 and res.passes[0] == o["passes"]
    assert np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL
    assert abs(res.residual[0] - o["residual"]) <= 1e-9
    assert np.linalg.norm(res.poses[0] - gt) < 2e-3
    # the Python mirror of cv::ppf_match_3d::ICP in reference mode drives the same path
    icp = pa.ICP(250, 0.005, 2.5, 8, engine=engine, mode="reference", upstream=False)  # the round 1-3 form (N x 3 model)
    p = pa.Pose3D()
    icp.registerModelToScene(src, tgt, [p])
    assert np.array_equal(p.pose, res.poses[0]) and p.residual == res.residual[0]


def test_reference_request_shape_is_independent_of_the_launch_layout(engine):
    """The reference's own request shape - 6 candidates of one 20k-point model against a 100k-point scene - is the uniform batch
    whose matrix-core launches are one round of workgroups and get class-weighted runs of tiles (JobDesc::first_class).  Every
    candidate against the oracle, and the whole result bit for bit against an engine with equal runs (MI355_ICP_CLASS_SPLIT=0)
    and against one whose chip 'has' 48 resident workgroups (other cuts, other class boundary)."""
    import pose_estimation_amd as pa
    T = st.perturbation(-1)
    tgt = st.synth_cloud_np(st.TARGET_SEED, 100_000, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED, 20_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    gt = np.linalg.inv(T)
    inits = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    engine.set_targets([tgt])
    engine.set_sources([src])
    res = engine.align_reference(inits, 250, 0.005, 2.5, 8)
    for k in (0, 3, 5):
        o = ora.icp_reference(src, tgt, inits[k], 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
        assert res.status[k] == 0 and res.passes[k] == o["passes"]
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL and abs(res.residual[k] - o["residual"]) <= 1e-9
    for env in ({"MI355_ICP_CLASS_SPLIT": "0"}, {"MI355_ICP_SCAN_SLOTS": "48"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            with pa.IcpEngine(0) as other:
                other.set_targets([tgt])
                other.set_sources([src])
                res2 = other.align_reference(inits, 250, 0.005, 2.5, 8)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert np.array_equal(res2.poses, res.poses) and np.array_equal(res2.residual, res.residual) and np.array_equal(res2.passes, res.passes)


def test_batched_candidates_and_mixed_objects(engine):
    pairs = [pair6(6000 + 500 * k, 5000 + 300 * k, k) for k in range(3)]
    engine.set_targets([p[1] for p in pairs])
    engine.set_sources([p[0] for p in pairs])
    sid = [0, 1, 2, 1, 1, 1]
    inits = [np.eye(4)] * 3 + [st.rigid_transform(a, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ pairs[1][2] for a in (1.0, 3.0, 6.0)]
    res = engine.align_reference(np.stack(inits), 60, 0.005, 2.5, 5, source_ids=sid, target_ids=sid)
    for k in range(len(sid)):
        o = ora.icp_reference(pairs[sid[k]][0], pairs[sid[k]][1], inits[k], 60, 0.005, 2.5, 5)
        assert res.passes[k] == o["passes"], k
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL, k
        assert abs(res.residual[k] - o["residual"]) <= 1e-9


def test_reference_mode_needs_scene_normals_and_skips_tiny_levels(engine):
    import pose_estimation_amd as pa
    src, tgt, gt = pair6(40, 60)
    engine.set_targets([tgt[:, :3].copy()])
    engine.set_sources([src])
    with pytest.raises(pa.IcpError) as e:
        engine.align_reference(np.eye(4)[None])
    assert e.value.status == pa._lib.ERR_INVALID_ARGUMENT and "normals" in str(e.value)
    engine.set_targets([tgt])
    res = engine.align_reference(np.eye(4)[None], 50, 0.005, 2.5, 8)
    o = ora.icp_reference(src, tgt, np.eye(4), 50, 0.005, 2.5, 8)
    assert res.passes[0] == o["passes"] and np.linalg.norm(res.poses[0] - o["pose"]) <= POSE_TOL


def test_golden_reference_mode_fixture(engine):
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "refmode_2k.npz"))
    meta = json.load(open(os.path.join(gold, "golden.json")))["refmode_2k"]
    it, tol, scale, levels = meta["params"]
    engine.set_targets([g["tgt6"]])
    engine.set_sources([g["src"]])
    idx, d2, sel, thr, sums = engine.reference_pass(g["start"], g["cur"], meta["pass_level"], scale)
    assert np.array_equal(idx, g["pass_idx"]) and np.array_equal(d2.view(np.uint32), g["pass_d2"].view(np.uint32))
    assert np.array_equal(sel, g["pass_sel"]) and np.float32(thr) == g["pass_thr"]
    np.testing.assert_allclose(sums, g["pass_sums"], rtol=1e-10, atol=1e-12)
    res = engine.align_reference(np.eye(4)[None], it, tol, scale, levels)
    assert res.passes[0] == meta["passes"] and abs(res.residual[0] - meta["residual"]) <= 1e-9
    assert np.linalg.norm(res.poses[0] - g["final_pose"]) <= POSE_TOL


def test_cpp_harness_reference_mode_matches_python_engine(engine):
    import subprocess
    import pose_estimation_amd as pa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "estimate_pose_harness")
    out = subprocess.run([exe, "2500", "3000", "4", "reference", "legacy"], capture_output=True, text=True, timeout=120)  # the round 1-3 form
    assert out.returncode == 0, out.stderr
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]).reshape(4, 4)
             for l in out.stdout.splitlines() if l.startswith("pose ")}
    T = st.perturbation(-1)
    scene = pa.synth_cloud(st.TARGET_SEED, 3000, normals=True)
    model = pa.synth_cloud(st.SOURCE_SEED, 2500, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    engine.set_targets([scene])
    engine.set_sources([model])
    inits = np.stack([st.rigid_transform(1.0 + k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) for k in range(4)])
    res = engine.align_reference(inits, 250, 0.005, 2.5, 8, source_ids=[0] * 4, target_ids=[0] * 4)
    for k in range(4):
        assert np.linalg.norm(poses[k] - res.poses[k]) <= 1e-7, k   # candidate builders differ in the last ulp of cos/sin
        o = ora.icp_reference(model, scene, inits[k], 250, 0.005, 2.5, 8)
        assert np.linalg.norm(poses[k] - o["pose"]) <= POSE_TOL


def test_sharded_code_path_with_one_rank_is_bit_identical(engine):
    """A one-rank external communicator sends every exchange of slab-sharded reference mode through the engine's
    collective layer (histogram sum, owner-table min, rank-ordered sums, slab-size gather): the results must be the
    single-GPU run's, bit for bit, and the threshold path switches to the many-workgroup selection."""
    src, tgt, gt = pair6(30011, 41000)
    engine.set_targets([tgt])
    engine.set_sources([src])
    init = np.stack([np.eye(4), st.rigid_transform(1.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt])
    ref = engine.align_reference(init, 120, 0.005, 2.5, 7, source_ids=[0, 0], target_ids=[0, 0])
    sizes = []

    def allgather(send):
        sizes.append(len(send))
        return bytes(send)

    engine.comm_attach_external(0, 1, allgather)
    try:
        got = engine.align_reference(init, 120, 0.005, 2.5, 7, source_ids=[0, 0], target_ids=[0, 0])
    finally:
        engine.comm_detach()
    assert np.array_equal(got.passes, ref.passes) and np.array_equal(got.poses, ref.poses) and np.array_equal(got.residual, ref.residual)
    assert sizes[0] == 8                               # the slab sizes: (cloud count, n) as uint32
    assert 2 * 2048 * 4 in sizes and 2 * 31 * 8 in sizes    # two jobs: histograms, and a row of 30 sums + the flag each
    again = engine.align_reference(init, 120, 0.005, 2.5, 7, source_ids=[0, 0], target_ids=[0, 0])   # and back to unsharded
    assert np.array_equal(again.poses, ref.poses)


def test_more_candidates_than_one_staging_launch_holds(engine):
    """13 candidate poses of one model: more than the 8 job descriptors that travel as kernel arguments of one launch (the
    all-exact scan stages them itself only up to 8; beyond, stage_jobs_kernel runs in chunks), with candidates that leave
    the active set at different passes.  Every candidate against the oracle."""
    src, tgt, gt = pair6(4000, 3500, 0)
    engine.set_targets([tgt])
    engine.set_sources([src])
    inits = [np.eye(4)] + [st.rigid_transform(0.5 * a, (0.2, 1, -0.4), (0.004 * a, -0.003 * a, 0.002 * a)) @ gt for a in range(1, 13)]
    res = engine.align_reference(np.stack(inits), 40, 0.005, 2.5, 4)
    for k in range(len(inits)):
        o = ora.icp_reference(src, tgt, inits[k], 40, 0.005, 2.5, 4)
        assert res.passes[k] == o["passes"], k
        assert np.linalg.norm(res.poses[k] - o["pose"]) <= POSE_TOL, k


def _engine_with_env(name, value):
    import os
    import pose_estimation_amd as pa
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return pa.IcpEngine(0)
    finally:
        if old is None:
            os.environ.pop(name)
        else:
            os.environ[name] = old


def test_pass_ending_launch_gives_the_separate_reduce_kernels_bits_every_time(engine):
    """The accumulation launch that ends a pass by itself (finish_pass: last workgroup of a job adds the partial rows up,
    handed over through agent-scope atomics, and writes the host's result block) must return exactly what the separate
    final-reduce kernel + copy return (MI355_ICP_PASS_TAIL=0) - poses, residuals, pass counts, bit for bit - and the same
    on every one of 150 repetitions (a row read before it had arrived would show up as a changed low bit sooner or later).
    Both control loops: reference mode (6 candidates, multi-workgroup jobs on the fine levels, single-workgroup ones on the
    coarse levels) and the point-to-point loop (5 jobs of different sizes)."""
    import pose_estimation_amd as pa
    src, tgt, gt = pair6(9000, 12000, 0)
    inits = np.stack([np.eye(4)] + [st.rigid_transform(a, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for a in (1.0, 2.0, 3.0, 4.0, 5.0)])
    pairs = [st.make_pair(st.synth_cloud_np, 700 + 911 * k, 900 + 1257 * k, object_index=k) for k in range(5)]
    params = pa.default_params(max_iterations=8)
    with _engine_with_env("MI355_ICP_PASS_TAIL", "0") as plain:
        plain.set_targets([tgt])
        plain.set_sources([src])
        want = plain.align_reference(inits, 250, 0.005, 2.5, 8)
        plain.set_targets([p[1] for p in pairs])
        plain.set_sources([p[0] for p in pairs])
        want_p2p = plain.align(np.stack([np.eye(4)] * 5), params, source_ids=list(range(5)), target_ids=list(range(5)))
    engine.set_targets([tgt])
    engine.set_sources([src])
    for rep in range(150):
        got = engine.align_reference(inits, 250, 0.005, 2.5, 8)
        assert np.array_equal(got.passes, want.passes), rep
        assert np.array_equal(got.poses.view(np.uint64), want.poses.view(np.uint64)), rep
        assert np.array_equal(np.asarray(got.residual).view(np.uint64), np.asarray(want.residual).view(np.uint64)), rep
    engine.set_targets([p[1] for p in pairs])
    engine.set_sources([p[0] for p in pairs])
    for rep in range(150):
        got = engine.align(np.stack([np.eye(4)] * 5), params, source_ids=list(range(5)), target_ids=list(range(5)))
        assert np.array_equal(got.iterations, want_p2p.iterations), rep
        assert np.array_equal(got.poses.view(np.uint64), want_p2p.poses.view(np.uint64)), rep


def test_service_path_picks_the_faster_exact_scan_per_pass(capfd):
    """mi355_icp_align_reference without a scan flag: the planner takes the exact PRUNED scan for the passes of a large frame
    and the brute-force scan for the coarse levels (icp_engine.hip: service_prune_pays).  Whatever it picks, poses, residuals
    and pass counts are those of the all-brute-force call (MI355_ICP_FLAG_BRUTE_SCAN) and of the all-pruned call, bit for bit."""
    import pose_estimation_amd as pa
    T = st.perturbation(-1)
    tgt = st.synth_cloud_np(st.TARGET_SEED, 300_000, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED, 60_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)
    gt = np.linalg.inv(T)
    inits = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    old = os.environ.get("MI355_ICP_PLAN_DEBUG")
    os.environ["MI355_ICP_PLAN_DEBUG"] = "1"
    try:
        eng = pa.IcpEngine(0)
    finally:
        if old is None:
            os.environ.pop("MI355_ICP_PLAN_DEBUG")
        else:
            os.environ["MI355_ICP_PLAN_DEBUG"] = old
    with eng:
        eng.set_targets([tgt])
        eng.set_sources([src])
        for flags in (pa._lib.FLAG_UPSTREAM, 0):
            capfd.readouterr()
            auto = eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
            plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("mi355_icp plan:")]
            # (a forced MI355_ICP_SCAN_MODE / MI355_ICP_SERVICE_PRUNE - tools/gpu_check.sh - takes the choice away from the planner:
            # then only the equality of the results is left to check)
            free = not os.environ.get("MI355_ICP_SCAN_MODE") and not os.environ.get("MI355_ICP_SERVICE_PRUNE")
            assert not free or (any(" pruned 1 " in ln for ln in plans) and any(" pruned 0 " in ln for ln in plans)), plans
            brute = eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags | pa._lib.FLAG_BRUTE_SCAN)
            assert not free or not any(" pruned 1 " in ln for ln in capfd.readouterr().err.splitlines())
            pruned = eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags | pa._lib.FLAG_PRUNED_SCAN)
            for other in (brute, pruned):
                assert np.array_equal(auto.passes, other.passes)
                assert np.array_equal(auto.poses.view(np.uint64), other.poses.view(np.uint64))
                assert np.array_equal(np.asarray(auto.residual).view(np.uint64), np.asarray(other.residual).view(np.uint64))
        # the request shape of the reference (20k model, 100k scene) stays brute force on every level: it is the faster one there
        eng.set_targets([tgt[:100_000].copy()])
        eng.set_sources([src[:20_000].copy()])
        capfd.readouterr()
        eng.align_reference(inits, 250, 0.005, 2.5, 8, flags=pa._lib.FLAG_UPSTREAM)
        assert not free or not any(" pruned 1 " in ln for ln in capfd.readouterr().err.splitlines())

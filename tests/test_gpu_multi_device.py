"""mi355_icp_create_multi: ONE handle over several devices of one process (the reference holds its backend by value
in one node process, pose_estimation.hpp:160-162).  The GPU boxes of this pool have one device, so the handle is created over
device_ids = {0, 0} and {0, 0, 0} - two or three engines, streams and host threads sharing the card, which exercises the same
fan-out, layouts and in-process exchange as distinct devices would.

By job (jobs >= devices): every job's result is the single-device engine's, bit for bit.  By source slab (fewer jobs than
devices): indices and distances bit for bit, sums in device order (1e-12 of the single engine's), poses 1e-12 (point-to-point)
/ 1e-9 (reference mode) and <= 1e-5 of the oracle, pass counts equal."""
import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5


def pair66(ns, nt, k=-1, shift=(0.0, 0.0, 0.0)):
    T = st.perturbation(k)
    off = 0 if k < 0 else 16 * k
    tgt = st.synth_cloud_np(st.TARGET_SEED + off, nt, normals=True)
    src = st.synth_cloud_np(st.SOURCE_SEED + off, ns, noise_seed=st.NOISE_SEED + off, sigma=st.SIGMA, transform=T, normals=True)
    tgt[:, :3] += np.asarray(shift, dtype=np.float32)
    S = np.eye(4)
    S[:3, 3] = shift
    return src, tgt, S @ np.linalg.inv(T)


@pytest.fixture(scope="module", params=[2, 3], ids=["two_slots", "three_slots"])
def multi(request):
    import pose_estimation_amd as pa
    eng = pa.IcpEngine(device_ids=[0] * request.param)
    yield eng
    eng.close()


def test_handle_reports_its_devices(multi):
    rank, n, kind = multi.comm_info()
    assert (rank, n, kind) == (0, len(multi.device_ids), 3)
    assert "gfx950" in multi.device_info()["name"]
    import pose_estimation_amd as pa
    with pytest.raises(pa.IcpError):
        multi.comm_attach_external(0, 1, lambda b: bytes(b))


def test_point_to_point_by_job_is_the_single_engine_bit_for_bit(engine, multi):
    import pose_estimation_amd as pa
    pairs = [st.make_pair(st.synth_cloud_np, 3000 + 811 * k, 2500 + 1257 * k, object_index=k) for k in range(5)]
    ids = [0, 1, 2, 3, 4, 2, 0]   # 7 jobs >= devices: whole jobs per device
    init = np.stack([np.eye(4)] * len(ids))
    p = pa.default_params(max_iterations=12)
    for e in (engine, multi):
        e.set_targets([q[1] for q in pairs])
        e.set_sources([q[0] for q in pairs])
    one = engine.align(init, p, ids, ids)
    got = multi.align(init, p, ids, ids)
    assert np.array_equal(got.poses.view(np.uint64), one.poses.view(np.uint64))
    assert np.array_equal(got.iterations, one.iterations) and np.array_equal(got.status, one.status)
    assert np.array_equal(np.asarray(got.residual).view(np.uint64), np.asarray(one.residual).view(np.uint64))
    i1, d1, s1 = engine.correspond(init, 0.0, ids, ids)
    i2, d2, s2 = multi.correspond(init, 0.0, ids, ids)
    assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(s1, s2)


def test_point_to_point_by_slab_matches_the_single_engine_and_the_oracle(engine, multi):
    """ONE job of 100k x 60k points (the matrix-core scan on every slab): fewer jobs than devices, so every device takes a
    slab of the model and the 17 sums per pass are exchanged through the host in device order."""
    import pose_estimation_amd as pa
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 100_003, 60_000)
    p = pa.default_params(max_iterations=8)
    for e in (engine, multi):
        e.set_targets([tgt])
        e.set_sources([src])
    i1, d1, s1 = engine.correspond(np.eye(4)[None])
    i2, d2, s2 = multi.correspond(np.eye(4)[None])
    assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_allclose(s2, s1, rtol=1e-12, atol=1e-12)
    one = engine.align(np.eye(4)[None], p)
    got = multi.align(np.eye(4)[None], p)
    again = multi.align(np.eye(4)[None], p)
    assert np.array_equal(got.poses.view(np.uint64), again.poses.view(np.uint64))   # run to run: the same bits
    assert got.iterations[0] == one.iterations[0] == 8 and got.status[0] == 0
    assert np.linalg.norm(got.poses[0] - one.poses[0]) <= 1e-12
    o = ora.icp(src, tgt, np.eye(4), ora.params_from(p), nthreads=ora.max_threads())
    assert np.linalg.norm(got.poses[0] - o["pose"]) <= POSE_TOL


def test_layout_follows_the_calls(engine, multi):
    """slab -> by job -> slab on one handle: the models are re-laid out from the host copy, results stay put."""
    import pose_estimation_amd as pa
    pairs = [st.make_pair(st.synth_cloud_np, 9000 + 500 * k, 7000, object_index=k) for k in range(3)]
    p = pa.default_params(max_iterations=5)
    for e in (engine, multi):
        e.set_targets([q[1] for q in pairs])
        e.set_sources([q[0] for q in pairs])
    n = len(multi.device_ids)
    one_job = engine.align(np.eye(4)[None], p, [1], [1])
    many = engine.align(np.stack([np.eye(4)] * 4), p, [0, 1, 2, 1], [0, 1, 2, 1])
    for _ in range(2):
        a = multi.align(np.eye(4)[None], p, [1], [1])                       # 1 job < devices: slabs
        b = multi.align(np.stack([np.eye(4)] * 4), p, [0, 1, 2, 1], [0, 1, 2, 1])   # 4 jobs >= devices: whole jobs
        assert np.linalg.norm(a.poses[0] - one_job.poses[0]) <= 1e-12
        assert np.array_equal(b.poses.view(np.uint64), many.poses.view(np.uint64)), n


@pytest.mark.parametrize("upstream", [False, True], ids=["round1to3_form", "upstream_form"])
def test_reference_mode_by_job_and_by_slab(engine, multi, upstream):
    import pose_estimation_amd as pa
    flags = pa._lib.FLAG_UPSTREAM if upstream else 0
    shift = (0.2, -0.1, 1.1)
    src, tgt, gt = pair66(20000, 60000, shift=shift)
    S = np.eye(4)
    S[:3, 3] = shift
    inits = np.stack([S] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    for e in (engine, multi):
        e.set_targets([tgt])
        e.set_sources([src])
    # the request shape: 6 candidates >= devices -> whole candidates per device, the single engine's bits
    one = engine.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
    got = multi.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
    assert np.array_equal(got.passes, one.passes)
    assert np.array_equal(got.poses.view(np.uint64), one.poses.view(np.uint64))
    assert np.array_equal(np.asarray(got.residual).view(np.uint64), np.asarray(one.residual).view(np.uint64))
    # one candidate < devices -> source slabs; thresholds, masks and winners are the single engine's, the sums' association differs
    one1 = engine.align_reference(inits[1:2], 250, 0.005, 2.5, 8, flags=flags)
    got1 = multi.align_reference(inits[1:2], 250, 0.005, 2.5, 8, flags=flags)
    assert got1.passes[0] == one1.passes[0] and got1.status[0] == 0
    assert np.linalg.norm(got1.poses[0] - one1.poses[0]) <= 1e-9
    assert abs(got1.residual[0] - one1.residual[0]) <= 1e-12 * max(1.0, one1.residual[0])
    o = (ora.icp_reference_upstream if upstream else ora.icp_reference)(src if upstream else src[:, :3], tgt, inits[1], 250, 0.005, 2.5, 8,
                                                                        nthreads=ora.max_threads())
    assert got1.passes[0] == o["passes"] and np.linalg.norm(got1.poses[0] - o["pose"]) <= POSE_TOL


def test_scene_prepared_on_every_device(engine, multi):
    """scene_load -> filter -> plane fit -> normals -> commit on the handle: every device prepares the same scene; the refine
    that follows (by job) matches a single engine fed the same frame."""
    import pose_estimation_amd as pa
    shift = (0.1, -0.1, 1.2)
    src, tgt, gt = pair66(5000, 30000, shift=shift)
    raw = tgt[:, :3].copy()
    raw[::19] = np.nan
    S = np.eye(4)
    S[:3, 3] = shift
    inits = np.stack([S, st.rigid_transform(2.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt, st.rigid_transform(3.0, (1, 0.1, 0)) @ gt])
    out = []
    for e in (engine, multi):
        e.set_sources([src])
        e.scene_load(raw)
        kept = e.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
        nrm = e.scene_normals(20, True, (0.0, 0.0, 0.0))
        e.scene_commit_target()
        out.append((kept, nrm, e.align_reference(inits, 250, 0.005, 2.5, 8, flags=pa._lib.FLAG_UPSTREAM)))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2].poses.view(np.uint64), out[1][2].poses.view(np.uint64)) and np.array_equal(out[0][2].passes, out[1][2].passes)


def test_failures_come_back_as_errors_not_hangs(multi):
    import pose_estimation_amd as pa
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 3000, 2500)
    bad = src.copy()
    bad[17, 1] = np.nan
    with pytest.raises(pa.IcpError) as e:
        multi.set_sources([bad])
    assert e.value.status == pa._lib.ERR_NONFINITE_INPUT
    multi.set_targets([tgt])
    multi.set_sources([src])
    with pytest.raises(pa.IcpError) as e:   # one job (slab layout, every device in the exchange) with a bad cloud id
        multi.align(np.eye(4)[None], pa.default_params(max_iterations=3), [5], [0])
    assert e.value.status == pa._lib.ERR_INVALID_ARGUMENT
    with pytest.raises(pa.IcpError):        # reference mode without scene normals
        multi.set_targets([tgt[:, :3].copy()])
        multi.align_reference(np.eye(4)[None])
    multi.set_targets([tgt])
    res = multi.align(np.eye(4)[None], pa.default_params(max_iterations=3))   # and the handle still works
    assert res.status[0] == 0 and res.iterations[0] == 3


@pytest.mark.parametrize("upstream", [False, True], ids=["round1to3_form", "upstream_form"])
def test_picky_offers_exchange_gives_the_table_reductions_bits(engine, upstream):
    """Slab-sharded reference mode exchanges the picky dedupe either as the whole owner table (min-reduced: 8 B x level targets)
    or as the ranks' OFFERED pairs (all-gathered and applied locally: 16 B x the model's level points) - whichever is fewer bytes.
    A 9000-point model against a 70000-point scene takes the offers on every level; MI355_ICP_PICKY_OFFERS=0 forces the table.
    Same owner table either way, hence the same bits; and the single engine's result to 1e-9."""
    import os
    import pose_estimation_amd as pa
    flags = pa._lib.FLAG_UPSTREAM if upstream else 0
    shift = (0.1, 0.0, 0.9)
    src, tgt, gt = pair66(9000, 70000, shift=shift)
    S = np.eye(4)
    S[:3, 3] = shift
    init = (st.rigid_transform(2.0, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt)[None]
    results = []
    for knob in (None, "0"):
        old = os.environ.get("MI355_ICP_PICKY_OFFERS")
        if knob is not None:
            os.environ["MI355_ICP_PICKY_OFFERS"] = knob
        try:
            eng = pa.IcpEngine(device_ids=[0, 0, 0])
        finally:
            if knob is not None:
                if old is None:
                    os.environ.pop("MI355_ICP_PICKY_OFFERS")
                else:
                    os.environ["MI355_ICP_PICKY_OFFERS"] = old
        with eng:
            eng.set_targets([tgt])
            eng.set_sources([src])
            results.append(eng.align_reference(init, 250, 0.005, 2.5, 8, flags=flags))
    a, b = results
    assert np.array_equal(a.passes, b.passes) and np.array_equal(a.poses.view(np.uint64), b.poses.view(np.uint64))
    assert np.array_equal(np.asarray(a.residual).view(np.uint64), np.asarray(b.residual).view(np.uint64))
    engine.set_targets([tgt])
    engine.set_sources([src])
    one = engine.align_reference(init, 250, 0.005, 2.5, 8, flags=flags)
    assert a.passes[0] == one.passes[0] and np.linalg.norm(a.poses[0] - one.poses[0]) <= 1e-9

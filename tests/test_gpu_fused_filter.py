"""The filter scans (icp_fused_kernels.hip: the fused surrogate, and the expansion surrogate on the sorted/centred target
copy) must give bit-identical results to the all-exact path and the oracle: each is forced on for every size here (the
planner only picks a filter above 2.5e8 pairs per launch), including data built to stress their error bands - near-ties spread
over many chunks (overflow -> exact fallback), two-chunk near-ties, exact duplicates, lattices, clouds far from the
origin (cancellation), subnormal and overflowing distances."""
import os

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st
from gpu_util import engine_with_mode, overflow_host_passes, overflow_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fused", "xfilter", "xmfma", "xhybrid", "xbf16"])
def fused(request):
    eng = engine_with_mode(request.param)
    eng.scan_mode_forced = request.param
    yield eng
    eng.close()


def same_as_oracle(eng, src, tgt, pose, max_corr=0.0, kd=True):
    eng.set_targets([tgt])
    eng.set_sources([src])
    idx, d2, sums = eng.correspond(pose[None], max_corr)
    oi, od, os_ = ora.correspond(pose, src, tgt, max_corr, use_kdtree=kd, nthreads=ora.max_threads())
    assert np.array_equal(d2.view(np.uint32), od.view(np.uint32)), "float32 squared distances differ"
    assert np.array_equal(idx, oi), "nearest-neighbour indices differ"
    np.testing.assert_allclose(sums[0], os_, rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize("ns,nt", [(1, 1), (3, 2), (63, 65), (257, 1023), (1000, 1000), (1025, 1024), (2049, 3333), (5000, 20000),
                                   (40000, 7), (30000, 130000)])
def test_fused_matches_oracle_ragged(fused, ns, nt):
    src, tgt, gt = st.make_pair(st.synth_cloud_np, ns, nt)
    for pose in (np.eye(4), gt, st.rigid_transform(33.0, (0.3, -1, 0.2), (0.5, 0.1, -0.4))):
        same_as_oracle(fused, src, tgt, pose, kd=(ns * nt > 10_000_000))


def test_near_ties_spread_over_many_chunks_take_the_exact_fallback(fused):
    """Targets on a sphere around the queries, radii jittered by a few float32 ulps: thousands of candidates inside the
    filter's band, in hundreds of different chunks.  The counter asserted on is the engine's own
    (mi355_icp_profile.overflow_points): points the resolve could not settle and the all-exact scan answered."""
    rng = np.random.default_rng(0)
    u = rng.standard_normal((20000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    tgt = (u * (1.0 + rng.integers(-3, 4, (20000, 1)) * 6e-8)).astype(np.float32)
    src = (rng.standard_normal((500, 3)) * 1e-4).astype(np.float32)       # all near the centre
    before, host_before = overflow_points(fused), overflow_host_passes(fused)
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)
    assert overflow_host_passes(fused) == host_before, "a small overflow list must be answered on the device"
    assert 100 <= overflow_points(fused) - before <= 500, "the tie cloud did not reach the overflow path"   # device-side pass (<= kOverflowCap)
    # more overflow points than the fused filter's device-side pass takes per job (kOverflowCap = 1024; it streams the whole
    # target cloud per listed point): the pass is repeated through the host path (gather -> all-exact tiled scan -> scatter).
    # The expansion filters have a sorted copy of the targets: their pass walks only the tiles a point can reach and strides over
    # the WHOLE list - they never need the host (round 4; until then the other pass's cap sent them round again too)
    src = (rng.standard_normal((3000, 3)) * 1e-4).astype(np.float32)
    before = overflow_points(fused)
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)
    assert overflow_points(fused) - before > 1024
    host_expected = 1 if fused.scan_mode_forced == "fused" else 0
    assert overflow_host_passes(fused) == host_before + host_expected, "the wrong overflow path ran"


def test_benign_cloud_never_overflows(fused):
    """Every query sits a hair away from one target and far (in units of that hair) from all others: the first granule
    settles every point, so the overflow counter must not move."""
    rng = np.random.default_rng(11)
    tgt = rng.random((20000, 3)).astype(np.float32)
    src = (tgt[rng.permutation(20000)[:4000]] + (rng.standard_normal((4000, 3)) * 1e-6).astype(np.float32)).astype(np.float32)
    before = overflow_points(fused)
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)
    assert overflow_points(fused) == before


def test_two_chunk_near_ties_and_exact_duplicates(fused):
    rng = np.random.default_rng(1)
    tgt = rng.standard_normal((9000, 3)).astype(np.float32) * 5
    src = rng.standard_normal((3000, 3)).astype(np.float32)
    # plant, for every source point, two targets at (almost) the same distance far apart in index
    off = (rng.standard_normal((3000, 3)) * 0.01).astype(np.float32)
    tgt[:3000] = src + off
    tgt[6000:9000] = src - off                                              # mirror image: equal distance up to rounding
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)
    tgt[6000:9000] = tgt[:3000]                                             # exact duplicates: lowest index must win
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)
    tgt[3000:6000] = tgt[:3000]                                             # three copies: overflow path
    same_as_oracle(fused, src, tgt, np.eye(4), kd=False)


def test_lattice_ties_denormals_and_overflowing_distances(fused):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ties.npz"))
    fused.set_targets([g["tgt"]])
    fused.set_sources([g["q"]])
    idx, d2, _ = fused.correspond(np.eye(4)[None])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(d2.view(np.uint32), g["d2"].view(np.uint32))
    rng = np.random.default_rng(5)
    for scale in (1e-20, 1e18, 3e19):
        cloud = (rng.standard_normal((3000, 3)) * scale).astype(np.float32)
        q = (cloud[rng.permutation(3000)[:777]] * np.float32(1.001)).astype(np.float32)
        fused.set_targets([cloud])
        fused.set_sources([q])
        idx, d2, _ = fused.correspond(np.eye(4)[None])
        oi, od = ora.nn_brute(q, cloud, nthreads=ora.max_threads())
        assert np.array_equal(d2.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oi)


def test_far_from_the_origin_and_anisotropic_clouds(fused):
    """Cancellation stress for the expansion filter: the same surface shifted 50 units away, squashed flat, and a scene
    made of two distant clusters (large tiles in Morton order)."""
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 6000, 9000)
    shift = np.float32([50.0, -30.0, 20.0])
    same_as_oracle(fused, (src + shift).astype(np.float32), (tgt + shift).astype(np.float32), np.eye(4), kd=False)
    flat = np.float32([1.0, 1.0, 1e-3])
    same_as_oracle(fused, (src * flat).astype(np.float32), (tgt * flat).astype(np.float32), np.eye(4), kd=False)
    two = np.concatenate([tgt, tgt + np.float32([1000.0, 0, 0])]).astype(np.float32)
    two = two[np.random.default_rng(3).permutation(len(two))]
    same_as_oracle(fused, np.concatenate([src, src + np.float32([1000.0, 0, 0])]).astype(np.float32), two, np.eye(4), kd=False)
    P = st.rigid_transform(25.0, (0.3, 1, -0.2), (40.0, -10.0, 5.0))      # a pose that throws the source far off the target
    same_as_oracle(fused, src, tgt, P, kd=False)


def test_fused_align_golden_and_batched(fused):
    import json
    import pose_estimation_amd as pa
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "c2_100k.npz"))
    src, tgt, gt = st.make_pair(pa.synth_cloud, 100_000, 100_000)
    fused.set_targets([tgt])
    fused.set_sources([src])
    res = fused.align(np.eye(4)[None], pa.default_params(max_iterations=50))
    assert res.iterations[0] == 50 and np.linalg.norm(res.poses[0] - g["final_pose"]) <= 1e-5
    pairs = [st.make_pair(st.synth_cloud_np, 1500 + 331 * k, 1200 + 257 * k, object_index=k) for k in range(4)]
    fused.set_targets([p[1] for p in pairs])
    fused.set_sources([p[0] for p in pairs])
    p = pa.default_params(max_iterations=8, max_correspondence_distance=0.08)
    res = fused.align(np.stack([np.eye(4)] * 4), p, source_ids=[0, 1, 2, 3], target_ids=[0, 1, 2, 3])
    for k in range(4):
        o = ora.icp(pairs[k][0], pairs[k][1], np.eye(4), ora.params_from(p))
        assert res.iterations[k] == o["iterations"] and np.linalg.norm(res.poses[k] - o["pose"]) <= 1e-5


def test_fused_equals_exact_engine_and_oracle_at_1m(fused):
    """Above the planner's threshold the engines run different kernels; every one of the 2^20 (index, d2) answers must be
    the oracle's (exact KD-tree over all 2^20 targets), and the all-exact HIP scan's, bit for bit."""
    import pose_estimation_amd as pa
    n = 1 << 20
    src, tgt, gt = st.make_pair(pa.synth_cloud, n, n)
    oi, od = ora.KdTree(tgt).query(ora.transform(np.eye(4), src), nthreads=ora.max_threads())
    with engine_with_mode("exact") as exact:
        outs = []
        for eng in (fused, exact):
            eng.set_targets([tgt])
            eng.set_sources([src])
            outs.append(eng.correspond(np.eye(4)[None]))
    for idx, d2, sums in outs:
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-12)

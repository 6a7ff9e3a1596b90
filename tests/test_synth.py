"""The synthetic generator (csrc/synth.c) against its numpy twin and the committed hashes."""
import hashlib
import json
import os

import numpy as np

import pose_estimation_amd as pa
import synth_twin as st

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_c_generator_equals_numpy_twin_bitwise():
    for seed, n, first in ((st.TARGET_SEED, 5000, 0), (st.SOURCE_SEED + 16, 3001, 12345), (1, 1, 0)):
        a = pa.synth_cloud(seed, n, first_index=first)
        b = st.synth_cloud_np(seed, n, first_index=first)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    T = st.perturbation(3)
    a = pa.synth_cloud(st.SOURCE_SEED, 4000, first_index=99, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    b = st.synth_cloud_np(st.SOURCE_SEED, 4000, first_index=99, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_slabs_concatenate_to_the_whole_cloud():
    """Slab-sharded ranks generate only their slab: it must be the matching slice of the whole."""
    n, world = 10007, 4
    whole = pa.synth_cloud(st.SOURCE_SEED, n, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=st.perturbation(-1))
    parts, covered = [], 0
    for r in range(world):
        b, c = pa.slab_range(n, r, world)
        assert b == covered
        covered += c
        parts.append(pa.synth_cloud(st.SOURCE_SEED, c, first_index=b, noise_seed=st.NOISE_SEED, sigma=st.SIGMA,
                                    transform=st.perturbation(-1)))
    assert covered == n
    assert np.array_equal(np.concatenate(parts), whole)
    sizes = [pa.slab_range(n, r, world)[1] for r in range(world)]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_surface_shape_and_known_hashes():
    meta = json.load(open(os.path.join(GOLD, "golden.json")))
    src, tgt, gt = st.make_pair(pa.synth_cloud, 1000, 1000)
    assert sha(src) == meta["c1_1k"]["src_sha"] and sha(tgt) == meta["c1_1k"]["tgt_sha"]
    r = np.linalg.norm(tgt.astype(np.float64), axis=1)
    assert 0.55 < r.min() and r.max() < 1.45          # bumpy unit sphere
    assert np.abs(tgt.mean(0)).max() < 0.2
    moved = src.astype(np.float64) @ gt[:3, :3].T + gt[:3, 3]   # undoing the perturbation lands on the surface
    assert abs(np.linalg.norm(moved, axis=1).mean() - r.mean()) < 0.02

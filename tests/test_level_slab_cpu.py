"""Slab-sharded reference mode, host logic (no GPU): mi355_icp_level_slab decides which points of a rank's slab belong to
a pyramid level of the WHOLE cloud.  Property: for any cloud size, rank count and level, the ranks' level points, taken in
rank order, are exactly the single-GPU pyramid's points 0, step, 2 step, ... (n >> level of them), each once, and
level_index_base numbers them consecutively - which is what makes the sharded picky tie rule ("lowest source index")
the whole cloud's rule."""
import numpy as np
import pytest

import pose_estimation_amd as pa


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_level_slabs_partition_the_single_gpu_pyramid(world):
    rng = np.random.default_rng(world)
    sizes = [0, 1, 5, 6, 7, 63, 64, 65, 1000, 4097] + [int(x) for x in rng.integers(1, 200000, 12)]
    for n in sizes:
        slabs = [pa.slab_range(n, r, world) for r in range(world)]
        assert slabs[0][0] == 0 and sum(c for _, c in slabs) == n
        for level in range(0, 9):
            step = 1 << level
            want = np.arange(n >> level, dtype=np.int64) * step          # the single-GPU level: rows 0, step, 2 step, ...
            got, next_index = [], 0
            for begin, count in slabs:
                first, n_local, base, total = pa.level_slab(n, begin, count, level)
                assert total == n >> level
                assert base == next_index                                  # consecutive numbering across ranks
                idx = begin + first + np.arange(n_local, dtype=np.int64) * step
                assert np.all(idx < begin + count) and (n_local == 0 or idx[0] >= begin)
                got.append(idx)
                next_index += n_local
            assert np.array_equal(np.concatenate(got) if got else np.zeros(0, np.int64), want), (n, world, level)


def test_uneven_slabs_and_argument_checks():
    # slabs need not come from slab_range: any contiguous cut works
    n, cuts = 1003, [0, 1, 2, 500, 501, 1003]
    for level in range(0, 7):
        step = 1 << level
        parts = []
        for b, e in zip(cuts[:-1], cuts[1:]):
            first, n_local, base, total = pa.level_slab(n, b, e - b, level)
            parts.append(b + first + np.arange(n_local) * step)
        assert np.array_equal(np.concatenate(parts), np.arange(n >> level) * step)
    with pytest.raises(pa.IcpError):
        pa.level_slab(10, 8, 5, 1)          # slab runs past the cloud
    with pytest.raises(pa.IcpError):
        pa.level_slab(10, 0, 10, 40)        # absurd level

"""Error behaviour of the C-ABI on a live device: bad arguments come back as status codes with a message, never as
crashes, and leave the context usable (the reference swallows exceptions on this path: opencv_surface_match.cpp:144-147)."""
import ctypes as C

import numpy as np
import pytest

import synth_twin as st

pytestmark = pytest.mark.gpu


def test_argument_errors_leave_the_context_usable(engine):
    import pose_estimation_amd as pa
    from pose_estimation_amd import _lib
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 500, 700)
    engine.set_targets([tgt])
    engine.set_sources([src])
    p = pa.default_params(max_iterations=3)
    for kwargs, code in ((dict(source_ids=[1]), _lib.ERR_INVALID_ARGUMENT), (dict(target_ids=[-1]), _lib.ERR_INVALID_ARGUMENT)):
        with pytest.raises(pa.IcpError) as e:
            engine.align(np.eye(4)[None], p, **kwargs)
        assert e.value.status == code and "out of range" in str(e.value)
    bad = np.eye(4)
    bad[0, 3] = np.inf
    with pytest.raises(pa.IcpError) as e:
        engine.align(bad[None], p)
    assert e.value.status == _lib.ERR_NONFINITE_INPUT
    with pytest.raises(pa.IcpError):
        engine.align(np.eye(4)[None], pa.default_params(max_iterations=0))
    with pytest.raises(pa.IcpError):
        engine.set_targets([np.zeros((0, 3), np.float32)])          # an empty scene is an error ...
    engine.set_targets([tgt])
    engine.set_sources([np.zeros((0, 3), np.float32), src])           # ... an empty source slab is legal (rank with no points)
    res = engine.align(np.stack([np.eye(4), np.eye(4)]), p, source_ids=[0, 1], target_ids=[0, 0])
    assert res.status[0] == _lib.ERR_TOO_FEW_PAIRS and res.status[1] == 0 and res.iterations[1] == 3
    with pytest.raises(TypeError):
        engine.set_targets([tgt.astype(np.float64)])
    with pytest.raises(pa.IcpError):
        engine.scene_filter([(7, [])])                                 # unknown filter kind (and no scene loaded)
    # raw ABI: null context / null outputs
    assert _lib.lib.mi355_icp_correspond(None, 1, None, None, None, C.c_float(0), None, None, None) == _lib.ERR_INVALID_ARGUMENT
    assert _lib.lib.mi355_icp_get_profile(None, None) == _lib.ERR_INVALID_ARGUMENT
    # still healthy afterwards
    engine.set_sources([src])
    ok = engine.align(np.eye(4)[None], p)
    assert ok.status[0] == 0 and engine.last_error() == ""


def test_two_engines_share_the_device(engine):
    import pose_estimation_amd as pa
    src, tgt, gt = st.make_pair(st.synth_cloud_np, 800, 900)
    with pa.IcpEngine(0) as other:
        other.set_targets([tgt])
        other.set_sources([src])
        engine.set_targets([tgt[::-1].copy()])
        engine.set_sources([src])
        a = other.align(np.eye(4)[None], pa.default_params(max_iterations=4))
        b = engine.align(np.eye(4)[None], pa.default_params(max_iterations=4))
    assert a.status[0] == 0 and b.status[0] == 0
    assert np.linalg.norm(a.poses[0] - b.poses[0]) < 1e-12     # a permuted target gives the same sums (up to order)
    with pytest.raises(pa.IcpError):
        pa.IcpEngine(99)                                        # device id out of range

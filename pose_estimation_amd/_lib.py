"""ctypes binding of libmi355_icp.so (include/mi355_icp.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C pose_estimation_amd/csrc``.
There is no Python or CPU stand-in for it: if it is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_ICP_LIB: developer override used by tools/ to load the diagnostic build (make -C csrc diag)
LIB_PATH = os.environ.get("MI355_ICP_LIB") or os.path.join(_HERE, "lib", "libmi355_icp.so")

NSUMS = 17
UNIQUE_ID_BYTES = 128
FLAG_PROFILE = 1
FLAG_PRUNED_SCAN = 2
FLAG_UPSTREAM = 4
FLAG_BRUTE_SCAN = 8

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_NO_DEVICE = 2
ERR_HIP = 3
ERR_NONFINITE_INPUT = 4
ERR_TOO_FEW_PAIRS = 5
ERR_NUMERIC = 6
ERR_COMM = 7
ERR_INTERNAL = 8

STATUS_NAMES = {
    0: "OK", 1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "HIP", 4: "NONFINITE_INPUT",
    5: "TOO_FEW_PAIRS", 6: "NUMERIC", 7: "COMM", 8: "INTERNAL",
}


class Params(C.Structure):
    """mi355_icp_params"""
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("max_correspondence_distance", C.c_float),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("relative_mse_epsilon", C.c_double),
        ("flags", C.c_int32),
    ]


class ReferenceParams(C.Structure):
    """mi355_icp_reference_params"""
    _fields_ = [
        ("iterations", C.c_int32),
        ("tolerance", C.c_float),
        ("rejection_scale", C.c_float),
        ("num_levels", C.c_int32),
        ("flags", C.c_int32),
    ]


NREFSUMS = 29
NREFSUMS_EX = 30


class Profile(C.Structure):
    """mi355_icp_profile"""
    _fields_ = [
        ("nn_scan_ms", C.c_double),
        ("resolve_ms", C.c_double),
        ("reduce_ms", C.c_double),
        ("nn_scan_launches", C.c_uint64),
        ("resolve_launches", C.c_uint64),
        ("reduce_launches", C.c_uint64),
        ("pairs_evaluated", C.c_double),
        ("iterations", C.c_uint64),
        ("knn_scan_ms", C.c_double),
        ("normals_ms", C.c_double),
        ("normals_launches", C.c_uint64),
        ("overflow_points", C.c_uint64),
        ("scene_filter_ms", C.c_double),
        ("scene_filter_launches", C.c_uint64),
        ("scene_points_in", C.c_uint64),
        ("scene_points_out", C.c_uint64),
        ("overflow_host_passes", C.c_uint64),
    ]


class PlaneFitParams(C.Structure):
    """mi355_plane_fit_params"""
    _fields_ = [("distance_threshold", C.c_float), ("max_iterations", C.c_int32), ("probability", C.c_double),
                ("seed", C.c_uint64), ("optimize", C.c_int32), ("reserved", C.c_int32)]


class SceneFilter(C.Structure):
    """mi355_scene_filter"""
    _fields_ = [("kind", C.c_int32), ("p", C.c_float * 5)]


SCENE_FILTER_FINITE = 0
SCENE_FILTER_SPHERE_KEEP_INSIDE = 1
SCENE_FILTER_SPHERE_REMOVE_INSIDE = 2
SCENE_FILTER_PLANE_BAND_REMOVE = 3
SCENE_MAX_FILTERS = 8


# mi355_icp_allgather_fn: int (*)(void* user, const void* send, void* recv, size_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

# every symbol include/mi355_icp.h declares: name -> (restype, argtypes)
_P = C.POINTER
_ctx = C.c_void_p
SIGNATURES = {
    "mi355_icp_abi_version": (C.c_uint, []),
    "mi355_icp_create": (C.c_int, [C.c_int, _P(_ctx)]),
    "mi355_icp_create_multi": (C.c_int, [_P(C.c_int), C.c_int, _P(_ctx)]),
    "mi355_icp_destroy": (None, [_ctx]),
    "mi355_icp_last_error": (C.c_char_p, [_ctx]),
    "mi355_icp_default_params": (None, [_P(Params)]),
    "mi355_icp_params_from_reference": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_int, _P(Params)]),
    "mi355_icp_set_targets": (C.c_int, [_ctx, C.c_int, _P(C.c_void_p), _P(C.c_size_t), C.c_size_t]),
    "mi355_icp_set_sources": (C.c_int, [_ctx, C.c_int, _P(C.c_void_p), _P(C.c_size_t), C.c_size_t]),
    "mi355_icp_slab_range": (C.c_int, [C.c_size_t, C.c_int, C.c_int, _P(C.c_size_t), _P(C.c_size_t)]),
    "mi355_icp_level_slab": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _P(C.c_size_t), _P(C.c_size_t), _P(C.c_size_t),
                                       _P(C.c_size_t)]),
    "mi355_icp_align": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _P(Params), C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_icp_correspond": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mi355_icp_debug_filter_bounds": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_size_t)]),
    "mi355_icp_align_reference": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _P(ReferenceParams),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_icp_reference_pass": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_float), C.c_void_p]),
    "mi355_icp_reference_pass_ex": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_float), C.c_void_p]),
    "mi355_icp_upstream_normalisation": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _P(C.c_double)]),
    "mi355_icp_level_slab_step": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P(C.c_size_t), _P(C.c_size_t),
                                            _P(C.c_size_t), _P(C.c_size_t)]),
    "mi355_icp_upstream_step": (C.c_uint, [C.c_size_t, C.c_int]),
    "mi355_icp_solve_point_to_plane": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_icp_solve_point_to_plane_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mi355_icp_scene_load": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_size_t]),
    "mi355_icp_scene_filter": (C.c_int, [_ctx, C.c_int, C.c_void_p, _P(C.c_size_t)]),
    "mi355_icp_scene_download": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "mi355_icp_scene_fit_plane": (C.c_int, [_ctx, _P(PlaneFitParams), C.c_void_p, _P(C.c_size_t), _P(C.c_int)]),
    "mi355_icp_scene_normals": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_icp_scene_commit_target": (C.c_int, [_ctx]),
    "mi355_icp_solve_rigid": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_icp_host_update": (C.c_int, [_P(Params), C.c_void_p, C.c_int, C.c_void_p, _P(C.c_double), _P(C.c_int),
                                        _P(C.c_int)]),
    "mi355_icp_pose_to_pose7": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_icp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mi355_icp_comm_attach": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "mi355_icp_comm_attach_external": (C.c_int, [_ctx, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "mi355_icp_comm_detach": (C.c_int, [_ctx]),
    "mi355_icp_comm_info": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "mi355_icp_set_profiling": (C.c_int, [_ctx, C.c_int]),
    "mi355_icp_get_profile": (C.c_int, [_ctx, _P(Profile)]),
    "mi355_icp_reset_profile": (C.c_int, [_ctx]),
    "mi355_icp_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_size_t, _P(C.c_int), _P(C.c_size_t)]),
    "mi355_icp_synth_cloud": (C.c_int, [C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t,
                                        C.c_void_p]),
    "mi355_icp_synth_cloud6": (C.c_int, [C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t,
                                         C.c_void_p]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pose_estimation_amd/csrc`. There is no CPU fallback for this engine.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.mi355_icp_abi_version() != 1:
        raise ImportError("libmi355_icp.so ABI version mismatch")
    return lib


lib = _load()

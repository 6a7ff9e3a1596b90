"""ctypes binding of oracle/libicp_oracle.so — the CPU checker (oracle/icp_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libicp_oracle.so")
_SO = os.environ.get("MI355_ORACLE_LIB") or _SO  # tools/cpu_sanitize.sh points this at its instrumented build
NSUMS = 17


class OracleParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("max_correspondence_distance", C.c_float),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("relative_mse_epsilon", C.c_double),
        ("flags", C.c_int32),
    ]


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _DIR], stdout=subprocess.DEVNULL)
    lib = C.CDLL(_SO)
    vp = C.c_void_p
    lib.oracle_pose_to_f32.argtypes = [vp, vp]
    lib.oracle_transform.argtypes = [vp, vp, C.c_size_t, vp]
    lib.oracle_nn_brute.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_int]
    lib.oracle_kdtree_build.restype = vp
    lib.oracle_kdtree_build.argtypes = [vp, C.c_size_t]
    lib.oracle_kdtree_free.argtypes = [vp]
    lib.oracle_kdtree_query.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_int]
    lib.oracle_accumulate.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_float, vp]
    lib.oracle_solve_rigid.restype = C.c_int
    lib.oracle_solve_rigid.argtypes = [vp, vp]
    lib.oracle_host_update.restype = C.c_int
    lib.oracle_host_update.argtypes = [C.POINTER(OracleParams), vp, C.c_int, vp, C.POINTER(C.c_double),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.oracle_icp.restype = C.c_int
    lib.oracle_icp.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, C.POINTER(OracleParams), C.c_int, C.c_int, vp,
                               C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), vp, vp, vp, vp]
    lib.oracle_pose_to_pose7.argtypes = [vp, vp, C.c_int]
    lib.oracle_omp_max_threads.restype = C.c_int
    return lib


lib = _load()


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def params(max_iterations=50, max_correspondence_distance=0.0, transformation_epsilon=0.0, rotation_epsilon=1e-5,
           relative_mse_epsilon=0.0) -> OracleParams:
    return OracleParams(max_iterations, max_correspondence_distance, transformation_epsilon, rotation_epsilon,
                        relative_mse_epsilon, 0)


def params_from(p) -> OracleParams:
    """Copies a pose_estimation_amd Params (same layout) into an OracleParams."""
    return OracleParams(p.max_iterations, p.max_correspondence_distance, p.transformation_epsilon,
                        p.rotation_epsilon, p.relative_mse_epsilon, 0)


def pose_to_f32(pose) -> np.ndarray:
    p = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
    out = np.empty(12, dtype=np.float32)
    lib.oracle_pose_to_f32(p.ctypes.data, out.ctypes.data)
    return out


def transform(pose, src) -> np.ndarray:
    src = _f32(src)
    pf = pose_to_f32(pose)
    out = np.empty_like(src)
    lib.oracle_transform(pf.ctypes.data, src.ctypes.data, src.shape[0], out.ctypes.data)
    return out


def nn_brute(q, t, nthreads=1):
    q, t = _f32(q), _f32(t)
    idx = np.empty(q.shape[0], dtype=np.int32)
    d2 = np.empty(q.shape[0], dtype=np.float32)
    lib.oracle_nn_brute(q.ctypes.data, q.shape[0], t.ctypes.data, t.shape[0], idx.ctypes.data, d2.ctypes.data,
                        nthreads)
    return idx, d2


class KdTree:
    def __init__(self, t):
        self.t = _f32(t)
        self.h = lib.oracle_kdtree_build(self.t.ctypes.data, self.t.shape[0])

    def query(self, q, nthreads=1):
        q = _f32(q)
        idx = np.empty(q.shape[0], dtype=np.int32)
        d2 = np.empty(q.shape[0], dtype=np.float32)
        lib.oracle_kdtree_query(self.h, q.ctypes.data, q.shape[0], idx.ctypes.data, d2.ctypes.data, nthreads)
        return idx, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib.oracle_kdtree_free(self.h)
            self.h = None


def accumulate(src0, tgt, idx, d2, max_corr=0.0) -> np.ndarray:
    src0, tgt = _f32(src0), _f32(tgt)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    d2 = np.ascontiguousarray(d2, dtype=np.float32)
    sums = np.empty(NSUMS, dtype=np.float64)
    lib.oracle_accumulate(src0.ctypes.data, tgt.ctypes.data, idx.ctypes.data, d2.ctypes.data, src0.shape[0],
                          float(max_corr), sums.ctypes.data)
    return sums


def correspond(pose, src, tgt, max_corr=0.0, use_kdtree=True, nthreads=1):
    """One oracle correspondence pass: (idx, d2, sums)."""
    moved = transform(pose, src)
    idx, d2 = (KdTree(tgt).query(moved, nthreads) if use_kdtree else nn_brute(moved, tgt, nthreads))
    return idx, d2, accumulate(src, tgt, idx, d2, max_corr)


def solve_rigid(sums):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    out = np.empty(16, dtype=np.float64)
    st = lib.oracle_solve_rigid(sums.ctypes.data, out.ctypes.data)
    return st, out.reshape(4, 4)


def host_update(p: OracleParams, sums, iteration, pose, prev_mse):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    po = np.array(pose, dtype=np.float64).reshape(16).copy()
    mse = C.c_double(prev_mse)
    done, conv = C.c_int(0), C.c_int(0)
    st = lib.oracle_host_update(C.byref(p), sums.ctypes.data, iteration, po.ctypes.data, C.byref(mse), C.byref(done),
                                C.byref(conv))
    return st, po.reshape(4, 4), mse.value, bool(done.value), bool(conv.value)


def icp(src, tgt, init_pose, p: OracleParams, use_kdtree=True, nthreads=1, trace=False):
    src, tgt = _f32(src), _f32(tgt)
    init = np.ascontiguousarray(init_pose, dtype=np.float64).reshape(16)
    out = np.empty(16, dtype=np.float64)
    res, iters, conv = C.c_double(), C.c_int(), C.c_int()
    tr_s = np.zeros((p.max_iterations, NSUMS)) if trace else None
    tr_p = np.zeros((p.max_iterations, 16)) if trace else None
    li = np.empty(src.shape[0], dtype=np.int32)
    ld = np.empty(src.shape[0], dtype=np.float32)
    st = lib.oracle_icp(src.ctypes.data, src.shape[0], tgt.ctypes.data, tgt.shape[0], init.ctypes.data, C.byref(p),
                        int(use_kdtree), nthreads, out.ctypes.data, C.byref(res), C.byref(iters), C.byref(conv),
                        tr_s.ctypes.data if trace else None, tr_p.ctypes.data if trace else None, li.ctypes.data,
                        ld.ctypes.data)
    r = {"status": st, "pose": out.reshape(4, 4), "residual": res.value, "iterations": iters.value,
         "converged": conv.value, "last_idx": li, "last_d2": ld}
    if trace:
        r["trace_sums"] = tr_s[:max(iters.value, 1)]
        r["trace_pose"] = tr_p[:max(iters.value, 1)].reshape(-1, 4, 4)
    return r


def pose_to_pose7(pose, reference_packing=False):
    p = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
    out = np.zeros(7, dtype=np.float32)
    lib.oracle_pose_to_pose7(p.ctypes.data, out.ctypes.data, int(reference_packing))
    return out


def max_threads() -> int:
    """Threads the oracle's OpenMP loops should use: what this job owns, not what the host has (a GPU box shows all
    256 host cores to a job that owns a 16-core share; 256 threads there ran the 64-object test 10x slower)."""
    try:
        owned = len(os.sched_getaffinity(0))
    except AttributeError:
        owned = os.cpu_count() or 1
    return max(1, min(lib.oracle_omp_max_threads(), owned, 16))


# ---- reference mode (point-to-plane / MAD rejection / picky / pyramid): oracle/icp_oracle.c, second half ----
NREF = 29


class OracleRefParams(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("tolerance", C.c_float), ("rejection_scale", C.c_float),
                ("num_levels", C.c_int32)]


lib.oracle_select_rank.restype = C.c_float
lib.oracle_select_rank.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
lib.oracle_rejection_threshold.restype = C.c_float
lib.oracle_rejection_threshold.argtypes = [C.c_void_p, C.c_size_t, C.c_float]
lib.oracle_solve_point_to_plane.restype = C.c_int
lib.oracle_solve_point_to_plane.argtypes = [C.c_void_p, C.c_void_p]
lib.oracle_euler_pose.argtypes = [C.c_void_p, C.c_void_p]
lib.oracle_ref_pass.restype = C.c_size_t
lib.oracle_ref_pass.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
lib.oracle_icp_reference.restype = C.c_int
lib.oracle_icp_reference.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(OracleRefParams),
                                     C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]


def rejection_threshold(d2, scale) -> float:
    d2 = np.ascontiguousarray(d2, dtype=np.float32)
    return float(lib.oracle_rejection_threshold(d2.ctypes.data, d2.shape[0], float(scale)))


def solve_point_to_plane(sums):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    x = np.empty(6)
    st = lib.oracle_solve_point_to_plane(sums.ctypes.data, x.ctypes.data)
    return st, x


def euler_pose(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.empty(16)
    lib.oracle_euler_pose(x.ctypes.data, P.ctypes.data)
    return P.reshape(4, 4)


def ref_pass(src_xyz, tgt6, pose_start, pose_cur, rejection_scale, nthreads=1):
    """One reference-mode pass on level clouds: (n_selected, sums[29], idx, d2, selected mask, threshold)."""
    src = _f32(src_xyz)
    tgt6 = np.ascontiguousarray(tgt6, dtype=np.float32)
    assert tgt6.ndim == 2 and tgt6.shape[1] == 6
    ps = np.ascontiguousarray(pose_start, dtype=np.float64).reshape(16)
    pc = np.ascontiguousarray(pose_cur, dtype=np.float64).reshape(16)
    sums = np.empty(NREF)
    idx = np.empty(src.shape[0], dtype=np.int32)
    d2 = np.empty(src.shape[0], dtype=np.float32)
    sel = np.empty(src.shape[0], dtype=np.uint8)
    thr = C.c_float()
    n = lib.oracle_ref_pass(src.ctypes.data, src.shape[0], tgt6.ctypes.data, tgt6.shape[0], ps.ctypes.data, pc.ctypes.data,
                            float(rejection_scale), nthreads, sums.ctypes.data, idx.ctypes.data, d2.ctypes.data,
                            sel.ctypes.data, C.byref(thr))
    return int(n), sums, idx, d2, sel.astype(bool), thr.value


def icp_reference(src_xyz, tgt6, init_pose, iterations=250, tolerance=0.005, rejection_scale=2.5, num_levels=8, nthreads=1):
    src = _f32(np.asarray(src_xyz)[:, :3])
    tgt6 = np.ascontiguousarray(tgt6, dtype=np.float32)
    init = np.ascontiguousarray(init_pose, dtype=np.float64).reshape(16)
    rp = OracleRefParams(iterations, tolerance, rejection_scale, num_levels)
    out = np.empty(16)
    res, passes = C.c_double(), C.c_int()
    st = lib.oracle_icp_reference(src.ctypes.data, src.shape[0], tgt6.ctypes.data, tgt6.shape[0], init.ctypes.data,
                                  C.byref(rp), nthreads, out.ctypes.data, C.byref(res), C.byref(passes))
    return {"status": st, "pose": out.reshape(4, 4), "residual": res.value, "passes": passes.value}


# ---- reference mode, upstream-faithful variant (oracle/icp_oracle.c: "UPSTREAM-FAITHFUL variant") ----
NREFX = 30
lib.oracle_upstream_step.restype = C.c_uint32
lib.oracle_upstream_step.argtypes = [C.c_size_t, C.c_int]
lib.oracle_upstream_max_iterations.restype = C.c_int
lib.oracle_upstream_max_iterations.argtypes = [C.c_int, C.c_int]
lib.oracle_from_centred.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.oracle_upstream_normalisation.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_double)]
lib.oracle_ref_pass_ex.restype = C.c_size_t
lib.oracle_ref_pass_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_float,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
lib.oracle_icp_reference_upstream.restype = C.c_int
lib.oracle_icp_reference_upstream.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(OracleRefParams),
                                              C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p,
                                              C.POINTER(C.c_double), C.c_void_p]


def _f32x6(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 6
    return a


def upstream_step(n_model: int, level: int) -> int:
    """Sampling step of pyramid level `level` for a model of n_model rows (0: the level has no samples)."""
    return int(lib.oracle_upstream_step(n_model, level))


def upstream_max_iterations(iterations: int, level: int) -> int:
    return int(lib.oracle_upstream_max_iterations(iterations, level))


def from_centred(Xc, centre, T0) -> np.ndarray:
    Xc = np.ascontiguousarray(Xc, dtype=np.float64).reshape(16)
    T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(16)
    c = np.ascontiguousarray(centre, dtype=np.float32)
    out = np.empty(16)
    lib.oracle_from_centred(Xc.ctypes.data, c.ctypes.data, T0.ctypes.data, out.ctypes.data)
    return out.reshape(4, 4)


def upstream_normalisation(src6, tgt6, init_pose):
    """(centre float32[3], scale) of upstream's pre-normalisation for the model under init_pose."""
    src6, tgt6 = _f32x6(src6), _f32x6(tgt6)
    init = np.ascontiguousarray(init_pose, dtype=np.float64).reshape(16)
    c = np.empty(3, dtype=np.float32)
    s = C.c_double()
    lib.oracle_upstream_normalisation(src6.ctypes.data, src6.shape[0], 6, tgt6.ctypes.data, tgt6.shape[0], 6, init.ctypes.data,
                                      c.ctypes.data, C.byref(s))
    return c, s.value


def ref_pass_ex(src, tgt6, pose_start, pose_cur, rejection_scale, centre=None, nthreads=1):
    """One pass on level clouds, either variant: (n_selected, sums[30], idx, d2, selected mask, threshold).
    centre=None: the round 1-3 form (src may be N x 3 or N x 6; sums[29] = 0); centre given: upstream-faithful (src N x 6)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    assert src.ndim == 2 and src.shape[1] in (3, 6) and (centre is None or src.shape[1] == 6)
    tgt6 = _f32x6(tgt6)
    ps = np.ascontiguousarray(pose_start, dtype=np.float64).reshape(16)
    pc = np.ascontiguousarray(pose_cur, dtype=np.float64).reshape(16)
    c = None if centre is None else np.ascontiguousarray(centre, dtype=np.float32)
    sums = np.empty(NREFX)
    idx = np.empty(src.shape[0], dtype=np.int32)
    d2 = np.empty(src.shape[0], dtype=np.float32)
    sel = np.empty(src.shape[0], dtype=np.uint8)
    thr = C.c_float()
    n = lib.oracle_ref_pass_ex(src.ctypes.data, src.shape[0], src.shape[1], tgt6.ctypes.data, tgt6.shape[0], ps.ctypes.data,
                               pc.ctypes.data, float(rejection_scale), None if c is None else c.ctypes.data, nthreads,
                               sums.ctypes.data, idx.ctypes.data, d2.ctypes.data, sel.ctypes.data, C.byref(thr))
    return int(n), sums, idx, d2, sel.astype(bool), thr.value


def icp_reference_upstream(src6, tgt6, init_pose, iterations=250, tolerance=0.005, rejection_scale=2.5, num_levels=8, nthreads=1):
    src6, tgt6 = _f32x6(src6), _f32x6(tgt6)
    init = np.ascontiguousarray(init_pose, dtype=np.float64).reshape(16)
    rp = OracleRefParams(iterations, tolerance, rejection_scale, num_levels)
    out = np.empty(16)
    res, passes, scale = C.c_double(), C.c_int(), C.c_double()
    centre = np.empty(3, dtype=np.float32)
    per_level = np.zeros(num_levels, dtype=np.int32)
    st = lib.oracle_icp_reference_upstream(src6.ctypes.data, src6.shape[0], tgt6.ctypes.data, tgt6.shape[0], init.ctypes.data,
                                           C.byref(rp), nthreads, out.ctypes.data, C.byref(res), C.byref(passes),
                                           centre.ctypes.data, C.byref(scale), per_level.ctypes.data)
    return {"status": st, "pose": out.reshape(4, 4), "residual": res.value, "passes": passes.value, "centre": centre,
            "scale": scale.value, "level_passes": per_level}


# ---- scene preparation filters (reference pose_estimation.cpp:248, :310-333, :347-372) ----
class OracleSceneFilter(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_float * 5)]


lib.oracle_scene_filter_apply.restype = C.c_size_t
lib.oracle_scene_filter_apply.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]


def scene_filter(xyz, filters):
    """filters: list of (kind, [p0..p4]).  Returns the kept points in original order."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    arr = (OracleSceneFilter * max(len(filters), 1))()
    for k, (kind, p) in enumerate(filters):
        arr[k].kind = kind
        for j, v in enumerate(list(p) + [0.0] * (5 - len(p))):
            arr[k].p[j] = v
    out = np.empty_like(xyz)
    n = lib.oracle_scene_filter_apply(xyz.ctypes.data, xyz.shape[0], arr, len(filters), out.ctypes.data)
    return out[:n].copy()


# ---- scene normals (k-NN PCA), reference opencv_surface_match.cpp:57-59 ----
lib.oracle_knn_self.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
lib.oracle_normals.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
lib.oracle_smallest_eigvec3.argtypes = [C.c_void_p, C.c_void_p]


def knn_self(xyz, k, use_kdtree=True, nthreads=1):
    xyz = _f32(xyz)
    k = min(k, xyz.shape[0])
    idx = np.empty((xyz.shape[0], k), dtype=np.int32)
    d2 = np.empty((xyz.shape[0], k), dtype=np.float32)
    lib.oracle_knn_self(xyz.ctypes.data, xyz.shape[0], k, idx.ctypes.data, d2.ctypes.data, int(use_kdtree), nthreads)
    return idx, d2


def normals(xyz, k=20, flip=True, viewpoint=(0.0, 0.0, 0.0), use_kdtree=True, nthreads=1, want_idx=False):
    xyz = _f32(xyz)
    kk = min(k, xyz.shape[0])
    out = np.empty_like(xyz)
    vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
    idx = np.empty((xyz.shape[0], kk), dtype=np.int32) if want_idx else None
    lib.oracle_normals(xyz.ctypes.data, xyz.shape[0], k, int(flip), vp.ctypes.data, out.ctypes.data,
                       idx.ctypes.data if want_idx else None, int(use_kdtree), nthreads)
    return (out, idx) if want_idx else out


def smallest_eigvec3(Cm):
    Cm = np.ascontiguousarray(Cm, dtype=np.float64).reshape(9)
    v = np.empty(3)
    lib.oracle_smallest_eigvec3(Cm.ctypes.data, v.ctypes.data)
    return v


# ---- RANSAC plane fit (reference pose_estimation.cpp:285-297) ----
lib.oracle_plane_fit.restype = C.c_int
lib.oracle_plane_fit.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_double, C.c_uint64, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]


def plane_fit(xyz, distance_threshold=0.0001, max_iterations=100, probability=0.99, seed=0x5EEDC0DE, optimize=True):
    """(status, coefficients float32[4], inliers, iterations, hypothesis planes [max_iterations, 4], their support counts)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    assert xyz.ndim == 2 and xyz.shape[1] == 3
    out = np.zeros(4, dtype=np.float32)
    planes = np.zeros((max_iterations, 4), dtype=np.float32)
    counts = np.zeros(max_iterations, dtype=np.uint32)
    n, it = C.c_size_t(), C.c_int()
    st = lib.oracle_plane_fit(xyz.ctypes.data, xyz.shape[0], float(distance_threshold), max_iterations, float(probability), seed,
                              int(optimize), out.ctypes.data, C.byref(n), C.byref(it), planes.ctypes.data, counts.ctypes.data)
    return st, out, n.value, it.value, planes, counts

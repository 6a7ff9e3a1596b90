"""IcpEngine: Python handle on one mi355_icp_ctx (include/mi355_icp.h).

Thin by design: numpy arrays in, numpy arrays out, every call goes through the C-ABI.
The alignment arithmetic lives in csrc/ (HIP kernels + host Kabsch); nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import lib, Params, Profile, ReferenceParams, NSUMS, NREFSUMS, NREFSUMS_EX


class IcpError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"mi355_icp: {_lib.STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def default_params(**overrides) -> Params:
    p = Params()
    lib.mi355_icp_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown ICP parameter {k!r}")
        setattr(p, k, v)
    return p


def params_from_reference(iterations: int, tolerance: float = 0.05, rejection_scale: float = 2.5,
                          num_levels: int = 6) -> Params:
    """Maps cv::ppf_match_3d::ICP(iterations, tolerance, rejectionScale, numLevels)
    (reference opencv_surface_match.cpp:85) onto engine parameters."""
    p = Params()
    st = lib.mi355_icp_params_from_reference(iterations, tolerance, rejection_scale, num_levels, C.byref(p))
    if st != _lib.OK:
        raise IcpError(st, "params_from_reference")
    return p


def slab_range(n: int, rank: int, n_ranks: int) -> tuple[int, int]:
    b, c = C.c_size_t(), C.c_size_t()
    st = lib.mi355_icp_slab_range(n, rank, n_ranks, C.byref(b), C.byref(c))
    if st != _lib.OK:
        raise IcpError(st, "slab_range")
    return b.value, c.value


def level_slab(n_total: int, begin: int, count: int, level: int) -> tuple[int, int, int, int]:
    """(first_local, n_level_local, level_index_base, n_level_total) of a slab on a pyramid level (mi355_icp_level_slab)."""
    f, n, b, t = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = lib.mi355_icp_level_slab(n_total, begin, count, level, C.byref(f), C.byref(n), C.byref(b), C.byref(t))
    if st != _lib.OK:
        raise IcpError(st, "level_slab")
    return f.value, n.value, b.value, t.value


def level_slab_step(n_total: int, begin: int, count: int, step: int) -> tuple[int, int, int, int]:
    """The same for a level that keeps every step-th point (mi355_icp_level_slab_step)."""
    f, n, b, t = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = lib.mi355_icp_level_slab_step(n_total, begin, count, step, C.byref(f), C.byref(n), C.byref(b), C.byref(t))
    if st != _lib.OK:
        raise IcpError(st, "level_slab_step")
    return f.value, n.value, b.value, t.value


def upstream_step(n_model: int, level: int) -> int:
    """Upstream's sampling step of pyramid level `level` for a model of n_model rows (0: the level has no samples)."""
    return int(lib.mi355_icp_upstream_step(n_model, level))


def solve_rigid(sums: np.ndarray) -> np.ndarray:
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    assert sums.shape == (NSUMS,)
    out = np.empty(16, dtype=np.float64)
    st = lib.mi355_icp_solve_rigid(sums.ctypes.data, out.ctypes.data)
    if st != _lib.OK:
        raise IcpError(st, "solve_rigid")
    return out.reshape(4, 4)


def host_update(params: Params, sums: np.ndarray, iteration: int, pose: np.ndarray, prev_mse: float):
    """One step of the host iteration controller. Returns (status, new_pose, mse, done, converged)."""
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    p = np.array(pose, dtype=np.float64).reshape(16).copy()
    mse = C.c_double(prev_mse)
    done, conv = C.c_int(0), C.c_int(0)
    st = lib.mi355_icp_host_update(C.byref(params), sums.ctypes.data, iteration, p.ctypes.data, C.byref(mse),
                                   C.byref(done), C.byref(conv))
    return st, p.reshape(4, 4), mse.value, bool(done.value), bool(conv.value)


def solve_point_to_plane(sums: np.ndarray) -> np.ndarray:
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    assert sums.shape == (NREFSUMS,)
    x = np.empty(6, dtype=np.float64)
    st = lib.mi355_icp_solve_point_to_plane(sums.ctypes.data, x.ctypes.data)
    if st != _lib.OK:
        raise IcpError(st, "solve_point_to_plane")
    return x


def solve_point_to_plane_batch(sums: np.ndarray):
    """n systems at once (n x NREFSUMS) -> (x [n, 6], status [n]); every row gets the bits solve_point_to_plane gives it."""
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    assert sums.ndim == 2 and sums.shape[1] == NREFSUMS
    n = sums.shape[0]
    x = np.zeros((n, 6), dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    st = lib.mi355_icp_solve_point_to_plane_batch(sums.ctypes.data, n, x.ctypes.data, status.ctypes.data)
    if st != _lib.OK:
        raise IcpError(st, "solve_point_to_plane_batch")
    return x, status


def pose_to_pose7(pose: np.ndarray, reference_packing: bool = False) -> np.ndarray:
    p = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
    out = np.zeros(7, dtype=np.float32)
    st = lib.mi355_icp_pose_to_pose7(p.ctypes.data, out.ctypes.data, int(reference_packing))
    if st != _lib.OK:
        raise IcpError(st, "pose_to_pose7")
    return out


def synth_cloud(seed: int, count: int, first_index: int = 0, noise_seed: int = 0, sigma: float = 0.0,
                transform: Optional[np.ndarray] = None, normals: bool = False) -> np.ndarray:
    """Deterministic synthetic surface sample (csrc/synth.c), float32 [count, 3] (or [count, 6] with normals)."""
    out = np.empty((count, 6 if normals else 3), dtype=np.float32)
    t = None
    if transform is not None:
        t = np.ascontiguousarray(np.asarray(transform, dtype=np.float64)[:3, :4]).reshape(12)
    fn = lib.mi355_icp_synth_cloud6 if normals else lib.mi355_icp_synth_cloud
    st = fn(seed, noise_seed, float(sigma), t.ctypes.data if t is not None else None, first_index, count,
            out.ctypes.data)
    if st != _lib.OK:
        raise IcpError(st, "synth_cloud")
    return out


@dataclass
class ReferenceResult:
    poses: np.ndarray     # [n_jobs, 4, 4]
    residual: np.ndarray  # [n_jobs] smallest fval of the finest level that ran (cloud units; upstream's normalised units
                          # with FLAG_UPSTREAM)
    passes: np.ndarray    # [n_jobs] correspondence passes over all pyramid levels
    status: np.ndarray    # [n_jobs]


@dataclass
class AlignResult:
    poses: np.ndarray       # [n_jobs, 4, 4] float64
    residual: np.ndarray    # [n_jobs] RMS correspondence distance of the last pass
    iterations: np.ndarray  # [n_jobs] int32
    status: np.ndarray      # [n_jobs] int32 (0 = OK)
    converged: np.ndarray   # [n_jobs] int32


def _cloud_ptrs(clouds: Sequence[np.ndarray]):
    keep, ptrs, counts, stride = [], [], [], None
    for a in clouds:
        a = np.asarray(a)
        if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3:
            raise TypeError("clouds must be float32 arrays of shape [N, >=3] (xyz first)")
        if a.shape[0] and (a.strides[1] != 4 or a.strides[0] % 4 != 0 or a.strides[0] < 12):
            a = np.ascontiguousarray(a)
        s = a.strides[0] if a.shape[0] else 4 * a.shape[1]
        if stride is None:
            stride = s
        elif s != stride:
            a = np.ascontiguousarray(a[:, :3])
            if 12 != stride:
                raise TypeError("all clouds of one upload must share a row stride")
        keep.append(a)
        ptrs.append(a.ctypes.data if a.shape[0] else None)
        counts.append(a.shape[0])
    arr_p = (C.c_void_p * len(ptrs))(*ptrs)
    arr_n = (C.c_size_t * len(counts))(*counts)
    return keep, arr_p, arr_n, stride


class IcpEngine:
    """One engine per GPU - or, with ``device_ids=[...]``, ONE handle over several devices of this process
    (mi355_icp_create_multi: jobs or source slabs are spread over them inside the library).  Not thread-safe (the reference
    calls this path from one executor thread)."""

    def __init__(self, device_id: int = 0, device_ids: Optional[Sequence[int]] = None):
        h = C.c_void_p()
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            st = lib.mi355_icp_create_multi(ids, len(device_ids), C.byref(h))
            device_id = int(device_ids[0]) if len(device_ids) else 0
        else:
            st = lib.mi355_icp_create(device_id, C.byref(h))
        if st != _lib.OK:
            raise IcpError(st, lib.mi355_icp_last_error(None).decode())
        self._h = h
        self.device_id = device_id
        self.device_ids = None if device_ids is None else [int(d) for d in device_ids]
        self.n_targets = 0
        self.n_sources = 0
        self._source_sizes: list[int] = []

    def close(self):
        if getattr(self, "_h", None):
            lib.mi355_icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int):
        if st != _lib.OK:
            raise IcpError(st, lib.mi355_icp_last_error(self._h).decode())

    def last_error(self) -> str:
        return lib.mi355_icp_last_error(self._h).decode()

    # -- clouds
    def set_targets(self, clouds: Sequence[np.ndarray]):
        keep, p, n, stride = _cloud_ptrs(clouds)
        self._check(lib.mi355_icp_set_targets(self._h, len(keep), p, n, stride))
        self.n_targets = len(keep)

    def set_sources(self, clouds: Sequence[np.ndarray]):
        keep, p, n, stride = _cloud_ptrs(clouds)
        self._check(lib.mi355_icp_set_sources(self._h, len(keep), p, n, stride))
        self.n_sources = len(keep)
        self._source_sizes = [a.shape[0] for a in keep]

    # -- alignment
    @staticmethod
    def _ids(ids, n_jobs):
        a = np.zeros(n_jobs, dtype=np.int32) if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        assert a.shape == (n_jobs,)
        return a

    def align(self, init_poses: np.ndarray, params: Params, source_ids=None, target_ids=None) -> AlignResult:
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(-1, 16)
        n = poses.shape[0]
        sid, tid = self._ids(source_ids, n), self._ids(target_ids, n)
        out_p = np.empty((n, 16), dtype=np.float64)
        out_r = np.empty(n, dtype=np.float64)
        out_i = np.empty(n, dtype=np.int32)
        out_s = np.empty(n, dtype=np.int32)
        out_c = np.empty(n, dtype=np.int32)
        self._check(lib.mi355_icp_align(self._h, n, sid.ctypes.data, tid.ctypes.data, poses.ctypes.data,
                                        C.byref(params), out_p.ctypes.data, out_r.ctypes.data, out_i.ctypes.data,
                                        out_s.ctypes.data, out_c.ctypes.data))
        return AlignResult(out_p.reshape(n, 4, 4), out_r, out_i, out_s, out_c)

    def correspond(self, poses: np.ndarray, max_correspondence_distance: float = 0.0, source_ids=None,
                   target_ids=None, want_pairs: bool = True):
        """One correspondence pass. Returns (index int32 [sum Ns], dist2 float32 [sum Ns], sums [n_jobs, 17])."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        n = poses.shape[0]
        sid, tid = self._ids(source_ids, n), self._ids(target_ids, n)
        total = int(sum(self._source_sizes[s] for s in sid))
        idx = np.empty(total, dtype=np.int32) if want_pairs else None
        d2 = np.empty(total, dtype=np.float32) if want_pairs else None
        sums = np.empty((n, NSUMS), dtype=np.float64)
        self._check(lib.mi355_icp_correspond(self._h, n, sid.ctypes.data, tid.ctypes.data, poses.ctypes.data,
                                             float(max_correspondence_distance),
                                             idx.ctypes.data if want_pairs else None,
                                             d2.ctypes.data if want_pairs else None, sums.ctypes.data))
        note = self.last_error()
        if note:  # status OK with a note: some point had no finite distance and got the oracle's (0, +inf) - see the header
            import warnings
            warnings.warn(f"mi355_icp_correspond: {note}", RuntimeWarning, stacklevel=2)
        return idx, d2, sums

    def debug_filter_bounds(self, pose: np.ndarray, source_id: int = 0, target_id: int = 0):
        """Diagnostic (parity tests): (lower [n_runs, n_source] float32, run_targets [n_runs, 128] int32, -1 = padding) as
        the matrix-core filter scan computes them for every (source point, 128-target run)."""
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
        n = C.c_size_t(0)
        self._check(lib.mi355_icp_debug_filter_bounds(self._h, source_id, target_id, p.ctypes.data, None, None, C.byref(n)))
        ns = self._source_sizes[source_id]
        lower = np.empty((n.value, ns), dtype=np.float32)
        runs = np.empty((n.value, 128), dtype=np.int32)
        self._check(lib.mi355_icp_debug_filter_bounds(self._h, source_id, target_id, p.ctypes.data, lower.ctypes.data,
                                                      runs.ctypes.data, C.byref(n)))
        return lower, runs

    # -- reference mode (pyramid + MAD rejection + picky + point-to-plane)
    def align_reference(self, init_poses: np.ndarray, iterations: int = 250, tolerance: float = 0.005,
                        rejection_scale: float = 2.5, num_levels: int = 8, source_ids=None, target_ids=None,
                        flags: int = 0) -> ReferenceResult:
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(-1, 16)
        n = poses.shape[0]
        sid, tid = self._ids(source_ids, n), self._ids(target_ids, n)
        rp = ReferenceParams(iterations, tolerance, rejection_scale, num_levels, flags)
        out_p = np.empty((n, 16), dtype=np.float64)
        out_r = np.empty(n, dtype=np.float64)
        out_i = np.empty(n, dtype=np.int32)
        out_s = np.empty(n, dtype=np.int32)
        self._check(lib.mi355_icp_align_reference(self._h, n, sid.ctypes.data, tid.ctypes.data, poses.ctypes.data,
                                                  C.byref(rp), out_p.ctypes.data, out_r.ctypes.data, out_i.ctypes.data,
                                                  out_s.ctypes.data))
        return ReferenceResult(out_p.reshape(n, 4, 4), out_r, out_i, out_s)

    def reference_pass(self, pose_start: np.ndarray, pose_cur: np.ndarray, level: int = 0, rejection_scale: float = 2.5,
                       source_id: int = 0, target_id: int = 0):
        """One reference-mode pass: (index, dist2, selected mask, threshold, sums[29]) on pyramid level `level`."""
        n = self._source_sizes[source_id] >> level
        ps = np.ascontiguousarray(pose_start, dtype=np.float64).reshape(16)
        pc = np.ascontiguousarray(pose_cur, dtype=np.float64).reshape(16)
        idx = np.empty(n, dtype=np.int32)
        d2 = np.empty(n, dtype=np.float32)
        sel = np.empty(n, dtype=np.uint8)
        thr = C.c_float()
        sums = np.empty(NREFSUMS, dtype=np.float64)
        self._check(lib.mi355_icp_reference_pass(self._h, source_id, target_id, level, ps.ctypes.data, pc.ctypes.data,
                                                 float(rejection_scale), idx.ctypes.data, d2.ctypes.data, sel.ctypes.data,
                                                 C.byref(thr), sums.ctypes.data))
        return idx, d2, sel.astype(bool), thr.value, sums

    def reference_pass_ex(self, pose_start: np.ndarray, pose_cur: np.ndarray, step: int = 1, rejection_scale: float = 2.5,
                          centre=None, source_id: int = 0, target_id: int = 0):
        """One pass of either form on the level that keeps every step-th point: (index, dist2, selected mask, threshold,
        sums[30]).  centre=None: the round 1-3 form; centre (3 floats): the upstream-faithful form."""
        n = self._source_sizes[source_id] // step
        ps = np.ascontiguousarray(pose_start, dtype=np.float64).reshape(16)
        pc = np.ascontiguousarray(pose_cur, dtype=np.float64).reshape(16)
        cc = None if centre is None else np.ascontiguousarray(centre, dtype=np.float32).reshape(3)
        idx = np.empty(n, dtype=np.int32)
        d2 = np.empty(n, dtype=np.float32)
        sel = np.empty(n, dtype=np.uint8)
        thr = C.c_float()
        sums = np.empty(NREFSUMS_EX, dtype=np.float64)
        self._check(lib.mi355_icp_reference_pass_ex(self._h, source_id, target_id, step, ps.ctypes.data, pc.ctypes.data,
                                                    float(rejection_scale), None if cc is None else cc.ctypes.data,
                                                    idx.ctypes.data, d2.ctypes.data, sel.ctypes.data, C.byref(thr), sums.ctypes.data))
        return idx, d2, sel.astype(bool), thr.value, sums

    def upstream_normalisation(self, pose: np.ndarray, source_id: int = 0, target_id: int = 0):
        """(centre float32[3], scale) of upstream's pre-normalisation for the model under `pose` against the scene."""
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
        centre = np.empty(3, dtype=np.float32)
        scale = C.c_double()
        self._check(lib.mi355_icp_upstream_normalisation(self._h, source_id, target_id, p.ctypes.data, centre.ctypes.data, C.byref(scale)))
        return centre, scale.value

    # -- scene preparation (the reference's point filters as one-pass stream compaction)
    def scene_load(self, xyz: np.ndarray):
        a = np.asarray(xyz)
        if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3:
            raise TypeError("scene must be a float32 array of shape [N, >=3]")
        if a.shape[0] and (a.strides[1] != 4 or a.strides[0] % 4 != 0 or a.strides[0] < 12):
            a = np.ascontiguousarray(a)
        stride = a.strides[0] if a.shape[0] else 12
        self._check(lib.mi355_icp_scene_load(self._h, a.ctypes.data if a.shape[0] else None, a.shape[0], stride))

    def scene_filter(self, filters) -> int:
        """filters: sequence of (kind, params) with kind one of _lib.SCENE_FILTER_*; all applied in one pass."""
        arr = (_lib.SceneFilter * max(len(filters), 1))()
        for k, (kind, p) in enumerate(filters):
            arr[k].kind = int(kind)
            for j, v in enumerate(list(p) + [0.0] * (5 - len(p))):
                arr[k].p[j] = float(v)
        n = C.c_size_t()
        self._check(lib.mi355_icp_scene_filter(self._h, len(filters), arr, C.byref(n)))
        return n.value

    def scene_download(self) -> np.ndarray:
        n = C.c_size_t()
        self._check(lib.mi355_icp_scene_download(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.float32)
        self._check(lib.mi355_icp_scene_download(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def scene_fit_plane(self, distance_threshold: float = 0.0001, max_iterations: int = 100, probability: float = 0.99,
                        seed: int = 0x5EEDC0DE, optimize: bool = True):
        """RANSAC plane fit of the current scene with the reference's parameters by default (pose_estimation.cpp:285-297).
        Returns (coefficients float32[4], inliers, iterations)."""
        fp = _lib.PlaneFitParams(distance_threshold, max_iterations, probability, seed, int(optimize), 0)
        coeff = np.empty(4, dtype=np.float32)
        n, it = C.c_size_t(), C.c_int()
        self._check(lib.mi355_icp_scene_fit_plane(self._h, C.byref(fp), coeff.ctypes.data, C.byref(n), C.byref(it)))
        return coeff, n.value, it.value

    def scene_normals(self, k: int = 20, flip_viewpoint: bool = True, viewpoint=(0.0, 0.0, 0.0), want_neighbors: bool = False):
        """k-NN PCA normals of the current scene (computeNormalsPC3d semantics). Returns normals [n, 3] (and the
        neighbour indices [n, min(k, n)] when asked); the normals also stay on the device for scene_commit_target."""
        n = C.c_size_t()
        self._check(lib.mi355_icp_scene_download(self._h, None, 0, C.byref(n)))
        vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
        out = np.empty((n.value, 3), dtype=np.float32)
        nb = np.empty((n.value, min(k, n.value)), dtype=np.int32) if want_neighbors else None
        self._check(lib.mi355_icp_scene_normals(self._h, k, int(flip_viewpoint), vp.ctypes.data, out.ctypes.data,
                                                nb.ctypes.data if want_neighbors else None))
        return (out, nb) if want_neighbors else out

    def scene_commit_target(self):
        self._check(lib.mi355_icp_scene_commit_target(self._h))
        self.n_targets = 1

    # -- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES)()
        st = lib.mi355_icp_comm_unique_id(buf)
        if st != _lib.OK:
            raise IcpError(st, lib.mi355_icp_last_error(None).decode())
        return bytes(buf)

    def comm_attach(self, unique_id: bytes, rank: int, n_ranks: int):
        assert len(unique_id) == _lib.UNIQUE_ID_BYTES
        buf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(lib.mi355_icp_comm_attach(self._h, buf, rank, n_ranks))

    def comm_attach_external(self, rank: int, n_ranks: int, allgather):
        """Slab-sharded run over a transport the caller supplies: ``allgather(send: bytes-like of n bytes) -> bytes-like
        of n_ranks * n bytes`` (rank r's block at offset r * n).  The engine reduces everything itself, in rank order."""
        def thunk(_user, send, recv, nbytes):
            try:
                out = allgather((C.c_ubyte * nbytes).from_address(send))
                C.memmove(recv, bytes(out), nbytes * n_ranks)
                return 0
            except Exception:  # must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._ext_cb = _lib.ALLGATHER_FN(thunk)  # keep the trampoline alive while attached
        self._check(lib.mi355_icp_comm_attach_external(self._h, rank, n_ranks, self._ext_cb, None))

    def comm_detach(self):
        self._check(lib.mi355_icp_comm_detach(self._h))
        self._ext_cb = None

    def comm_info(self) -> tuple[int, int, int]:
        """(rank, n_ranks, kind) with kind 0 = none, 1 = RCCL (as the communicator reports), 2 = external."""
        r, n, k = C.c_int(), C.c_int(), C.c_int()
        self._check(lib.mi355_icp_comm_info(self._h, C.byref(r), C.byref(n), C.byref(k)))
        return r.value, n.value, k.value

    # -- measurement
    def set_profiling(self, enabled: bool):
        self._check(lib.mi355_icp_set_profiling(self._h, int(enabled)))

    def profile(self) -> Profile:
        p = Profile()
        self._check(lib.mi355_icp_get_profile(self._h, C.byref(p)))
        return p

    def reset_profile(self):
        self._check(lib.mi355_icp_reset_profile(self._h))

    def device_info(self) -> dict:
        name = C.create_string_buffer(128)
        cu, mem = C.c_int(), C.c_size_t()
        self._check(lib.mi355_icp_device_info(self._h, name, 128, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cu.value, "total_mem": mem.value}

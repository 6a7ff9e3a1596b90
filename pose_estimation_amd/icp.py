"""Host-side mirror of the reference's operator interface for this path.

The reference calls, in OpenCVSurfaceMatch::find_object_in_scene
(pose_estimation/src/opencv_surface_match.cpp:85-94):

    cv::ppf_match_3d::ICP icp(250, 0.005f, 2.5f, 8);
    icp.registerModelToScene(models_[object], pc_scene_normals, results_subset);

and then reads ``pose``, ``residual``, ``numVotes``, ``t`` and ``q`` off each Pose3D
(:102-142).  ``ICP`` and ``Pose3D`` here keep those names, argument meanings and the
in-place update behaviour, so code (and tests) written against the reference read the same;
the arithmetic runs on the GPU through IcpEngine (C-ABI, include/mi355_icp.h).

Two modes.  ``mode="reference"`` (the default: it is what the reference's call runs) keeps all four
constructor arguments live — resolution pyramid (``numLevels``), median+MAD rejection
(``rejectionScale``), picky dedupe and point-to-plane least squares on the scene normals (SURVEY.md
8f rank 1; numeric definition in oracle/icp_oracle.c) — and needs the scene as N x 6, which is what
the reference hands over (opencv_surface_match.cpp:57-59).  Within it ``upstream=True`` (the default,
MI355_ICP_FLAG_UPSTREAM) follows upstream's control flow as recalled — model normals in the stop
rule, pre-normalisation, cvRound sampling step — and needs the MODEL as N x 6 too, as the reference's
models are (opencv_surface_match.cpp:22); ``upstream=False`` is the form of rounds 1-3 (xyz-only
stop rule, step 2^l, residual in cloud units).  ``mode="point_to_point"`` must be asked
for by name: the engine BASELINE.json specifies (exact NN, Kabsch), which uses ``iterations`` and
``tolerence`` only.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .engine import IcpEngine, params_from_reference, pose_to_pose7


def _quat_to_rot(q: Sequence[float]) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float64)


def _rot_to_quat(R: np.ndarray) -> np.ndarray:
    """(w, x, y, z), w >= 0 branch-stable (Shepperd)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q, dtype=np.float64)
    return q / np.linalg.norm(q)


class Pose3D:
    """Mirror of cv::ppf_match_3d::Pose3D as the reference uses it: ``pose`` is the row-major
    4x4 model->scene transform, ``t`` its translation, ``q`` its rotation as (w, x, y, z),
    ``residual`` the ICP residual, ``numVotes`` the PPF vote count."""

    def __init__(self, alpha: float = 0.0, modelIndex: int = 0, numVotes: int = 0):
        self.alpha = float(alpha)
        self.modelIndex = int(modelIndex)
        self.numVotes = int(numVotes)
        self.residual = 0.0
        self.angle = 0.0
        self.pose = np.eye(4, dtype=np.float64)
        self.t = np.zeros(3, dtype=np.float64)
        self.q = np.array([1.0, 0.0, 0.0, 0.0], dtype=np.float64)

    def updatePose(self, pose: np.ndarray) -> None:
        pose = np.asarray(pose, dtype=np.float64).reshape(4, 4)
        self.pose = pose.copy()
        self.t = pose[:3, 3].copy()
        self.q = _rot_to_quat(pose[:3, :3])
        self.angle = float(2.0 * np.arccos(np.clip(self.q[0], -1.0, 1.0)))

    def updatePoseQuat(self, q: Sequence[float], t: Sequence[float]) -> None:
        """opencv_surface_match.cpp:82 (after the caller normalised q, :71-79)."""
        pose = np.eye(4, dtype=np.float64)
        pose[:3, :3] = _quat_to_rot(q)
        pose[:3, 3] = np.asarray(t, dtype=np.float64)
        self.pose = pose
        self.t = pose[:3, 3].copy()
        self.q = np.asarray(q, dtype=np.float64).copy()
        self.angle = float(2.0 * np.arccos(np.clip(self.q[0], -1.0, 1.0)))

    def appendPose(self, incremental: np.ndarray) -> None:
        self.updatePose(np.asarray(incremental, dtype=np.float64).reshape(4, 4) @ self.pose)

    def clone(self) -> "Pose3D":
        p = Pose3D(self.alpha, self.modelIndex, self.numVotes)
        p.residual = self.residual
        p.updatePose(self.pose)
        return p

    def printPose(self) -> None:
        print(f"\n-- Pose to Model Index {self.modelIndex}: NumVotes = {self.numVotes}, Residual = {self.residual}")
        print(self.pose)


class ICP:
    """Mirror of cv::ppf_match_3d::ICP.

    ``ICP(iterations, tolerence, rejectionScale, numLevels)`` — same positional meaning as the
    reference call site ``ICP icp(250, 0.005f, 2.5f, 8)``.  ``registerModelToScene(srcPC, dstPC,
    poses)`` refines every Pose3D of ``poses`` in place (pose and residual) and returns 0, like
    upstream.  srcPC (model) and dstPC (scene) are float32 [N, 3] or [N, 6] (xyz first).
    """

    def __init__(self, iterations: int = 250, tolerence: float = 0.05, rejectionScale: float = 2.5,
                 numLevels: int = 6, sampleType: int = 0, numMaxCorr: int = 1, device_id: int = 0,
                 engine: Optional[IcpEngine] = None, mode: str = "reference", pruned_scan: bool = False,
                 upstream: bool = True):
        if mode not in ("point_to_point", "reference"):
            raise ValueError("mode must be 'point_to_point' or 'reference'")
        self.mode = mode
        self.iterations = int(iterations)
        self.tolerence = float(tolerence)
        self.rejectionScale = float(rejectionScale)
        self.numLevels = int(numLevels)
        self.sampleType = int(sampleType)
        self.numMaxCorr = int(numMaxCorr)
        self.params = params_from_reference(self.iterations, self.tolerence, self.rejectionScale, self.numLevels)
        # not upstream API: answer the NN queries with the exact pruned scan (same poses; off by default, see the header)
        self.pruned_scan = bool(pruned_scan)
        if self.pruned_scan:
            self.params.flags |= _lib.FLAG_PRUNED_SCAN
        # reference mode: follow upstream where the round 1-3 form left its control flow (needs the model as N x 6)
        self.upstream = bool(upstream)
        self._engine = engine
        self._device_id = device_id

    @property
    def engine(self) -> IcpEngine:
        if self._engine is None:
            self._engine = IcpEngine(self._device_id)  # raises without a GPU: no CPU fallback
        return self._engine

    def registerModelToScene(self, srcPC: np.ndarray, dstPC: np.ndarray, poses: Optional[List[Pose3D]] = None):
        eng = self.engine
        eng.set_sources([np.asarray(srcPC)])
        eng.set_targets([np.asarray(dstPC)])
        return self.registerUploaded(poses)

    def registerUploaded(self, poses: Optional[List[Pose3D]] = None, source_id: int = 0, target_id: int = 0):
        """Same as registerModelToScene for clouds already resident on the GPU."""
        single = poses is None
        plist = [Pose3D()] if single else list(poses)
        if not plist:
            return 0
        init = np.stack([p.pose for p in plist])
        if self.mode == "reference":
            res = self.engine.align_reference(init, self.iterations, self.tolerence, self.rejectionScale, self.numLevels,
                                              source_ids=[source_id] * len(plist), target_ids=[target_id] * len(plist),
                                              flags=(_lib.FLAG_PRUNED_SCAN if self.pruned_scan else 0) |
                                                    (_lib.FLAG_UPSTREAM if self.upstream else 0))
        else:
            res = self.engine.align(init, self.params, source_ids=[source_id] * len(plist),
                                    target_ids=[target_id] * len(plist))
        for k, p in enumerate(plist):
            # a failed job leaves the candidate unrefined (the reference swallows the exception:
            # opencv_surface_match.cpp:144-147); res.poses[k] is then the last valid pose
            p.updatePose(res.poses[k])
            p.residual = float(res.residual[k])
        self.last_result = res
        if single:
            return 0, plist[0].residual, plist[0].pose
        return 0


def select_best_pose(results: Sequence[Pose3D], n_results_total: int) -> int:
    """Best-candidate rule of opencv_surface_match.cpp:102-123: most votes; when more than 5
    results came back from matching, the smallest residual among candidates with > 400 votes
    overrides as it is met."""
    max_votes, best_idx, min_res = 0, 0, 10000.0
    for i, r in enumerate(results):
        if r.numVotes > max_votes:
            max_votes = r.numVotes
            best_idx = i
        if n_results_total > 5:
            if r.residual < min_res and r.numVotes > 400:
                min_res = np.float32(r.residual)  # reference keeps min_res in a float
                best_idx = i
    return best_idx


def refine_candidates(icp: ICP, model: np.ndarray, scene: np.ndarray, results: List[Pose3D],
                      reference_packing: bool = True):
    """The ICP half of find_object_in_scene (opencv_surface_match.cpp:85-142): keep the first
    <= 6 candidates, refine them in one batched call, pick the best, pack 7 floats.
    Returns (pose7 float32[7], best_index, refined_subset)."""
    subset = list(results[:6])
    icp.registerModelToScene(model, scene, subset)
    best = select_best_pose(subset, len(results))
    # the reference packs from results[best_idx]; results_subset shares the same Pose3DPtr objects
    return pose_to_pose7(results[best].pose, reference_packing), best, subset

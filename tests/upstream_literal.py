"""A LITERAL numpy restatement of cv::ppf_match_3d::ICP::registerModelToScene as recalled from upstream's published
algorithm (opencv_contrib surface_matching; the source is absent from /root/reference and from this image, so nothing
here is copied and nothing can be checked offline - PARITY UNPINNED).

Purpose: test infrastructure only.  oracle/icp_oracle.c restates upstream's pre-normalisation ALGEBRAICALLY (clouds stay
as they are; centre and scale are carried through the formulas).  This module does what upstream does step by step -
it MATERIALISES the candidate-transformed, centred, scaled and level-start clouds in float32, runs the level loop from
the identity on them and un-normalises the pose at the end - so that tests/test_upstream_variant_cpu.py can show the two
agree up to the float32 roundings of those intermediate clouds.  It shares no code with the oracle (numpy + scipy cKDTree).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

EPS = 1.192092896e-07


def cv_round(x: float) -> int:
    return int(np.rint(x))  # half to even, like cvtsd2si


def transform_pc_pose(pc: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """Points under the 4x4 pose in double, rounded to float32; normals rotated, re-normalised, rounded."""
    pose = np.asarray(pose, dtype=np.float64)
    out = np.zeros_like(pc, dtype=np.float32)
    p = pc[:, :3].astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]
    out[:, :3] = p.astype(np.float32)
    if pc.shape[1] == 6:
        n2 = pc[:, 3:6].astype(np.float64) @ pose[:3, :3].T
        nn = np.linalg.norm(n2, axis=1)
        ok = nn > EPS
        out[ok, 3:6] = (n2[ok] / nn[ok, None]).astype(np.float32)
    return out


def sample_pc_uniform(pc: np.ndarray, step: int) -> np.ndarray:
    rows = pc.shape[0] // step
    return pc[::step][:rows].copy()


def rejection_threshold(d: np.ndarray, scale: float) -> np.float32:
    n = d.shape[0]
    med = np.sort(d)[(n - 1) // 2]
    dev = np.abs(d.astype(np.float64) - np.float64(med)).astype(np.float32)
    s = np.float32(1.48257968) * np.sort(dev)[(n - 1) // 2]
    return np.float32(np.float32(scale) * s + med)


def euler_to_pose(rpy, t) -> np.ndarray:
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(rpy[0]), -s(rpy[0])], [0, s(rpy[0]), c(rpy[0])]])
    Ry = np.array([[c(rpy[1]), 0, s(rpy[1])], [0, 1, 0], [-s(rpy[1]), 0, c(rpy[1])]])
    Rz = np.array([[c(rpy[2]), -s(rpy[2]), 0], [s(rpy[2]), c(rpy[2]), 0], [0, 0, 1]])
    P = np.eye(4)
    P[:3, :3] = Rx @ (Ry @ Rz)
    P[:3, 3] = t
    return P


def register_inner(src: np.ndarray, dst: np.ndarray, iterations: int, tolerance: float, rejection_scale: float, num_levels: int):
    """The single-pose registerModelToScene(srcPC, dstPC, residual, pose): returns (pose, residual, passes per level, scale, meanAvg)."""
    n = src.shape[0]
    src_t, dst_t = src.copy(), dst.copy()
    mean_src = src_t[:, :3].astype(np.float64).mean(axis=0)
    mean_dst = dst_t[:, :3].astype(np.float64).mean(axis=0)
    mean_avg = 0.5 * (mean_src + mean_dst)
    src_t[:, :3] -= mean_avg.astype(np.float32)
    dst_t[:, :3] -= mean_avg.astype(np.float32)

    def dist_to_origin(pc):
        x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
        return float(np.sqrt(x * x + y * y + z * z, dtype=np.float32).astype(np.float64).sum())

    scale = n / ((dist_to_origin(src_t) + dist_to_origin(dst_t)) * 0.5)
    src_t[:, :3] *= np.float32(scale)
    dst_t[:, :3] *= np.float32(scale)
    pose = np.eye(4)
    residual = 0.0
    level_passes = [0] * num_levels
    for level in range(num_levels - 1, -1, -1):
        num_samples = cv_round(n / 2.0 ** level)
        if num_samples < 1:
            continue  # upstream divides by zero here; the restatements skip the level
        tol_p = tolerance * (level + 1) * (level + 1)
        max_it = cv_round(iterations / (level + 1))
        step = cv_round(n / num_samples)
        src_pct = sample_pc_uniform(transform_pc_pose(src_t, pose), step)
        dst_pcs = sample_pc_uniform(dst_t, step)
        if src_pct.shape[0] < 6 or dst_pcs.shape[0] < 1:
            continue  # upstream: one search, fewer than six pairs, level left without a pose change
        tree = cKDTree(dst_pcs[:, :3].astype(np.float64))
        fval_old, fval_perc, fval_min = 9999999999.0, 0.0, 9999999999.0
        moved = src_pct.copy()
        pose_x = np.eye(4)
        i = 0
        while not (fval_perc < 1 + tol_p and fval_perc > 1 - tol_p) and i < max_it:
            dist, idx = tree.query(moved[:, :3].astype(np.float64))
            d2 = (dist * dist).astype(np.float32)
            level_passes[level] += 1
            keep = np.ones(d2.shape[0], dtype=bool)
            if rejection_scale > 0:
                keep = d2 < rejection_threshold(d2, rejection_scale)
            ii = np.nonzero(keep)[0]
            # picky: per scene point the kept model point with the smallest distance (lowest index on ties)
            order = ii[np.lexsort((ii, d2[ii], idx[ii]))]
            first = np.ones(order.shape[0], dtype=bool)
            first[1:] = idx[order][1:] != idx[order][:-1]
            model_i = order[first]
            scene_j = idx[model_i]
            if model_i.shape[0] < 6:
                break
            src_m = src_pct[model_i].astype(np.float64)
            dst_m = dst_pcs[scene_j].astype(np.float64)
            A = np.concatenate([np.cross(src_m[:, :3], dst_m[:, 3:6]), dst_m[:, 3:6]], axis=1)
            b = ((dst_m[:, :3] - src_m[:, :3]) * dst_m[:, 3:6]).sum(axis=1)
            x = np.linalg.lstsq(A, b, rcond=None)[0]
            if not np.isfinite(x).all():
                break
            pose_x = euler_to_pose(x[:3], x[3:])
            moved = transform_pc_pose(src_pct, pose_x)
            fval = float(np.linalg.norm(src_m - dst_m)) / moved.shape[0]
            fval_perc = fval / fval_old
            fval_old = fval
            fval_min = min(fval_min, fval)
            i += 1
        pose = pose_x @ pose
        residual = fval_min
    R, t = pose[:3, :3], pose[:3, 3]
    out = np.eye(4)
    out[:3, :3] = R
    out[:3, 3] = t / scale + mean_avg - R @ mean_avg
    return out, residual, level_passes, scale, mean_avg


def register_model_to_scene(src6: np.ndarray, dst6: np.ndarray, candidate_pose: np.ndarray, iterations=250, tolerance=0.005,
                            rejection_scale=2.5, num_levels=8):
    """The vector<Pose3DPtr> overload for one candidate: transform the model by the candidate pose, refine from the identity,
    append.  Returns dict(pose, residual, passes, level_passes, scale, centre)."""
    src6 = np.ascontiguousarray(src6, dtype=np.float32)
    dst6 = np.ascontiguousarray(dst6, dtype=np.float32)
    cand = np.asarray(candidate_pose, dtype=np.float64)
    src_t = transform_pc_pose(src6, cand)
    pose_icp, residual, level_passes, scale, mean_avg = register_inner(src_t, dst6, iterations, tolerance, rejection_scale, num_levels)
    return {"pose": pose_icp @ cand, "residual": residual, "passes": int(sum(level_passes)), "level_passes": level_passes,
            "scale": scale, "centre": mean_avg}

"""numpy twin of pose_estimation_amd/csrc/synth.c (test infrastructure).

Bit-identical by construction: both use only IEEE-exact double +,-,*,/,sqrt in the same
order; tests/test_synth.py asserts equality.  Also holds the named workload configurations
of SURVEY.md 8d so tests, fixtures and bench.py agree on seeds.
"""
from __future__ import annotations

import math

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)

TARGET_SEED = 0x5EED0001
SOURCE_SEED = 0x5EED0002
NOISE_SEED = 0x5EED0003
SIGMA = 1e-3


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(seed: int, ctr: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = _mix64(np.uint64(seed) + ctr * GOLDEN)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_cloud_np(seed, count, first_index=0, noise_seed=0, sigma=0.0, transform=None, normals=False) -> np.ndarray:
    i = np.arange(first_index, first_index + count, dtype=np.uint64)
    a = np.ones(count)
    b = np.zeros(count)
    c = np.zeros(count)
    r2 = np.ones(count)
    pending = np.ones(count, dtype=bool)
    with np.errstate(over="ignore"):
        for k in range(64):
            if not pending.any():
                break
            base = (i * np.uint64(64) + np.uint64(k)) * np.uint64(4)
            ta = 2.0 * _u01(seed, base) - 1.0
            tb = 2.0 * _u01(seed, base + np.uint64(1)) - 1.0
            tc = 2.0 * _u01(seed, base + np.uint64(2)) - 1.0
            t2 = (ta * ta + tb * tb) + tc * tc
            ok = pending & (t2 >= 0.01) & (t2 <= 1.0)
            a = np.where(ok, ta, a)
            b = np.where(ok, tb, b)
            c = np.where(ok, tc, c)
            r2 = np.where(ok, t2, r2)
            pending &= ~ok
        inv = 1.0 / np.sqrt(r2)
        ux, uy, uz = a * inv, b * inv, c * inv
        rad = 1.0 + 0.25 * (ux * uy)
        rad = rad + 0.15 * ((uz * uz) * ux)
        rad = rad + 0.10 * ((uy * uy) * uy)
        rad = rad - 0.08 * uz
        p = np.stack([rad * ux, rad * uy, rad * uz], axis=1)
        nrm = None
        if normals:
            g0 = 0.25 * uy + 0.15 * (uz * uz)
            g1 = 0.25 * ux + 0.30 * (uy * uy)
            g2 = 0.30 * (uz * ux) - 0.08
            gu = (g0 * ux + g1 * uy) + g2 * uz
            m0 = ux - (g0 - gu * ux) / rad
            m1 = uy - (g1 - gu * uy) / rad
            m2 = uz - (g2 - gu * uz) / rad
            ml = np.sqrt((m0 * m0 + m1 * m1) + m2 * m2)
            nrm = np.stack([m0 / ml, m1 / ml, m2 / ml], axis=1)
        if sigma > 0.0:
            sqrt3 = math.sqrt(3.0)
            for d in range(3):
                base = (i * np.uint64(4) + np.uint64(d)) * np.uint64(4)
                s = ((_u01(noise_seed, base) + _u01(noise_seed, base + np.uint64(1))) +
                     _u01(noise_seed, base + np.uint64(2))) + _u01(noise_seed, base + np.uint64(3))
                p[:, d] = p[:, d] + sigma * ((s - 2.0) * sqrt3)
    if transform is not None:
        T = np.asarray(transform, dtype=np.float64)[:3, :4]
        q = np.empty_like(p)
        for r in range(3):
            q[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
        p = q
        if nrm is not None:
            qn = np.empty_like(nrm)
            for r in range(3):
                qn[:, r] = (T[r, 0] * nrm[:, 0] + T[r, 1] * nrm[:, 1]) + T[r, 2] * nrm[:, 2]
            nrm = qn
    if nrm is not None:
        return np.concatenate([p, nrm], axis=1).astype(np.float32)
    return p.astype(np.float32)


def rigid_transform(angle_deg: float, axis=(1.0, 2.0, 3.0), translation=(0.02, -0.01, 0.03)) -> np.ndarray:
    """4x4 double: rotation by angle about axis (Rodrigues), then translation."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / math.sqrt(float(ax @ ax))
    th = math.radians(angle_deg)
    c, s = math.cos(th), math.sin(th)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) * c + s * K + (1 - c) * np.outer(ax, ax)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = translation
    return T


def perturbation(object_index: int = 0) -> np.ndarray:
    """Known SE(3) applied to the source: 5 deg about (1,2,3) for the single-object
    configs; config C4 object k uses 2 deg + 0.1 deg * k."""
    return rigid_transform(5.0) if object_index < 0 else rigid_transform(2.0 + 0.1 * object_index)


def make_pair(synth, n_src: int, n_tgt: int, object_index: int = -1, src_first: int = 0, src_count=None):
    """(source, target, ground-truth pose source->target) for one object.
    `synth` is a synth_cloud-compatible callable (the library's or synth_cloud_np).
    object_index -1 = the single-object configs (C1/C2/C3/C5); k >= 0 = object k of C4 (seeds + 16k)."""
    off = 0 if object_index < 0 else 16 * object_index
    T = perturbation(object_index)
    tgt = synth(TARGET_SEED + off, n_tgt)
    cnt = n_src if src_count is None else src_count
    src = synth(SOURCE_SEED + off, cnt, first_index=src_first, noise_seed=NOISE_SEED + off, sigma=SIGMA, transform=T)
    gt = np.linalg.inv(T)
    return src, tgt, gt

#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run here, in the build container).

PARITY UNPINNED: the reference (yumi-crew/pose_estimation) holds no tests, vectors or clouds for
this path and its ICP (opencv_contrib surface_matching) is absent from this image, so these
vectors are outputs of oracle/icp_oracle.c (the CPU restatement), each cross-checked here against
an independent implementation before it is written:
  * nearest neighbours against scipy.spatial.cKDTree (float64 distances: a mismatch is admissible
    only where the float32 distances tie),
  * Kabsch against numpy.linalg.svd,
  * quaternions against scipy.spatial.transform.Rotation.
Inputs come from the deterministic generator (csrc/synth.c == tests/synth_twin.py).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz|json
    python tests/golden/make_golden.py --check  # regenerates into a scratch directory and FAILS (exit 1) if anything differs
                                                # from the committed fixtures (tests/test_oracle.py runs this)
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE  # where the fixtures are written (--check: a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ora  # noqa: E402
import synth_twin as st  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_nn_vs_scipy(moved, tgt, idx, d2):
    dd, ii = cKDTree(tgt.astype(np.float64)).query(moved.astype(np.float64))
    bad = np.nonzero(ii != idx)[0]
    for i in bad:  # admissible only if the float32 distances tie
        alt = ora.nn_brute(moved[i:i + 1], tgt[ii[i]:ii[i] + 1])[1][0]
        assert alt == d2[i], f"NN mismatch vs scipy at {i}: {alt} != {d2[i]}"
    return len(bad)


def kabsch_numpy(sums):
    n = sums[0]
    sb, tb = sums[1:4] / n, sums[4:7] / n
    H = sums[7:16].reshape(3, 3) - np.outer(sums[1:4], sums[4:7]) / n
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = tb - R @ sb
    return P


def icp_fixture(name, src, tgt, gt, p, sample_stride=None, hash_iters=()):
    r = ora.icp(src, tgt, np.eye(4), p, use_kdtree=True, nthreads=ora.max_threads(), trace=True)
    assert r["status"] == 0
    out = {"src_sha": sha(src), "tgt_sha": sha(tgt), "n_src": int(src.shape[0]), "n_tgt": int(tgt.shape[0]),
           "iterations": int(r["iterations"]), "converged": int(r["converged"]), "residual": r["residual"],
           "idx_sha": {}}
    arrays = {"gt_pose": gt, "final_pose": r["pose"], "trace_sums": r["trace_sums"], "trace_pose": r["trace_pose"]}
    # every traced pose must be reproduced by numpy Kabsch from the traced sums
    for k in range(r["iterations"]):
        Pn = kabsch_numpy(r["trace_sums"][k])
        nxt = r["trace_pose"][k + 1] if k + 1 < r["iterations"] else r["pose"]
        assert np.abs(Pn - nxt).max() < 1e-9, (name, k, np.abs(Pn - nxt).max())
    for it in sorted(set(list(hash_iters) + [0])):
        if it >= r["iterations"]:
            continue
        idx, d2, sums = ora.correspond(r["trace_pose"][it], src, tgt, nthreads=ora.max_threads())
        assert np.array_equal(sums, r["trace_sums"][it])
        nbad = check_nn_vs_scipy(ora.transform(r["trace_pose"][it], src), tgt, idx, d2)
        out["idx_sha"][str(it)] = {"idx": sha(idx), "d2": sha(d2), "scipy_tie_mismatches": int(nbad)}
        if it == 0:
            if sample_stride:
                arrays["idx0_sample"] = idx[::sample_stride]
                arrays["d2_0_sample"] = d2[::sample_stride]
                out["sample_stride"] = sample_stride
            else:
                arrays["idx0"] = idx
                arrays["d2_0"] = d2
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    return out


def main():
    meta = {"note": "oracle outputs; parity unpinned by the reference (see module docstring)"}
    synth = st.synth_cloud_np

    # (1) C1 1k<->1k ("bunny stand-in"), 50 iterations, full vectors; clouds stored too
    src, tgt, gt = st.make_pair(synth, 1000, 1000)
    p = ora.params(max_iterations=50)
    meta["c1_1k"] = icp_fixture("c1_1k", src, tgt, gt, p, hash_iters=(0, 1, 49))
    np.savez_compressed(os.path.join(OUT, "c1_1k_clouds.npz"), src=src, tgt=tgt)

    # (2) C2 100k<->100k, 50 iterations: hashes + strided sample
    src, tgt, gt = st.make_pair(synth, 100_000, 100_000)
    meta["c2_100k"] = icp_fixture("c2_100k", src, tgt, gt, p, sample_stride=100, hash_iters=(0, 1, 49))

    # (3) ragged sizes: Ns, Nt not multiples of 64 / 1024 / 32, Nt < one tile; convergence criteria on
    src, tgt, gt = st.make_pair(synth, 1337, 777)
    pc = ora.params(max_iterations=60, transformation_epsilon=1e-10, rotation_epsilon=1e-9, relative_mse_epsilon=1e-7)
    meta["ragged"] = icp_fixture("ragged", src, tgt, gt, pc, hash_iters=(0,))
    np.savez_compressed(os.path.join(OUT, "ragged_clouds.npz"), src=src, tgt=tgt)

    # (4) tie fixture: integer lattice, queries on cell centres -> many exact float32 ties
    g = np.arange(12, dtype=np.float32)
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(7)
    tgt = tgt[rng.permutation(tgt.shape[0])]
    q = (rng.integers(0, 11, size=(900, 3)).astype(np.float32) + np.float32(0.5))
    q[::3, 0] -= 0.5  # some on faces/edges: 4- and 2-way ties
    idx, d2 = ora.nn_brute(q, tgt)
    idx_k, d2_k = ora.KdTree(tgt).query(q)
    assert np.array_equal(idx, idx_k) and np.array_equal(d2, d2_k)
    # lowest-index rule, checked independently in float64 (exact for these half-integers)
    D = ((q[:, None, :].astype(np.float64) - tgt[None, :, :].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(idx, D.argmin(1).astype(np.int32))
    ties = int(((D == D.min(1, keepdims=True)).sum(1) > 1).sum())
    np.savez_compressed(os.path.join(OUT, "ties.npz"), q=q, tgt=tgt, idx=idx, d2=d2)
    meta["ties"] = {"queries_with_ties": ties, "n": int(q.shape[0])}

    # (5) degenerate clouds for the Kabsch reflection / rank branches
    rng = np.random.default_rng(11)
    Rtrue = Rotation.from_rotvec([0.1, -0.2, 0.15]).as_matrix()
    ttrue = np.array([0.3, -0.1, 0.2])
    deg = {}
    plane = np.c_[rng.uniform(-1, 1, (400, 2)), np.zeros(400)].astype(np.float32)
    line = (np.outer(rng.uniform(-1, 1, 400), [1.0, 2.0, -1.0])).astype(np.float32)
    for nm, pts in (("coplanar", plane), ("collinear", line)):
        moved = (pts.astype(np.float64) @ Rtrue.T + ttrue).astype(np.float32)
        sums = ora.accumulate(pts, moved, np.arange(pts.shape[0], dtype=np.int32), np.zeros(pts.shape[0], np.float32))
        stt, P = ora.solve_rigid(sums)
        assert stt == 0 and abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12
        resid = np.abs(pts.astype(np.float64) @ P[:3, :3].T + P[:3, 3] - moved).max()
        assert resid < 1e-5, (nm, resid)
        if nm == "coplanar":
            assert np.abs(P - kabsch_numpy(sums)).max() < 1e-9
        deg[nm + "_src"], deg[nm + "_dst"], deg[nm + "_sums"], deg[nm + "_pose"] = pts, moved, sums, P
    # mirrored plane: the unconstrained optimum is a reflection; Kabsch must return the best proper rotation
    mirrored = plane.copy()
    mirrored[:, 0] = -mirrored[:, 0]
    sums = ora.accumulate(plane, mirrored, np.arange(400, dtype=np.int32), np.zeros(400, np.float32))
    stt, P = ora.solve_rigid(sums)
    assert stt == 0 and abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12 and np.abs(P - kabsch_numpy(sums)).max() < 1e-9
    deg["mirror_sums"], deg["mirror_pose"] = sums, P
    np.savez_compressed(os.path.join(OUT, "degenerate.npz"), **deg)

    # (6) pose7 known answers (reference formula chessboard_pose_estimator.cpp:196-207), incl. w ~ 0
    rots = [Rotation.identity(), Rotation.from_euler("z", 90, degrees=True), Rotation.from_euler("x", 180, degrees=True),
            Rotation.from_rotvec([0.3, -0.5, 0.2]), Rotation.from_rotvec(np.array([1.0, 2.0, 3.0]) / np.sqrt(14) * 3.1),
            Rotation.from_euler("y", 179.9, degrees=True)]
    poses, p7, p7ref = [], [], []
    for k, Rm in enumerate(rots):
        P = np.eye(4)
        P[:3, :3] = Rm.as_matrix()
        P[:3, 3] = [0.1 * k, -0.2, 0.3 + k]
        a = ora.pose_to_pose7(P, False)
        b = ora.pose_to_pose7(P, True)
        q = Rm.as_quat()  # x,y,z,w
        if q[3] < 0:
            q = -q
        assert np.abs(np.abs(a[3:]) - np.abs(q)).max() < 2e-3, (k, a, q)  # float formula loses digits near w=0
        assert b[5] == 0.0 and b[6] == a[6] and np.array_equal(b[:5], a[:5])
        poses.append(P), p7.append(a), p7ref.append(b)
    np.savez_compressed(os.path.join(OUT, "pose7.npz"), poses=np.array(poses), pose7=np.array(p7), pose7_ref=np.array(p7ref))

    # (7) reference mode (pyramid + MAD rejection + picky + point-to-plane), small enough to store the clouds
    T = st.perturbation(-1)
    tgt6 = st.synth_cloud_np(st.TARGET_SEED, 2500, normals=True)
    srcr = st.synth_cloud_np(st.SOURCE_SEED, 2000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)
    r = ora.icp_reference(srcr, tgt6, np.eye(4), 60, 0.005, 2.5, 5)
    assert r["status"] == 0 and np.linalg.norm(r["pose"] - np.linalg.inv(T)) < 5e-3
    start = st.rigid_transform(0.7, (0.1, 1, 0.3), (0.004, 0.001, -0.002))
    cur = st.rigid_transform(0.2, (1, 0, 0), (0.001, 0, 0)) @ start
    n_sel, sums, idx, d2, sel, thr = ora.ref_pass(srcr[::2][:1000], tgt6[::2][:1250], start, cur, 2.5)
    # independent check of the pass: NN vs scipy, threshold vs numpy, least squares vs numpy.linalg.lstsq
    check_nn_vs_scipy(ora.transform(cur, srcr[::2][:1000]), np.ascontiguousarray(tgt6[::2][:1250, :3]), idx, d2)
    med = np.sort(d2)[(len(d2) - 1) // 2]
    mad = np.sort(np.abs(d2.astype(np.float64) - np.float64(med)).astype(np.float32))[(len(d2) - 1) // 2]
    assert thr == np.float32(2.5) * (np.float32(1.48257968) * mad) + med
    pp = ora.transform(start, srcr[::2][:1000])[sel].astype(np.float64)
    qq = tgt6[::2][:1250][idx[sel], :3].astype(np.float64)
    nn = tgt6[::2][:1250][idx[sel], 3:].astype(np.float64)
    A = np.concatenate([np.cross(pp, nn), nn], axis=1)
    x_np = np.linalg.lstsq(A, ((qq - pp) * nn).sum(1), rcond=None)[0]
    assert np.abs(ora.solve_point_to_plane(sums)[1] - x_np).max() < 1e-9
    np.savez_compressed(os.path.join(OUT, "refmode_2k.npz"), src=srcr, tgt6=tgt6, final_pose=r["pose"], start=start, cur=cur,
                        pass_idx=idx, pass_d2=d2, pass_sel=sel, pass_thr=np.float32(thr), pass_sums=sums)
    meta["refmode_2k"] = {"params": [60, 0.005, 2.5, 5], "passes": int(r["passes"]), "residual": r["residual"],
                          "pass_level": 1, "pass_selected": int(n_sel)}

    # (8) reference mode, UPSTREAM-FAITHFUL variant (MI355_ICP_FLAG_UPSTREAM): a 1000-point N x 6 model - its pyramid samples
    # with steps 125, 62, 32, 16, ... (cvRound of the model size, not 2^l) - against a 3000-point scene a metre from the camera.
    # Independent check before writing: the literal numpy restatement that materialises upstream's clouds (upstream_literal.py)
    import upstream_literal as ul
    shift = np.array([0.0, 0.0, 1.0])
    tgt_u = st.synth_cloud_np(st.TARGET_SEED, 3000, normals=True)
    tgt_u[:, :3] += shift.astype(np.float32)
    src_u = st.synth_cloud_np(st.SOURCE_SEED, 1000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)
    init_u = np.eye(4)
    init_u[:3, 3] = shift
    init_u = st.rigid_transform(1.5, (0.2, 1, -0.4), (0.004, -0.003, 0.002)) @ init_u
    ru = ora.icp_reference_upstream(src_u, tgt_u, init_u, 250, 0.005, 2.5, 8)
    lit = ul.register_model_to_scene(src_u, tgt_u, init_u, 250, 0.005, 2.5, 8)
    assert ru["status"] == 0 and list(ru["level_passes"]) == list(lit["level_passes"])
    assert np.linalg.norm(ru["pose"] - lit["pose"]) <= 2e-6 and abs(ru["residual"] - lit["residual"]) <= 1e-6 * lit["residual"]
    steps = [ora.upstream_step(1000, l) for l in range(8)]
    assert steps == [1, 2, 4, 8, 16, 32, 62, 125]
    step_u = 62
    start_u = st.rigid_transform(0.7, (0.1, 1, 0.3), (0.004, 0.001, -0.002)) @ init_u
    cur_u = st.rigid_transform(0.2, (1, 0, 0), (0.001, 0, 0)) @ start_u
    n_u, sums_u, idx_u, d2_u, sel_u, thr_u = ora.ref_pass_ex(src_u[::step_u][:1000 // step_u], tgt_u[::step_u][:3000 // step_u], start_u, cur_u, 2.5,
                                                               centre=ru["centre"])
    np.savez_compressed(os.path.join(OUT, "refmode_upstream_1k.npz"), src6=src_u, tgt6=tgt_u, init=init_u, final_pose=ru["pose"],
                        centre=ru["centre"], level_passes=ru["level_passes"], start=start_u, cur=cur_u, pass_idx=idx_u, pass_d2=d2_u,
                        pass_sel=sel_u, pass_thr=np.float32(thr_u), pass_sums=sums_u)
    meta["refmode_upstream_1k"] = {"params": [250, 0.005, 2.5, 8], "passes": int(ru["passes"]), "residual": ru["residual"], "scale": ru["scale"],
                                   "steps": steps, "pass_step": step_u, "pass_selected": int(n_u)}

    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True)[:1500])


def compare_with_committed(fresh_dir):
    """Every array of every regenerated .npz and the whole golden.json against the committed files; returns the differences."""
    diffs = []
    for name in sorted(os.listdir(fresh_dir)):
        a, b = os.path.join(fresh_dir, name), os.path.join(HERE, name)
        if not os.path.exists(b):
            diffs.append(f"{name}: not committed")
            continue
        if name.endswith(".json"):
            if json.load(open(a)) != json.load(open(b)):
                diffs.append(f"{name}: contents differ")
            continue
        fa, fb = np.load(a), np.load(b)
        if sorted(fa.files) != sorted(fb.files):
            diffs.append(f"{name}: arrays {sorted(fa.files)} vs committed {sorted(fb.files)}")
            continue
        for k in fa.files:
            x, y = fa[k], fb[k]
            if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                diffs.append(f"{name}[{k}]: differs from the committed array")
    for name in sorted(os.listdir(HERE)):
        if name.endswith((".npz", ".json")) and not os.path.exists(os.path.join(fresh_dir, name)):
            diffs.append(f"{name}: committed but no longer generated")
    return diffs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="regenerate into a scratch directory and fail if anything differs from the committed fixtures")
    args = ap.parse_args()
    if args.check:
        with tempfile.TemporaryDirectory() as tmp:
            OUT = tmp
            main()
            diffs = compare_with_committed(tmp)
        if diffs:
            print("make_golden.py --check: the regenerated fixtures differ from the committed ones:\n  " + "\n  ".join(diffs), file=sys.stderr)
            sys.exit(1)
        print("make_golden.py --check: regenerated fixtures are identical to the committed ones")
    else:
        main()

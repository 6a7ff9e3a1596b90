"""Randomised differential test of reference mode (mi355_icp_align_reference) against the oracle: random cloud sizes down to
a handful of points (levels that are skipped, levels that end early), random pyramid depths, iteration budgets, tolerances,
rejection on and off, batches that mix clouds, random initial poses; surfaces with their normals and structureless Gaussian
clouds with random unit normals.  Same seeds every run."""
import os

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("MI355_ICP_REFRANDOM_CASES", "40"))


def random_cloud6(rng, n, kind):
    if kind == 0:  # the benchmark surface
        return st.synth_cloud_np(int(rng.integers(1, 1 << 30)), n, normals=True)
    p = rng.standard_normal((n, 3)).astype(np.float32) * np.float32(rng.uniform(0.05, 2.0))
    if kind == 2:  # flattened: nearly planar, the 6x6 system is close to singular
        p[:, 2] *= np.float32(1e-3)
    nrm = rng.standard_normal((n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.ascontiguousarray(np.hstack([p, nrm.astype(np.float32)]))


@pytest.mark.parametrize("upstream", [False, True], ids=["round1to3_form", "upstream_form"])
def test_randomised_reference_mode_against_the_oracle(engine, upstream):
    """upstream: MI355_ICP_FLAG_UPSTREAM against oracle_icp_reference_upstream - the models carry normals, the sizes below
    16,385 points sample with steps that are not powers of two, mixed batches run their levels in groups by step."""
    rng = np.random.default_rng(20261004)
    worst = 0.0
    for case in range(CASES):
        n_clouds = int(rng.integers(1, 4))
        kind = int(rng.integers(0, 3))
        tgts, srcs, gts = [], [], []
        for _ in range(n_clouds):
            nt = int(rng.choice([1, 7, 40, 333, 1024, 1025, 5000, 20000]))
            ns = int(rng.choice([6, 9, 47, 500, 2048, 2049, 7777]))
            tgt = random_cloud6(rng, nt, kind)
            T = st.rigid_transform(float(rng.uniform(0.0, 4.0)), rng.standard_normal(3), rng.standard_normal(3) * 0.01)
            pick = rng.integers(0, nt, ns)
            base = tgt[pick, :3].astype(np.float64) + rng.standard_normal((ns, 3)) * 1e-3
            src = ((base - T[:3, 3]) @ T[:3, :3]).astype(np.float32)  # T maps src onto the scene, up to the noise
            if upstream:  # N x 6 model: the scene normals of the picked points, taken back to the model frame
                src = np.hstack([src, (tgt[pick, 3:].astype(np.float64) @ T[:3, :3]).astype(np.float32)])
            tgts.append(tgt)
            srcs.append(np.ascontiguousarray(src))
            gts.append(T)
        engine.set_targets(tgts)
        engine.set_sources(srcs)
        n_jobs = int(rng.integers(1, 5))
        ids = [int(rng.integers(0, n_clouds)) for _ in range(n_jobs)]
        inits = np.stack([st.rigid_transform(float(rng.uniform(0, 2.0)), rng.standard_normal(3), rng.standard_normal(3) * 0.005) @ gts[i] for i in ids])
        iters = int(rng.choice([1, 3, 20, 250]))
        tol = float(rng.choice([0.005, 0.05, 1e-6]))
        scale = float(rng.choice([2.5, 0.0, 1.0]))
        levels = int(rng.integers(1, 11))
        res = engine.align_reference(inits, iters, tol, scale, levels, source_ids=ids, target_ids=ids, flags=4 if upstream else 0)
        for k, i in enumerate(ids):
            o = (ora.icp_reference_upstream if upstream else ora.icp_reference)(srcs[i], tgts[i], inits[k], iters, tol, scale, levels, nthreads=4)
            tag = f"case {case} job {k}: clouds {srcs[i].shape[0]} x {tgts[i].shape[0]}, kind {kind}, ({iters}, {tol}, {scale}, {levels})"
            assert res.status[k] == 0, tag
            assert res.passes[k] == o["passes"], f"{tag}: passes {res.passes[k]} vs oracle {o['passes']}"
            err = float(np.linalg.norm(res.poses[k] - o["pose"]))
            worst = max(worst, err)
            assert err <= 1e-6 * max(1.0, float(np.abs(o["pose"]).max())), f"{tag}: pose differs by {err}"
            assert abs(res.residual[k] - o["residual"]) <= 1e-9 * max(1.0, abs(o["residual"])), tag
    print(f"{CASES} random reference-mode cases ({'upstream' if upstream else 'round 1-3'} form), worst pose difference to the oracle {worst:.2e}")


def test_randomised_point_to_point_loop_against_the_oracle(engine):
    """The same for mi355_icp_align: random sizes, correspondence gates that reject most pairs or all of them, every
    combination of the convergence criteria, mixed batches.  Clouds start at 63 x 64 points: with a handful of points several
    sources hit one target, the matched targets are collinear, the cross-covariance has rank 1 and the rotation about that line
    is arbitrary - the oracle (eigen-decomposition of H^T H) and the product (one-sided Jacobi SVD of H) then both return a
    minimiser, but not the same one (seen at 3 x 5 points: 0.17 apart).  No rigid-transform solver pins that case."""
    import pose_estimation_amd as pa
    rng = np.random.default_rng(4711)
    worst = 0.0
    for case in range(CASES):
        n_clouds = int(rng.integers(1, 4))
        kind = int(rng.integers(0, 3))
        tgts, srcs, gts = [], [], []
        for _ in range(n_clouds):
            nt = int(rng.choice([64, 100, 1023, 1024, 3000, 17000]))
            ns = int(rng.choice([63, 64, 65, 511, 512, 2047, 2048, 2049, 6000]))
            tgt = random_cloud6(rng, nt, kind)[:, :3].copy()
            T = st.rigid_transform(float(rng.uniform(0.0, 3.0)), rng.standard_normal(3), rng.standard_normal(3) * 0.01)
            base = tgt[rng.integers(0, nt, ns)].astype(np.float64) + rng.standard_normal((ns, 3)) * 1e-3
            srcs.append(np.ascontiguousarray(((base - T[:3, 3]) @ T[:3, :3]).astype(np.float32)))
            tgts.append(tgt)
            gts.append(T)
        engine.set_targets(tgts)
        engine.set_sources(srcs)
        n_jobs = int(rng.integers(1, 6))
        ids = [int(rng.integers(0, n_clouds)) for _ in range(n_jobs)]
        inits = np.stack([st.rigid_transform(float(rng.uniform(0, 1.0)), rng.standard_normal(3), rng.standard_normal(3) * 0.003) @ gts[i] for i in ids])
        p = pa.default_params(max_iterations=int(rng.choice([1, 2, 7, 30])),
                              # (a tight gate on a small cloud leaves a handful of near-collinear pairs: ill-posed as above)
                              max_correspondence_distance=float(rng.choice([0.0, 0.0, 0.05, 0.004, 1e-6] if min(x.shape[0] for x in srcs) >= 2047 else [0.0, 0.0, 0.05, 1e-6])),
                              transformation_epsilon=float(rng.choice([0.0, 1e-10, 1e-6])),
                              relative_mse_epsilon=float(rng.choice([0.0, 0.005, 0.1])))
        res = engine.align(inits, p, ids, ids)
        for k, i in enumerate(ids):
            o = ora.icp(srcs[i], tgts[i], inits[k], ora.params_from(p), nthreads=4)
            tag = (f"case {case} job {k}: clouds {srcs[i].shape[0]} x {tgts[i].shape[0]}, kind {kind}, ({p.max_iterations}, "
                   f"{p.max_correspondence_distance}, {p.transformation_epsilon}, {p.relative_mse_epsilon})")
            assert res.status[k] == o["status"], f"{tag}: status {res.status[k]} vs oracle {o['status']}"
            assert res.iterations[k] == o["iterations"] and res.converged[k] == o["converged"], f"{tag}: {res.iterations[k]}/{res.converged[k]} vs {o['iterations']}/{o['converged']}"
            err = float(np.linalg.norm(res.poses[k] - o["pose"]))
            worst = max(worst, err)
            assert err <= 1e-6 * max(1.0, float(np.abs(o["pose"]).max())), f"{tag}: pose differs by {err}"
    print(f"{CASES} random point-to-point cases, worst pose difference to the oracle {worst:.2e}")

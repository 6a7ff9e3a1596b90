"""State fuzz: ONE engine, a seeded random sequence of uploads, scene preparations, correspondence queries and alignments in every
mode - without re-uploading between calls, which is what the other randomised tests do before each call.  What this looks for is
state that outlives its input: sorted copies, pyramid levels (the two forms of reference mode cache levels of different steps on the
same upload), cloud sums, pooled buffers handed from one cloud set to the next, a committed scene replacing uploaded targets and
the other way round.  Every result is compared with the oracle run on the clouds the test itself holds (bit for bit for indices and
distances, 1e-6 relative for poses, equal pass counts)."""
import os
import time

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st

pytestmark = pytest.mark.gpu
STEPS = int(os.environ.get("MI355_ICP_FUZZ_STEPS", "70"))


@pytest.mark.parametrize("slots", [1, 3], ids=["one_context", "handle_over_three_slots"])
def test_random_call_sequences_on_one_engine(slots):
    """slots = 3: the same sequence through mi355_icp_create_multi over three slots of device 0 - jobs spread over the slots when a
    call has enough of them, source slabs otherwise, the scene prepared on every slot."""
    import pose_estimation_amd as pa
    rng = np.random.default_rng(20261005 + slots)
    bases = [st.synth_cloud_np(1000 + b, 20000, normals=True) for b in range(3)]   # three surfaces everything is sampled from
    tgts, tgt_base = [], []          # what the engine's target set holds: N x 6 clouds and the surface each came from
    srcs, src_base, src_T = [], [], []
    counts = {}
    compared = [0, 0.0]   # jobs compared with the oracle, seconds the oracle took

    def new_targets(eng):
        nonlocal tgts, tgt_base
        tgts, tgt_base = [], []
        for _ in range(int(rng.integers(1, 4))):
            b = int(rng.integers(0, 3))
            n = int(rng.choice([40, 333, 1024, 1025, 3000, 6000]))
            tgts.append(np.ascontiguousarray(bases[b][rng.permutation(20000)[:n]]))
            tgt_base.append(b)
        eng.set_targets(tgts)

    def new_sources(eng):
        nonlocal srcs, src_base, src_T
        srcs, src_base, src_T = [], [], []
        for _ in range(int(rng.integers(1, 4))):
            # mostly a model of a surface the target set holds (an alignment needs one), sometimes of another
            b = int(rng.choice(tgt_base)) if tgt_base and rng.random() < 0.8 else int(rng.integers(0, 3))
            n = int(rng.choice([47, 500, 2048, 2049, 4000]))
            T = st.rigid_transform(float(rng.uniform(0.0, 3.0)), rng.standard_normal(3), rng.standard_normal(3) * 0.01)
            pick = rng.integers(0, 20000, n)
            p = bases[b][pick, :3].astype(np.float64) + rng.standard_normal((n, 3)) * 1e-3
            xyz = ((p - T[:3, 3]) @ T[:3, :3]).astype(np.float32)          # T maps the model onto the surface, up to the noise
            nrm = (bases[b][pick, 3:].astype(np.float64) @ T[:3, :3]).astype(np.float32)
            srcs.append(np.ascontiguousarray(np.hstack([xyz, nrm])))
            src_base.append(b)
            src_T.append(T)
        eng.set_sources(srcs)

    def new_scene(eng):
        """raw frame with NaN pixels -> finite filter -> k-NN normals -> commit: the target set becomes that one scene"""
        nonlocal tgts, tgt_base
        b = int(rng.integers(0, 3))
        n = int(rng.choice([1500, 4000]))
        raw = bases[b][rng.permutation(20000)[:n], :3].copy()
        raw[rng.random(n) < 0.05] = np.nan
        eng.scene_load(raw)
        kept = eng.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
        assert kept == int(np.isfinite(raw).all(axis=1).sum())
        normals = eng.scene_normals(int(rng.choice([8, 20])), True, (0.0, 0.0, 0.0))
        pts = eng.scene_download()
        assert np.array_equal(pts, raw[np.isfinite(raw).all(axis=1)])
        eng.scene_commit_target()
        tgts, tgt_base = [np.ascontiguousarray(np.hstack([pts, normals]))], [b]

    def jobs(matching):
        """(source ids, target ids): any pairing for a correspondence query, same-surface pairs for an alignment (or none)"""
        pairs = [(i, j) for i in range(len(srcs)) for j in range(len(tgts)) if not matching or src_base[i] == tgt_base[j]]
        if not pairs:
            return [], []
        chosen = [pairs[int(rng.integers(0, len(pairs)))] for _ in range(int(rng.integers(1, 4)))]
        return [p[0] for p in chosen], [p[1] for p in chosen]

    def start_poses(sids):
        return np.stack([st.rigid_transform(float(rng.uniform(0, 1.5)), rng.standard_normal(3), rng.standard_normal(3) * 0.004) @ src_T[i] for i in sids])

    with (pa.IcpEngine(0) if slots == 1 else pa.IcpEngine(device_ids=[0] * slots)) as eng:
        new_targets(eng)
        new_sources(eng)
        for step in range(STEPS):
            op = str(rng.choice(["targets", "sources", "scene", "correspond", "correspond", "align", "ref0", "ref4", "ref0", "ref4"]))
            tag = f"step {step} ({op}; {[c.shape[0] for c in srcs]} models, {[c.shape[0] for c in tgts]} targets)"
            if op == "targets":
                new_targets(eng)
            elif op == "sources":
                new_sources(eng)
            elif op == "scene":
                new_scene(eng)
            elif op == "correspond":
                sids, tids = jobs(matching=False)
                poses = np.stack([st.rigid_transform(float(rng.uniform(0, 20)), rng.standard_normal(3), rng.standard_normal(3) * 0.1) for _ in sids])
                gate = float(rng.choice([0.0, 0.0, 0.05]))
                idx, d2, sums = eng.correspond(poses, gate, source_ids=sids, target_ids=tids)
                off = 0
                for k, (i, j) in enumerate(zip(sids, tids)):
                    t_o = time.perf_counter()
                    oi, od, os_ = ora.correspond(poses[k], srcs[i][:, :3], tgts[j][:, :3], gate, use_kdtree=False, nthreads=4)
                    compared[0] += 1
                    compared[1] += time.perf_counter() - t_o
                    n = srcs[i].shape[0]
                    gi, gd = (idx[k], d2[k]) if isinstance(idx, (list, tuple)) else (idx[off:off + n], d2[off:off + n])
                    off += n
                    assert np.array_equal(np.asarray(gi), oi) and np.array_equal(np.asarray(gd).view(np.uint32), od.view(np.uint32)), tag
                    np.testing.assert_allclose(sums[k], os_, rtol=1e-10, atol=1e-10, err_msg=tag)
            elif op == "align":
                sids, tids = jobs(matching=True)
                if not sids:
                    continue
                p = pa.default_params(max_iterations=int(rng.choice([1, 4, 15])), relative_mse_epsilon=float(rng.choice([0.0, 0.01])))
                inits = start_poses(sids)
                res = eng.align(inits, p, sids, tids)
                for k, (i, j) in enumerate(zip(sids, tids)):
                    t_o = time.perf_counter()
                    o = ora.icp(srcs[i][:, :3], tgts[j][:, :3], inits[k], ora.params_from(p), nthreads=4)
                    compared[0] += 1
                    compared[1] += time.perf_counter() - t_o
                    assert res.status[k] == o["status"] and res.iterations[k] == o["iterations"], tag
                    assert np.linalg.norm(res.poses[k] - o["pose"]) <= 1e-6 * max(1.0, float(np.abs(o["pose"]).max())), tag
            else:
                upstream = op == "ref4"
                sids, tids = jobs(matching=True)
                sids_tids = [(i, j) for i, j in zip(sids, tids) if tgts[j].shape[0] >= 40]
                if not sids_tids:
                    continue
                sids, tids = [p[0] for p in sids_tids], [p[1] for p in sids_tids]
                iters, tol = int(rng.choice([3, 30, 250])), float(rng.choice([0.005, 0.05]))
                scale, levels = float(rng.choice([2.5, 0.0])), int(rng.integers(1, 9))
                inits = start_poses(sids)
                res = eng.align_reference(inits, iters, tol, scale, levels, source_ids=sids, target_ids=tids, flags=4 if upstream else 0)
                for k, (i, j) in enumerate(zip(sids, tids)):
                    t_o = time.perf_counter()
                    o = (ora.icp_reference_upstream if upstream else ora.icp_reference)(srcs[i], tgts[j], inits[k], iters, tol, scale, levels, nthreads=4)
                    compared[0] += 1
                    compared[1] += time.perf_counter() - t_o
                    assert res.status[k] == 0, tag
                    assert res.passes[k] == o["passes"], f"{tag}: passes {res.passes[k]} vs oracle {o['passes']}"
                    assert np.linalg.norm(res.poses[k] - o["pose"]) <= 1e-6 * max(1.0, float(np.abs(o["pose"]).max())), tag
            counts[op] = counts.get(op, 0) + 1
    assert compared[0] >= STEPS // 2
    print(f"{STEPS} steps on one engine ({slots} slot(s)): {counts}; {compared[0]} jobs compared with the oracle ({compared[1]:.1f} s of oracle time)")

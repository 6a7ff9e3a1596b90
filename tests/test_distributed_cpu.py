"""N > 1 host logic on CPU: world_size-2 gloo.  Each rank owns a contiguous source slab
(mi355_icp_slab_range), the per-iteration sums are all-reduced, and every rank runs the product's
host controller (mi355_icp_host_update) on the reduced sums.  The per-slab correspondence pass is
supplied by the oracle here (there is no GPU in this container); on the GPU box the same driver
runs with pose_estimation_amd.distributed.engine_pass (HIP) — see tests/test_gpu_distributed.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import pose_estimation_amd as pa
    from pose_estimation_amd import distributed as pd
    import oracle_lib as ora
    import synth_twin as st

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # two jobs: different clouds, one of them with rejection-free early convergence
        jobs = []
        for k, (ns, nt) in enumerate(((3001, 2500), (1777, 2100))):
            tgt = pa.synth_cloud(st.TARGET_SEED + 16 * k, nt)
            b, c = pa.slab_range(ns, rank, world)
            slab = pa.synth_cloud(st.SOURCE_SEED + 16 * k, c, first_index=b, noise_seed=st.NOISE_SEED + 16 * k,
                                  sigma=st.SIGMA, transform=st.perturbation(k))
            jobs.append((slab, tgt, ora.KdTree(tgt)))

        def local_pass(poses, job_ids):
            out = np.empty((len(job_ids), 17))
            for a, j in enumerate(job_ids):
                slab, tgt, tree = jobs[j]
                idx, d2 = tree.query(ora.transform(poses[a], slab))
                out[a] = ora.accumulate(slab, tgt, idx, d2)
            return out

        params = pa.default_params(max_iterations=12, relative_mse_epsilon=2e-3)
        res = pd.align_external_collective(local_pass, pd.torch_allreduce(), np.stack([np.eye(4), np.eye(4)]), params)
        q.put((rank, res.poses, res.residual, res.iterations, res.status, res.converged))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_slab_sharded_alignment_matches_single_process_oracle():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ora
    import pose_estimation_amd as pa
    import synth_twin as st

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, poses0, res0, it0, st0, cv0), (r1, poses1, res1, it1, st1, cv1) = got
    # both ranks saw the same reduced sums, so they hold bit-identical results
    assert np.array_equal(poses0, poses1) and np.array_equal(it0, it1) and np.array_equal(cv0, cv1)
    params = pa.default_params(max_iterations=12, relative_mse_epsilon=2e-3)
    for k, (ns, nt) in enumerate(((3001, 2500), (1777, 2100))):
        tgt = pa.synth_cloud(st.TARGET_SEED + 16 * k, nt)
        src = pa.synth_cloud(st.SOURCE_SEED + 16 * k, ns, noise_seed=st.NOISE_SEED + 16 * k, sigma=st.SIGMA,
                             transform=st.perturbation(k))
        o = ora.icp(src, tgt, np.eye(4), ora.params_from(params))
        assert st0[k] == 0 and it0[k] == o["iterations"] and cv0[k] == o["converged"]
        assert np.linalg.norm(poses0[k] - o["pose"]) <= 1e-9     # only the summation order differs
        assert abs(res0[k] - o["residual"]) <= 1e-12
    assert it0.min() < 12 or cv0.max() == 0  # bookkeeping sanity: iterations are per job

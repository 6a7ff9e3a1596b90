#!/usr/bin/env python3
"""One rank of a NATIVE-RCCL slab-sharded run (started by tests/test_gpu_rccl_ranks.py with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set): torch.distributed (backend nccl = RCCL) only carries the 128-byte unique id, the engine joins an RCCL
communicator of its own (mi355_icp_comm_attach) and runs BOTH of its loops on a slab of every source cloud - the
point-to-point loop (ncclAllGather of the 17 sums + rank-ordered sum per pass) and reference mode (summed rank-select
histograms, min-reduced owner table, rank-ordered 29 sums).  Rank 0 also holds the whole clouds on a second engine without a
communicator and compares: same iteration / pass counts, poses equal to 1e-12 (sums differ by summation order only).
Prints one line "RCCL_RANKS_OK <world>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import pose_estimation_amd as pa
    from pose_estimation_amd import distributed as pd
    import synth_twin as st

    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    try:
        sizes = ((40001, 30000), (9777, 21000))  # do not divide by the rank count or the pyramid strides
        targets, slabs, wholes = [], [], []
        for k, (ns, nt) in enumerate(sizes):
            targets.append(pa.synth_cloud(st.TARGET_SEED + 16 * k, nt, normals=True))
            b, c = pa.slab_range(ns, rank, world)
            kw = dict(noise_seed=st.NOISE_SEED + 16 * k, sigma=st.SIGMA, transform=st.perturbation(k))
            slabs.append(pa.synth_cloud(st.SOURCE_SEED + 16 * k, c, first_index=b, **kw))
            if rank == 0:
                wholes.append(pa.synth_cloud(st.SOURCE_SEED + 16 * k, ns, **kw))
        ids = [0, 1, 0]
        inits = np.stack([np.eye(4), np.eye(4), st.rigid_transform(1.5, (0.2, 1, -0.4), (0.01, -0.02, 0.015))])
        params = pa.default_params(max_iterations=7)
        with pa.IcpEngine(local) as eng:
            eng.set_targets(targets)
            eng.set_sources(slabs)
            pd.attach_native_comm(eng)
            info = eng.comm_info()
            assert info[0] == rank and info[1] == world, info
            p2p = eng.align(inits, params, source_ids=ids, target_ids=ids)
            ref = eng.align_reference(inits, 60, 0.005, 2.5, 5, source_ids=ids, target_ids=ids)
            eng.comm_detach()
        # every rank must hold the same answer, bit for bit (rank-ordered sums, identical host arithmetic)
        for arr in (p2p.poses, ref.poses):
            t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
            lo, hi = t.clone(), t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), "ranks disagree"
        if rank == 0:
            with pa.IcpEngine(local) as one:
                one.set_targets(targets)
                one.set_sources(wholes)
                w_p2p = one.align(inits, params, source_ids=ids, target_ids=ids)
                w_ref = one.align_reference(inits, 60, 0.005, 2.5, 5, source_ids=ids, target_ids=ids)
            assert np.array_equal(p2p.iterations, w_p2p.iterations) and np.array_equal(ref.passes, w_ref.passes)
            assert np.abs(p2p.poses - w_p2p.poses).max() <= 1e-12, np.abs(p2p.poses - w_p2p.poses).max()
            assert np.abs(ref.poses - w_ref.poses).max() <= 1e-9, np.abs(ref.poses - w_ref.poses).max()
            print(f"RCCL_RANKS_OK {world}", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

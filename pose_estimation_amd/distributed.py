"""Slab-sharded alignment with the collective run by the caller.

Three ways to run the per-iteration exchange of the MI355_ICP_NSUMS x n_jobs doubles:

* native (preferred, used by bench.py): ``IcpEngine.comm_attach`` joins the engine to an RCCL
  communicator and ``IcpEngine.align`` exchanges the sums once per iteration on its own stream
  (``ncclAllGather`` + a rank-ordered sum on every rank) — no Python in the loop.
  ``attach_native_comm`` below does the rendezvous through an existing torch.distributed process
  group (it only ships the 128-byte RCCL unique id).
* native loop, caller's transport: ``attach_external_comm`` hands the engine an all-gather that runs
  over any torch.distributed group (gloo in the tests, where two ranks share one GPU and RCCL cannot
  run); the engine's own loop and its rank-ordered reductions are what executes.
* external (this module's ``align_external_collective``): the caller supplies the reduction
  (e.g. ``torch.distributed.all_reduce`` on RCCL or, in the CPU tests, gloo) and the loop runs here,
  calling the SAME host controller (``mi355_icp_host_update``) the native loop uses, so both make
  identical decisions.

Every rank holds a contiguous slab of each source cloud (``slab_range``) and the whole target.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib
from .engine import AlignResult, IcpEngine, host_update, slab_range  # noqa: F401

PassFn = Callable[[np.ndarray, Sequence[int]], np.ndarray]
ReduceFn = Callable[[np.ndarray], np.ndarray]


def align_external_collective(local_pass: PassFn, allreduce_sum: ReduceFn, init_poses: np.ndarray, params) -> AlignResult:
    """ICP loop with a caller-supplied collective.

    local_pass(poses [n_active, 4, 4], job_ids) -> this rank's sums [n_active, NSUMS] (no reduction);
    allreduce_sum(array) -> the element-wise sum over all ranks (every rank must get identical bits).
    Mirrors mi355_icp_align: jobs leave the active set when they converge, hit max_iterations, or
    fail (status kept, pose left at its last valid value).
    """
    poses = np.array(init_poses, dtype=np.float64).reshape(-1, 4, 4).copy()
    n = poses.shape[0]
    prev_mse = np.zeros(n)
    residual = np.zeros(n)
    iters = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    conv = np.zeros(n, dtype=np.int32)
    active = list(range(n))
    for it in range(params.max_iterations):
        if not active:
            break
        local = np.ascontiguousarray(local_pass(poses[active], active), dtype=np.float64)
        assert local.shape == (len(active), _lib.NSUMS)
        total = allreduce_sum(local)
        nxt = []
        for a, k in enumerate(active):
            st, new_pose, mse, done, c = host_update(params, total[a], it, poses[k], prev_mse[k])
            if st != _lib.OK:
                status[k] = st
                continue
            poses[k] = new_pose
            prev_mse[k] = mse
            residual[k] = np.sqrt(mse)
            iters[k] = it + 1
            conv[k] = int(c)
            if not done:
                nxt.append(k)
        active = nxt
    return AlignResult(poses, residual, iters, status, conv)


def engine_pass(engine: IcpEngine, source_ids: Sequence[int], target_ids: Sequence[int],
                max_correspondence_distance: float = 0.0) -> PassFn:
    """local_pass backed by the HIP engine (one mi355_icp_correspond per iteration)."""
    def run(poses: np.ndarray, job_ids: Sequence[int]) -> np.ndarray:
        sid = [source_ids[j] for j in job_ids]
        tid = [target_ids[j] for j in job_ids]
        _, _, sums = engine.correspond(poses, max_correspondence_distance, sid, tid, want_pairs=False)
        return sums
    return run


def torch_allreduce(group=None, device: Optional[str] = None) -> ReduceFn:
    """allreduce_sum over a torch.distributed process group (RCCL for backend 'nccl', gloo on CPU)."""
    import torch
    import torch.distributed as dist

    def run(local: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    return run


def attach_native_comm(engine: IcpEngine, group=None) -> None:
    """Joins `engine` to an RCCL communicator spanning the torch.distributed group: rank 0 draws the
    unique id, torch.distributed broadcasts its 128 bytes, every rank calls ncclCommInitRank inside
    the engine."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [IcpEngine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_attach(box[0], rank, world)


def attach_external_comm(engine: IcpEngine, group=None) -> None:
    """Joins `engine` to the ranks of a torch.distributed group through a HOST all-gather (any backend; gloo in the
    tests): the engine's own slab-sharded loops run, staging each exchange through host memory."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def allgather(send) -> bytes:
        mine = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        out = torch.empty(world * mine.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, mine, group=group)
        return out.numpy().tobytes()

    engine.comm_attach_external(rank, world, allgather)

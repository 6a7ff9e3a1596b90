"""Native RCCL across ranks, one GPU per rank ("a 2-rank native-RCCL parity check that runs whenever at least 2
GPUs are visible").  The worker (tests/rccl_ranks_worker.py) runs both engine loops slab-sharded over an RCCL communicator
and compares with the unsharded engine.  World size 1 always runs on the GPU box - it validates the worker itself and the
one-rank communicator; world size 2 runs when two devices are visible (never on the 1-GPU pool: then it is skipped, and
the 2-/3-rank arithmetic is covered over gloo by test_gpu_distributed.py)."""
import os
import socket
import subprocess
import sys
import tempfile
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def visible_gpus():
    import torch
    return torch.cuda.device_count()  # counts devices without initialising the GPU in this process


def run_world(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # every rank writes to files of its own: a rank that fills a 64 KiB pipe (an RCCL banner, a traceback) while the parent
    # is waiting on another rank would block until the timeout
    with tempfile.TemporaryDirectory() as tmp:
        procs, logs = [], []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            out, err = open(os.path.join(tmp, f"rank{r}.out"), "w+"), open(os.path.join(tmp, f"rank{r}.err"), "w+")
            logs.append((out, err))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_ranks_worker.py")], env=env, stdout=out, stderr=err))
        deadline = time.monotonic() + 300
        try:
            # a dead rank leaves the others in the rendezvous: end the run as soon as one fails
            while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
                if any(p.poll() not in (None, 0) for p in procs):
                    break
                time.sleep(0.2)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                p.wait()
        texts = []
        for out, err in logs:
            out.seek(0), err.seek(0)
            texts.append((out.read(), err.read()))
            out.close(), err.close()
    for r, (p, (out, err)) in enumerate(zip(procs, texts)):
        assert p.returncode == 0, f"rank {r} failed (rc {p.returncode}):\n{err[-3000:]}"
    assert f"RCCL_RANKS_OK {world}" in texts[0][0], texts[0]


def test_native_rccl_one_rank():
    run_world(1)


def test_native_rccl_two_ranks():
    if visible_gpus() < 2:
        pytest.skip("needs two visible GPUs (one per rank)")
    run_world(2)

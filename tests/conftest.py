import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Tests of the multi-process paths import torch (for torch.distributed / torch.multiprocessing).  On a freshly leased box the
# FIRST import of torch pages the whole wheel in and has been seen to take more than five minutes (normally 1.5 s) - longer
# than those tests' own time limits, which then fail for a reason that has nothing to do with them.  Pay for that import here,
# once, right after collection and outside any per-test limit; the children the tests start find a warm page cache.
_TORCH_USERS = {"test_gpu_distributed", "test_distributed_cpu", "test_bench_launcher_cpu", "test_gpu_rccl_ranks"}


def pytest_collection_finish(session):
    if any(item.module.__name__.rsplit(".", 1)[-1] in _TORCH_USERS for item in session.items):
        import torch  # noqa: F401


@pytest.fixture(scope="session")
def engine():
    """One IcpEngine on device 0 for the whole GPU session; fails loudly without a GPU."""
    import pose_estimation_amd as pa
    eng = pa.IcpEngine(0)
    yield eng
    eng.close()

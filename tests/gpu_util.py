"""Helpers shared by the -m gpu tests (test infrastructure; nothing here computes)."""
import os


def engine_with_mode(mode, device=0):
    """A fresh IcpEngine whose scan kernel is forced with MI355_ICP_SCAN_MODE=mode (read once, at create)."""
    import pose_estimation_amd as pa
    old = os.environ.get("MI355_ICP_SCAN_MODE")
    os.environ["MI355_ICP_SCAN_MODE"] = mode
    try:
        return pa.IcpEngine(device)
    finally:
        if old is None:
            os.environ.pop("MI355_ICP_SCAN_MODE")
        else:
            os.environ["MI355_ICP_SCAN_MODE"] = old


def overflow_points(eng) -> int:
    """Source points the filter scans handed to the all-exact scan so far (mi355_icp_profile.overflow_points)."""
    return int(eng.profile().overflow_points)


def overflow_host_passes(eng) -> int:
    """Passes whose overflow points took the host path (mi355_icp_profile.overflow_host_passes)."""
    return int(eng.profile().overflow_host_passes)

"""pose_estimation_amd — MI355X-native ICP refinement for the yumi-crew/pose_estimation hot path.

Scope: the call cv::ppf_match_3d::ICP::registerModelToScene made by
OpenCVSurfaceMatch::find_object_in_scene (reference pose_estimation/src/opencv_surface_match.cpp:85,94),
rebuilt as hand-written gfx950 HIP kernels behind the C-ABI of include/mi355_icp.h.
Importing this package loads the in-tree libmi355_icp.so and fails loudly if it is not built.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from .engine import (AlignResult, IcpEngine, IcpError, ReferenceResult, default_params, host_update, level_slab, level_slab_step, upstream_step,
                     params_from_reference, pose_to_pose7, slab_range, solve_point_to_plane, solve_point_to_plane_batch, solve_rigid, synth_cloud)
from .icp import ICP, Pose3D  # noqa: F401

__all__ = ["AlignResult", "IcpEngine", "IcpError", "ICP", "Pose3D", "ReferenceResult", "default_params", "host_update",
           "level_slab", "level_slab_step", "upstream_step", "params_from_reference", "pose_to_pose7", "slab_range", "solve_point_to_plane", "solve_point_to_plane_batch", "solve_rigid", "synth_cloud"]

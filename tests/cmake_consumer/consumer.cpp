// Links against libmi355_icp.so through the CMake package and calls host-only entry points (no GPU needed).
#include <cstdio>

#include <mi355_icp.h>
#include <pose_estimation_amd/host/mi355_icp.hpp>

int main() {
  mi355_icp_params p;
  if (mi355_icp_params_from_reference(250, 0.005f, 2.5f, 8, &p) != MI355_ICP_OK) return 1;
  const double pose[16] = {1, 0, 0, 0.5, 0, 1, 0, -0.25, 0, 0, 1, 2, 0, 0, 0, 1};
  float pose7[7];
  if (mi355_icp_pose_to_pose7(pose, pose7, 0) != MI355_ICP_OK) return 2;
  mi355::ppf_match_3d::Pose3D candidate;  // the C++ mirror's types are visible too
  candidate.numVotes = 1;
  std::printf("abi %u max_iterations %d pose7 %g %g %g %g\n", mi355_icp_abi_version(), p.max_iterations, pose7[0], pose7[1], pose7[2], pose7[6]);
  return pose7[6] == 1.0f && p.max_iterations == 250 ? 0 : 3;
}

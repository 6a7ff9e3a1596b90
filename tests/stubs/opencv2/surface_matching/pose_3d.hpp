// TEST DOUBLE - NOT opencv_contrib.  cv::ppf_match_3d::Pose3D as far as the reference's call site and
// pose_estimation_amd/host/opencv_adapter.hpp use it: pose (cv::Matx44d), residual, numVotes, updatePose(Matx44d).
// See tests/stubs/opencv2/core.hpp for why this exists.
#pragma once
#include <memory>
#include "../core.hpp"

namespace cv {
namespace ppf_match_3d {

class Pose3D {
 public:
  Matx44d pose;
  double residual = 0.0;
  std::size_t numVotes = 0;
  int updates = 0;  // test instrumentation: how often updatePose ran
  Pose3D() {
    for (int k = 0; k < 4; ++k) pose(k, k) = 1.0;
  }
  void updatePose(const Matx44d& m) {
    pose = m;
    ++updates;
  }
};
using Pose3DPtr = std::shared_ptr<Pose3D>;  // upstream: cv::Ptr<Pose3D>, which is a std::shared_ptr

}  // namespace ppf_match_3d
}  // namespace cv

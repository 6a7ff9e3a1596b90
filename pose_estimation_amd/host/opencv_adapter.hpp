// opencv_adapter.hpp — drop-in for the two ICP lines of the reference's
// OpenCVSurfaceMatch::find_object_in_scene (pose_estimation/src/opencv_surface_match.cpp:85,94):
//
//   -  cv::ppf_match_3d::ICP icp(250, 0.005f, 2.5f, 8);
//   +  mi355::cv_adapter::ICP icp(250, 0.005f, 2.5f, 8);
//      ...
//      icp.registerModelToScene(models_[object], pc_scene_normals, results_subset);   // unchanged
//
// Needs OpenCV + opencv_contrib headers, which exist neither in the build container nor on the GPU
// box, so this file is compile-guarded and shipped as source (SURVEY.md 8b "Build constraint").
// Everything numeric is in libmi355_icp.so; this only converts cv::Mat / Pose3DPtr <-> raw pointers.
#pragma once
#ifdef MI355_ICP_WITH_OPENCV

#include <opencv2/core.hpp>
#include <opencv2/surface_matching/pose_3d.hpp>

#include "mi355_icp.hpp"

namespace mi355 {
namespace cv_adapter {

inline CloudView view_of(const cv::Mat& m) {
  CV_Assert(m.type() == CV_32FC1 && m.cols >= 3);  // N x 3 (pose_estimation.cpp:263) or N x 6 (opencv_surface_match.cpp:22,59)
  CloudView v;
  v.data = m.ptr<float>(0);
  v.n = static_cast<std::size_t>(m.rows);
  v.stride_bytes = m.step[0];
  return v;
}

class ICP {
 public:
  // Mode::Reference (default here) keeps all four arguments live like upstream; the scene must then be N x 6
  // (pc_scene_normals at the call site is: opencv_surface_match.cpp:57-59).
  ICP(int iterations, float tolerence = 0.05f, float rejectionScale = 2.5f, int numLevels = 6,
      ppf_match_3d::ICP::Mode mode = ppf_match_3d::ICP::Mode::Reference)
      : impl_(iterations, tolerence, rejectionScale, numLevels, 0, mode) {}

  int registerModelToScene(const cv::Mat& srcPC, const cv::Mat& dstPC, std::vector<cv::ppf_match_3d::Pose3DPtr>& poses) {
    std::vector<ppf_match_3d::Pose3DPtr> mine;
    for (const auto& p : poses) {
      auto q = std::make_shared<ppf_match_3d::Pose3D>();
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) q->pose[4 * r + c] = p->pose(r, c);
      mine.push_back(q);
    }
    const int rc = impl_.registerModelToScene(view_of(srcPC), view_of(dstPC), mine);
    for (std::size_t k = 0; k < poses.size(); ++k) {
      cv::Matx44d m;
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = mine[k]->pose[4 * r + c];
      poses[k]->updatePose(m);  // keeps t, q, angle consistent, as upstream's appendPose does
      poses[k]->residual = mine[k]->residual;
    }
    return rc;
  }

 private:
  ppf_match_3d::ICP impl_;
};

}  // namespace cv_adapter
}  // namespace mi355

#endif  // MI355_ICP_WITH_OPENCV

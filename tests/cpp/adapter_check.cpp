// Drives pose_estimation_amd/host/opencv_adapter.hpp exactly as the patched reference would
// (opencv_surface_match.cpp:85-94 with the one-line change of INTEGRATION.md): an N x 6 model, an M x 6 scene with
// normals, <= 6 Pose3DPtr refined in place.  Compiled against the test doubles of tests/stubs/ (OpenCV does not exist
// here); prints the refined poses and residuals so tests/test_gpu_opencv_adapter.py can compare them with the engine's.
#define MI355_ICP_WITH_OPENCV
#include "../../pose_estimation_amd/host/opencv_adapter.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
  const int n_model = argc > 1 ? std::atoi(argv[1]) : 3000, n_scene = argc > 2 ? std::atoi(argv[2]) : 5000;
  cv::Mat model(n_model, 6, CV_32FC1), scene(n_scene, 6, CV_32FC1);
  const double T[12] = {0.9961946980917455, -0.0716791773074155, 0.0491085115000797, 0.02,
                        0.0738010430565223, 0.9963425807891292, -0.0428275146684137, -0.01,
                        -0.0458590115510097, 0.0462887393339173, 0.9978748139193905, 0.03};  // ~5 degrees about (1,2,3)
  mi355_icp_synth_cloud6(0x5EED0001ull, 0, 0.0, nullptr, 0, static_cast<size_t>(n_scene), scene.ptr<float>(0));
  mi355_icp_synth_cloud6(0x5EED0002ull, 0x5EED0003ull, 1e-3, T, 0, static_cast<size_t>(n_model), model.ptr<float>(0));
  try {
    mi355::cv_adapter::ICP icp(250, 0.005f, 2.5f, 8);  // the reference's arguments; Mode::Reference by default
    std::vector<cv::ppf_match_3d::Pose3DPtr> results_subset;
    for (int k = 0; k < 3; ++k) {
      auto p = std::make_shared<cv::ppf_match_3d::Pose3D>();
      p->pose(0, 3) = 0.002 * k;  // three candidates: identity and two small translations
      p->numVotes = 100 + k;
      results_subset.push_back(p);
    }
    const int rc = icp.registerModelToScene(model, scene, results_subset);
    std::printf("rc %d\n", rc);
    for (std::size_t k = 0; k < results_subset.size(); ++k) {
      std::printf("pose %zu", k);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) std::printf(" %.17g", results_subset[k]->pose(r, c));
      std::printf("\nresidual %zu %.17g updates %d votes %zu\n", k, results_subset[k]->residual, results_subset[k]->updates, results_subset[k]->numVotes);
    }
  } catch (const mi355::Error& e) {
    std::fprintf(stderr, "mi355::Error %d: %s\n", e.status, e.what());
    return 1;
  }
  return 0;
}

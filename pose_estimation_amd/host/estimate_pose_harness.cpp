// estimate_pose_harness.cpp — ROS-free stand-in for the EstimatePose request path
// (pose_estimation_interface/srv/EstimatePose.srv:1-7 -> PoseEstimation::estimate_pose_service_handler,
// pose_estimation.cpp:65-95 -> OpenCVSurfaceMatch::find_object_in_scene, opencv_surface_match.cpp:53-148),
// driving the refine half through the C++ mirror.  PPF matching is not part of this path, so candidate
// poses are synthesised around the known alignment the way PPF votes would deliver them.
//
//   estimate_pose_harness <n_model> <n_scene> [n_candidates] [reference|pipeline] [pruned] [legacy]
//     prints the 7-float pose and per-candidate results; "reference" selects Mode::Reference (scene with normals);
//     "pipeline" starts from a raw camera-like frame (NaN dropouts + a table plane) and runs the whole
//     create_surface_match_pc -> computeNormalsPC3d -> ICP chain on the device: NaN removal, plane band, k = 20
//     normals, commit, reference-mode refine (pose_estimation.cpp:211-279, opencv_surface_match.cpp:57-94); the request is
//     repeated three times (MI355_HARNESS_PASSES=n: n times) with one "pipeline pass" line and one "detail pass" line
//     (per-call stamps of the scene preparation) each - the second request must already be as fast as the third
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mi355_icp.hpp"

namespace {

// twin of EstimatePose.srv (request fields :1-5, response :7)
struct EstimatePoseRequest {
  std::string object;
  long long num_planes = 0;
  std::string filter_out;
  float filter_radius = 0.f;
  bool store_filter_pose = false;
};
struct EstimatePoseResponse {
  bool success = false;
};

mi355::Pose rotation_about(double ax, double ay, double az, double deg, double tx, double ty, double tz) {
  const double n = std::sqrt(ax * ax + ay * ay + az * az), x = ax / n, y = ay / n, z = az / n;
  const double th = deg * 3.14159265358979323846 / 180.0, c = std::cos(th), s = std::sin(th), C = 1 - c;
  return mi355::Pose{c + x * x * C,     x * y * C - z * s, x * z * C + y * s, tx,
                     y * x * C + z * s, c + y * y * C,     y * z * C - x * s, ty,
                     z * x * C - y * s, z * y * C + x * s, c + z * z * C,     tz,
                     0, 0, 0, 1};
}

}  // namespace

int main(int argc, char** argv) {
  const size_t n_model = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000;
  const size_t n_scene = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1000;
  const int n_cand = argc > 3 ? std::atoi(argv[3]) : 7;
  const bool pipeline = argc > 4 && std::string(argv[4]) == "pipeline";
  const bool reference_mode = pipeline || (argc > 4 && std::string(argv[4]) == "reference");
  const bool pruned = argc > 5 && std::string(argv[5]) == "pruned";  // MI355_ICP_FLAG_PRUNED_SCAN for the ICP
  const bool legacy = (argc > 5 && std::string(argv[5]) == "legacy") || (argc > 6 && std::string(argv[6]) == "legacy");  // the round 1-3 form of reference mode
  EstimatePoseRequest req;
  req.object = "bumpy_sphere";
  EstimatePoseResponse resp;
  try {
    // scene = target sample; model = independent noisy sample moved by a known 5 degree perturbation
    const mi355::Pose T = rotation_about(1, 2, 3, 5.0, 0.02, -0.01, 0.03);
    // reference mode: both clouds as N x 6 rows (xyz + normal), the layout of the reference's models
    // (opencv_surface_match.cpp:22) and of the scene it hands to the ICP (:57-59)
    const size_t scene_cols = reference_mode ? 6 : 3, model_cols = scene_cols;
    std::vector<float> scene(scene_cols * n_scene), model(model_cols * n_model);
    if (reference_mode) mi355_icp_synth_cloud6(0x5EED0001ull, 0, 0.0, nullptr, 0, n_scene, scene.data());
    else mi355_icp_synth_cloud(0x5EED0001ull, 0, 0.0, nullptr, 0, n_scene, scene.data());
    if (reference_mode) mi355_icp_synth_cloud6(0x5EED0002ull, 0x5EED0003ull, 1e-3, T.data(), 0, n_model, model.data());
    else mi355_icp_synth_cloud(0x5EED0002ull, 0x5EED0003ull, 1e-3, T.data(), 0, n_model, model.data());

    using mi355::ppf_match_3d::ICP;
    ICP icp(250, 0.005f, 2.5f, 8, 0, reference_mode ? ICP::Mode::Reference : ICP::Mode::PointToPoint);  // the reference's arguments
    icp.setPrunedScan(pruned);
    icp.setUpstreamFaithful(!legacy);
    std::vector<mi355::ppf_match_3d::Pose3DPtr> results;
    const size_t votes[] = {300, 900, 500, 450, 100, 420, 50, 10};
    for (int k = 0; k < n_cand; ++k) {
      auto p = std::make_shared<mi355::ppf_match_3d::Pose3D>();
      p->numVotes = votes[k % 8];
      p->updatePose(rotation_about(0.2, 1, -0.4, 1.0 + k, 0.01, -0.02, 0.015));
      results.push_back(p);
    }
    size_t best = 0;
    std::vector<float> pose;
    if (pipeline) {
      // raw frame = the scene sample + a table plane z = -1.3 under it + 5 % NaN pixels, interleaved
      std::vector<float> raw;
      raw.reserve(3 * (n_scene + n_scene / 2));
      for (size_t i = 0; i < n_scene; ++i) {
        const bool drop = (i % 20) == 7;
        for (int c2 = 0; c2 < 3; ++c2) raw.push_back(drop ? std::nanf("") : scene[6 * i + c2]);
        if (i % 2 == 0) {
          raw.push_back(-1.5f + 3.0f * static_cast<float>(i % 1000) / 1000.0f);
          raw.push_back(-1.5f + 3.0f * static_cast<float>((i / 7) % 1000) / 1000.0f);
          raw.push_back(-1.3f);
        }
      }
      mi355::Engine& eng = icp.engine();
      // the object model goes up once, like the reference's (load_models at start-up keeps it in the detector:
      // pose_estimation.cpp, opencv_surface_match.cpp:load_model); a request brings a new scene only
      eng.set_sources({mi355::CloudView{model.data(), n_model, 4 * model_cols}});
      // the request is run three times: the last pass shows the steady state of a service (buffers sized, code loaded)
      const char* reps_env = std::getenv("MI355_HARNESS_PASSES");  // diagnostic: more repetitions of the request
      const int reps = reps_env && std::atoi(reps_env) > 0 ? std::atoi(reps_env) : 3;
      for (int rep = 0; rep < reps; ++rep) {
      for (int k = 0; k < n_cand; ++k) results[k]->updatePose(rotation_about(0.2, 1, -0.4, 1.0 + k, 0.01, -0.02, 0.015));
      const auto t0 = std::chrono::steady_clock::now();
      eng.scene_load(mi355::CloudView{raw.data(), raw.size() / 3, 12});
      const auto ta = std::chrono::steady_clock::now();
      // create_surface_match_pc's chain (pose_estimation.cpp:248-258): NaN removal, then remove_planes(num_planes = 1):
      // RANSAC plane fit (:285-297) on the cleaned cloud, then the distance band around the FITTED plane (:310-333)
      mi355_scene_filter fin{};
      fin.kind = MI355_SCENE_FILTER_FINITE;
      eng.scene_filter({fin});
      const auto tb = std::chrono::steady_clock::now();
      std::size_t plane_inliers = 0;
      const std::array<float, 4> plane = eng.scene_fit_plane(0.0001f, 100, &plane_inliers);
      const auto tc = std::chrono::steady_clock::now();
      mi355_scene_filter band{};
      band.kind = MI355_SCENE_FILTER_PLANE_BAND_REMOVE;  // band 0.005 (pose_estimation.cpp:320)
      band.p[0] = plane[0]; band.p[1] = plane[1]; band.p[2] = plane[2]; band.p[3] = plane[3]; band.p[4] = 0.005f;
      const size_t kept = eng.scene_filter({band});
      if (rep == 0)
        std::printf("fitted plane %.7g %.7g %.7g %.7g with %zu inliers (the table was put at z = -1.3)\n", plane[0], plane[1], plane[2], plane[3], plane_inliers);
      const auto t1 = std::chrono::steady_clock::now();
      eng.scene_normals(20, true, {0.f, 0.f, 0.f});  // computeNormalsPC3d(scene, out, 20, true, viewpoint 0)
      const auto t2 = std::chrono::steady_clock::now();
      eng.scene_commit_target();
      const auto t3 = std::chrono::steady_clock::now();
      std::vector<mi355::ppf_match_3d::Pose3DPtr> subset(results.begin(), results.begin() + static_cast<std::ptrdiff_t>(results.size() > 6 ? 6 : results.size()));
      icp.registerUploaded(subset, 0, 0);
      const auto t4 = std::chrono::steady_clock::now();
      best = mi355::ppf_match_3d::select_best_pose(subset, results.size());
      const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      std::printf("pipeline pass %d: %zu raw points -> %zu kept | upload+filters+plane fit %.3f ms, normals %.3f ms, commit %.3f ms, "
                  "ICP of %zu candidates %.3f ms, total %.3f ms\n",
                  rep, raw.size() / 3, kept, ms(t0, t1), ms(t1, t2), ms(t2, t3), subset.size(), ms(t3, t4), ms(t0, t4));
      std::printf("  detail pass %d: load %.3f ms, NaN removal %.3f ms, plane fit %.3f ms, band removal %.3f ms\n", rep, ms(t0, ta), ms(ta, tb),
                  ms(tb, tc), ms(tc, t1));
      }
      pose.assign(7, 0.f);
      mi355_icp_pose_to_pose7(results[best]->pose.data(), pose.data(), 1);
    } else {
      pose = mi355::ppf_match_3d::refine_candidates(icp, mi355::CloudView{model.data(), n_model, 4 * model_cols},
                                                    mi355::CloudView{scene.data(), n_scene, 4 * scene_cols}, results, true, &best);
    }
    for (size_t k = 0; k < results.size(); ++k)
      std::printf("candidate %zu votes %zu residual %.17g iterations %d\n", k, results[k]->numVotes, results[k]->residual,
                  k < icp.last_result().iterations.size() ? icp.last_result().iterations[k] : 0);
    std::printf("best %zu\n", best);
    std::printf("pose7 %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
    for (size_t k = 0; k < results.size(); ++k) {
      std::printf("pose %zu", k);
      for (double v : results[k]->pose) std::printf(" %.17g", v);
      std::printf("\n");
    }
    resp.success = true;  // pose_estimation.cpp:178 reports success unconditionally on this path
  } catch (const mi355::Error& e) {
    std::fprintf(stderr, "%s (object %s)\n", e.what(), req.object.c_str());
    resp.success = false;
  }
  return resp.success ? 0 : 1;
}

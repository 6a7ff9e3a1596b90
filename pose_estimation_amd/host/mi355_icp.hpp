// mi355_icp.hpp — C++ host side above the C-ABI (include/mi355_icp.h), written in the reference's own
// language and mirroring the operator interface it uses on this path:
//
//     cv::ppf_match_3d::ICP icp(250, 0.005f, 2.5f, 8);                  // opencv_surface_match.cpp:85
//     icp.registerModelToScene(models_[object], scene, results_subset);  // opencv_surface_match.cpp:94
//     results_subset[i]->pose / ->residual / ->numVotes / ->t / ->q      // :102-142
//
// mi355::ppf_match_3d::{ICP, Pose3D} keep those names, argument meanings and the in-place update of
// pose + residual, on plain float buffers (no OpenCV here; the cv::Mat adapter is opencv_adapter.hpp).
// Header-only; link with -lmi355_icp.  Errors surface as mi355::Error (never across the C-ABI itself).
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355_icp.h"

namespace mi355 {

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// A cloud as the reference holds it: rows of >= 3 float32 (x, y, z first), stride_bytes apart
// (12 for cv::Mat Nx3, 24 for Nx6 with normals).
struct CloudView {
  const float* data = nullptr;
  std::size_t n = 0;
  std::size_t stride_bytes = 12;
};

using Pose = std::array<double, 16>;  // row-major 4x4, model -> scene (cv::Matx44d)

inline Pose identity_pose() { return Pose{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

class Engine {
 public:
  explicit Engine(int device_id = 0) {
    const int st = mi355_icp_create(device_id, &ctx_);
    if (st != MI355_ICP_OK) throw Error(st, std::string("mi355_icp_create: ") + mi355_icp_last_error(nullptr));
  }
  // ONE handle over several devices of this process (mi355_icp_create_multi): the node keeps holding one backend by value
  // (pose_estimation.hpp:160-162) and the library spreads jobs or source slabs over the devices itself
  explicit Engine(const std::vector<int>& device_ids) {
    const int st = mi355_icp_create_multi(device_ids.data(), static_cast<int>(device_ids.size()), &ctx_);
    if (st != MI355_ICP_OK) throw Error(st, std::string("mi355_icp_create_multi: ") + mi355_icp_last_error(nullptr));
  }
  ~Engine() { mi355_icp_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  void set_targets(const std::vector<CloudView>& clouds) { upload(clouds, true); }
  void set_sources(const std::vector<CloudView>& clouds) { upload(clouds, false); }

  struct Result {
    std::vector<Pose> poses;
    std::vector<double> residual;
    std::vector<int> iterations, status, converged;
    std::string note;  // what mi355_icp_last_error() says after a call that returned MI355_ICP_OK: empty, or e.g. "a
                       // non-finite distance was met" (coordinates or a pose overflowing float32) - the answer is the oracle's
                       // for that case, but a caller will want to know
  };
  // the engine's note after the last call (see Result::note)
  std::string last_note() const { return mi355_icp_last_error(ctx_); }

  Result align(const std::vector<int>& source_ids, const std::vector<int>& target_ids, const std::vector<Pose>& init,
               const mi355_icp_params& params) {
    const int n = static_cast<int>(init.size());
    if (static_cast<int>(source_ids.size()) != n || static_cast<int>(target_ids.size()) != n)
      throw Error(MI355_ICP_ERR_INVALID_ARGUMENT, "align: id/pose count mismatch");
    Result r;
    r.poses.resize(n);
    r.residual.resize(n);
    r.iterations.resize(n);
    r.status.resize(n);
    r.converged.resize(n);
    check(mi355_icp_align(ctx_, n, source_ids.data(), target_ids.data(), init.empty() ? nullptr : init[0].data(), &params,
                          n ? r.poses[0].data() : nullptr, r.residual.data(), r.iterations.data(), r.status.data(),
                          r.converged.data()),
          "mi355_icp_align");
    r.note = last_note();
    return r;
  }

  // --- scene preparation on the device (pose_estimation.cpp:229-276 and opencv_surface_match.cpp:57-59) ---
  void scene_load(const CloudView& raw) { check(mi355_icp_scene_load(ctx_, raw.data, raw.n, raw.stride_bytes), "mi355_icp_scene_load"); }
  std::size_t scene_filter(const std::vector<mi355_scene_filter>& filters) {
    std::size_t kept = 0;
    check(mi355_icp_scene_filter(ctx_, static_cast<int>(filters.size()), filters.data(), &kept), "mi355_icp_scene_filter");
    return kept;
  }
  std::vector<float> scene_download() {
    std::size_t n = 0;
    check(mi355_icp_scene_download(ctx_, nullptr, 0, &n), "mi355_icp_scene_download");
    std::vector<float> out(3 * n);
    check(mi355_icp_scene_download(ctx_, out.data(), n, &n), "mi355_icp_scene_download");
    return out;
  }
  // pcl::SACSegmentation as the reference sets it up (pose_estimation.cpp:285-297): plane, RANSAC, optimised coefficients,
  // distance threshold 0.0001, 100 iterations.  Returns a, b, c, d of ax + by + cz + d = 0.
  std::array<float, 4> scene_fit_plane(float distance_threshold = 0.0001f, int max_iterations = 100, std::size_t* inliers = nullptr,
                                       std::uint64_t seed = 0x5EEDC0DEull) {
    mi355_plane_fit_params fp{};
    fp.distance_threshold = distance_threshold;
    fp.max_iterations = max_iterations;
    fp.probability = 0.99;
    fp.seed = seed;
    fp.optimize = 1;
    std::array<float, 4> coeff{};
    check(mi355_icp_scene_fit_plane(ctx_, &fp, coeff.data(), inliers, nullptr), "mi355_icp_scene_fit_plane");
    return coeff;
  }
  void scene_normals(int k, bool flip, const std::array<float, 3>& viewpoint) {
    check(mi355_icp_scene_normals(ctx_, k, flip ? 1 : 0, viewpoint.data(), nullptr, nullptr), "mi355_icp_scene_normals");
  }
  void scene_commit_target() { check(mi355_icp_scene_commit_target(ctx_), "mi355_icp_scene_commit_target"); }

  mi355_icp_ctx* handle() { return ctx_; }

 private:
  void upload(const std::vector<CloudView>& clouds, bool targets) {
    std::vector<const float*> ptr;
    std::vector<std::size_t> cnt;
    std::size_t stride = clouds.empty() ? 12 : clouds[0].stride_bytes;
    for (const auto& c : clouds) {
      if (c.stride_bytes != stride) throw Error(MI355_ICP_ERR_INVALID_ARGUMENT, "clouds of one upload must share a row stride");
      ptr.push_back(c.data);
      cnt.push_back(c.n);
    }
    const int n = static_cast<int>(clouds.size());
    check(targets ? mi355_icp_set_targets(ctx_, n, ptr.data(), cnt.data(), stride)
                  : mi355_icp_set_sources(ctx_, n, ptr.data(), cnt.data(), stride),
          targets ? "mi355_icp_set_targets" : "mi355_icp_set_sources");
  }
  void check(int st, const char* what) {
    if (st != MI355_ICP_OK) throw Error(st, std::string(what) + ": " + mi355_icp_last_error(ctx_));
  }
  mi355_icp_ctx* ctx_ = nullptr;
};

namespace ppf_match_3d {

// Mirror of cv::ppf_match_3d::Pose3D as the reference reads and writes it.
struct Pose3D {
  double alpha = 0, residual = 0, angle = 0;
  std::size_t modelIndex = 0, numVotes = 0;
  Pose pose = identity_pose();
  std::array<double, 3> t{0, 0, 0};
  std::array<double, 4> q{1, 0, 0, 0};  // (w, x, y, z), as cv::Vec4d q in upstream

  void updatePose(const Pose& p) {
    pose = p;
    t = {p[3], p[7], p[11]};
    const double tr = p[0] + p[5] + p[10];
    double w, x, y, z;
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      w = 0.25 * s; x = (p[9] - p[6]) / s; y = (p[2] - p[8]) / s; z = (p[4] - p[1]) / s;
    } else if (p[0] > p[5] && p[0] > p[10]) {
      const double s = std::sqrt(1.0 + p[0] - p[5] - p[10]) * 2;
      w = (p[9] - p[6]) / s; x = 0.25 * s; y = (p[1] + p[4]) / s; z = (p[2] + p[8]) / s;
    } else if (p[5] > p[10]) {
      const double s = std::sqrt(1.0 + p[5] - p[0] - p[10]) * 2;
      w = (p[2] - p[8]) / s; x = (p[1] + p[4]) / s; y = 0.25 * s; z = (p[6] + p[9]) / s;
    } else {
      const double s = std::sqrt(1.0 + p[10] - p[0] - p[5]) * 2;
      w = (p[4] - p[1]) / s; x = (p[2] + p[8]) / s; y = (p[6] + p[9]) / s; z = 0.25 * s;
    }
    const double nrm = std::sqrt(w * w + x * x + y * y + z * z);
    q = {w / nrm, x / nrm, y / nrm, z / nrm};
    angle = 2.0 * std::acos(std::fmin(1.0, std::fmax(-1.0, q[0])));
  }

  // opencv_surface_match.cpp:82 (the caller has normalised q, :71-79)
  void updatePoseQuat(const std::array<double, 4>& nq, const std::array<double, 3>& nt) {
    const double w = nq[0], x = nq[1], y = nq[2], z = nq[3];
    pose = Pose{1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),     nt[0],
                2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),     nt[1],
                2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y), nt[2],
                0, 0, 0, 1};
    q = nq;
    t = nt;
    angle = 2.0 * std::acos(std::fmin(1.0, std::fmax(-1.0, q[0])));
  }
};
using Pose3DPtr = std::shared_ptr<Pose3D>;

// Mirror of cv::ppf_match_3d::ICP: same constructor arguments as the reference call site.
// Mode::Reference (default: it is what the reference's call runs) keeps all four arguments live (pyramid, MAD
// rejection, picky, point-to-plane on the scene normals: mi355_icp_align_reference) and needs the scene as N x 6 rows,
// which is what the reference hands over (opencv_surface_match.cpp:57-59).  It runs with MI355_ICP_FLAG_UPSTREAM -
// upstream's control flow as recalled: model normals in the stop rule, pre-normalisation, cvRound sampling step - so the
// model must be N x 6 rows too, as the reference's models are (opencv_surface_match.cpp:22); setUpstreamFaithful(false)
// selects the round 1-3 form (xyz-only stop rule, step 2^l, residual in cloud units; works with an N x 3 model).
// Mode::PointToPoint must be asked for by name: the engine of BASELINE.json (exact NN + Kabsch; numLevels and
// rejectionScale are recorded only).
class ICP {
 public:
  enum class Mode { PointToPoint, Reference };

  ICP(int iterations, float tolerence = 0.05f, float rejectionScale = 2.5f, int numLevels = 6, int device_id = 0,
      Mode mode = Mode::Reference)
      : ICP(iterations, tolerence, rejectionScale, numLevels, std::make_shared<Engine>(device_id), mode) {}
  // the same on several devices of this process: ICP icp(250, 0.005f, 2.5f, 8, std::vector<int>{0, 1, 2, 3});
  ICP(int iterations, float tolerence, float rejectionScale, int numLevels, const std::vector<int>& device_ids, Mode mode = Mode::Reference)
      : ICP(iterations, tolerence, rejectionScale, numLevels, std::make_shared<Engine>(device_ids), mode) {}
  ICP(int iterations, float tolerence, float rejectionScale, int numLevels, std::shared_ptr<Engine> engine, Mode mode)
      : engine_(std::move(engine)), mode_(mode) {
    const int st = mi355_icp_params_from_reference(iterations, tolerence, rejectionScale, numLevels, &params_);
    if (st != MI355_ICP_OK) throw Error(st, "mi355_icp_params_from_reference");
    ref_params_.iterations = iterations;
    ref_params_.tolerance = tolerence;
    ref_params_.rejection_scale = rejectionScale;
    ref_params_.num_levels = numLevels;
    ref_params_.flags = MI355_ICP_FLAG_UPSTREAM;
  }

  // Refines every pose in place (pose, t, q, residual), returns 0 like upstream.  A candidate whose solve fails keeps
  // its last valid pose (the reference swallows the exception: opencv_surface_match.cpp:144-147).
  int registerModelToScene(const CloudView& srcPC, const CloudView& dstPC, std::vector<Pose3DPtr>& poses) {
    engine_->set_sources({srcPC});
    engine_->set_targets({dstPC});
    return registerUploaded(poses, 0, 0);
  }

  // Same for clouds already resident on the GPU (models persist across requests: opencv_surface_match.hpp:45).
  int registerUploaded(std::vector<Pose3DPtr>& poses, int source_id, int target_id) {
    if (poses.empty()) return 0;
    std::vector<Pose> init;
    for (const auto& p : poses) init.push_back(p->pose);
    const std::vector<int> sid(poses.size(), source_id), tid(poses.size(), target_id);
    if (mode_ == Mode::Reference) {
      const int n = static_cast<int>(poses.size());
      last_ = Engine::Result{};
      last_.poses.resize(n);
      last_.residual.resize(n);
      last_.iterations.resize(n);
      last_.status.resize(n);
      last_.converged.assign(n, 0);
      const int st = mi355_icp_align_reference(engine_->handle(), n, sid.data(), tid.data(), init[0].data(), &ref_params_,
                                               last_.poses[0].data(), last_.residual.data(), last_.iterations.data(),
                                               last_.status.data());
      if (st != MI355_ICP_OK) throw Error(st, std::string("mi355_icp_align_reference: ") + mi355_icp_last_error(engine_->handle()));
      last_.note = engine_->last_note();
    } else {
      last_ = engine_->align(sid, tid, init, params_);
    }
    for (std::size_t k = 0; k < poses.size(); ++k) {
      poses[k]->updatePose(last_.poses[k]);
      poses[k]->residual = last_.residual[k];
    }
    return 0;
  }

  // Answer the nearest-neighbour queries with the exact pruned scan (MI355_ICP_FLAG_PRUNED_SCAN): same poses, less time
  // from ~100k points.  Not upstream API; off by default (the replaced path is specified as brute force).
  void setPrunedScan(bool on) {
    params_.flags = on ? (params_.flags | MI355_ICP_FLAG_PRUNED_SCAN) : (params_.flags & ~MI355_ICP_FLAG_PRUNED_SCAN);
    ref_params_.flags = on ? (ref_params_.flags | MI355_ICP_FLAG_PRUNED_SCAN) : (ref_params_.flags & ~MI355_ICP_FLAG_PRUNED_SCAN);
  }

  // Reference mode: upstream's control flow (default) or the round 1-3 form.  Not upstream API.
  void setUpstreamFaithful(bool on) {
    ref_params_.flags = on ? (ref_params_.flags | MI355_ICP_FLAG_UPSTREAM) : (ref_params_.flags & ~MI355_ICP_FLAG_UPSTREAM);
  }

  Engine& engine() { return *engine_; }
  const mi355_icp_params& params() const { return params_; }
  const Engine::Result& last_result() const { return last_; }

 private:
  std::shared_ptr<Engine> engine_;
  Mode mode_;
  mi355_icp_params params_{};
  mi355_icp_reference_params ref_params_{};
  Engine::Result last_;
};

// Best-candidate rule of opencv_surface_match.cpp:102-123.
inline std::size_t select_best_pose(const std::vector<Pose3DPtr>& subset, std::size_t n_results_total) {
  std::size_t max_votes = 0, best_idx = 0;
  float min_res = 10000.0f;
  for (std::size_t i = 0; i < subset.size(); ++i) {
    if (subset[i]->numVotes > max_votes) {
      max_votes = subset[i]->numVotes;
      best_idx = i;
    }
    if (n_results_total > 5 && subset[i]->residual < min_res && subset[i]->numVotes > 400) {
      min_res = static_cast<float>(subset[i]->residual);
      best_idx = i;
    }
  }
  return best_idx;
}

// The ICP half of OpenCVSurfaceMatch::find_object_in_scene (opencv_surface_match.cpp:85-142): first <= 6
// candidates, one batched refine, best-pose rule, 7-float packing (reference_packing reproduces :138-141).
inline std::vector<float> refine_candidates(ICP& icp, const CloudView& model, const CloudView& scene,
                                            std::vector<Pose3DPtr>& results, bool reference_packing = true,
                                            std::size_t* best_out = nullptr) {
  std::vector<Pose3DPtr> subset(results.begin(), results.begin() + static_cast<std::ptrdiff_t>(results.size() > 6 ? 6 : results.size()));
  icp.registerModelToScene(model, scene, subset);
  const std::size_t best = select_best_pose(subset, results.size());
  if (best_out) *best_out = best;
  std::vector<float> pose(7, 0.0f);
  if (!results.empty()) mi355_icp_pose_to_pose7(results[best]->pose.data(), pose.data(), reference_packing ? 1 : 0);
  return pose;
}

}  // namespace ppf_match_3d
}  // namespace mi355

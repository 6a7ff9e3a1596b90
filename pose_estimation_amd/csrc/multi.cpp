// multi.cpp — ONE handle over several devices of ONE process (mi355_icp_create_multi, include/mi355_icp.h).
//
// Why: the reference holds its ICP backend by value inside one node process (pose_estimation/include/pose_estimation.hpp:160-162,
// one component container: launch/pose_estimation.launch.py:17-35).  A rank-per-process design needs a launcher that
// process does not have; SURVEY.md 8(b) asks for `create(device_ids[], n_devices, &handle)` with the multi-GPU threads
// internal to the library.  This file is that: pure host C++ above the single-device C-ABI - no kernel, no HIP call, no RCCL,
// no torch.  One single-device context ("sub-context") and one worker thread per entry of device_ids; every public call
// on the handle is fanned out to them and joined.
//
// How work is spread (decided per call):
//   * at least as many jobs as devices (batched bin-picking, BASELINE config[3]; a request's candidates on 2 GPUs):
//     by JOB - contiguous runs of jobs per device, sources and targets replicated, no exchange at all; every job is computed
//     exactly as a single-device engine computes it (same bits);
//   * fewer jobs than devices (BASELINE config[4]: one 8M-point model): by SOURCE SLAB - device d holds slab d of every source
//     cloud (mi355_icp_slab_range), the target is replicated, and the sub-contexts run the slab-sharded loops of
//     icp_engine.hip over an in-process all-gather (below) attached through mi355_icp_comm_attach_external: per pass each
//     device's rows of 17 (reference mode: 30) doubles go to pinned host memory and are added in DEVICE ORDER on the host - the
//     collective-free alternative SURVEY.md 5 asked to be measured against RCCL.  Same arithmetic as a gloo or RCCL run of the
//     same rank count, bit for bit.
// The source clouds are kept on the host (models are small and persistent: opencv_surface_match.hpp:45) so that the layout
// can follow the calls; a service that always makes the same kind of call uploads once.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "icp_internal.h"

namespace mi355 {

namespace {

struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> task;
  bool has_task = false, done = false, quit = false;
  int result = MI355_ICP_OK;
};

}  // namespace

struct MultiState {
  std::vector<mi355_icp_ctx*> sub;
  std::vector<std::unique_ptr<Worker>> workers;
  // the in-process all-gather: every rank writes its own slot of the exchange's parity, all meet at a barrier, every rank
  // reads all slots.  Exchange k + 2 reuses the slots of exchange k: to get there a rank has passed barrier k + 1, which
  // every rank reaches only after it has finished reading exchange k.
  struct CommUser { MultiState* ms = nullptr; int rank = 0; uint32_t seq = 0; };
  std::vector<CommUser> users;
  std::vector<std::vector<unsigned char>> slots[2];
  std::atomic<uint32_t> arrived{0}, generation{0};
  std::atomic<bool> abort{false};
  // the source clouds as the caller uploaded them (rows of src_stride floats)
  std::vector<std::vector<float>> src_rows;
  std::vector<size_t> src_n;
  size_t src_stride_bytes = 12;
  enum Layout { kNone, kFull, kSlab } layout = kNone;
  size_t n_targets = 0;
  std::string err;
};

namespace {

int n_dev(const MultiState* ms) { return static_cast<int>(ms->sub.size()); }

bool barrier(MultiState* ms) {  // false: another rank gave up (its call failed); never blocks for ever
  const uint32_t gen = ms->generation.load(std::memory_order_acquire);
  if (ms->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(n_dev(ms))) {
    ms->arrived.store(0, std::memory_order_relaxed);
    ms->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  for (uint32_t spin = 0; ms->generation.load(std::memory_order_acquire) == gen; ++spin) {
    if (ms->abort.load(std::memory_order_acquire)) return false;
    if (spin > 20000) std::this_thread::yield();  // a pass is microseconds: spin first, then be polite (long scans)
  }
  return true;
}

// mi355_icp_allgather_fn of the sub-contexts
int multi_allgather(void* user, const void* send, void* recv, size_t bytes) {
  auto* u = static_cast<MultiState::CommUser*>(user);
  MultiState* ms = u->ms;
  const uint32_t parity = (u->seq++) & 1u;
  std::vector<unsigned char>& mine = ms->slots[parity][static_cast<size_t>(u->rank)];
  mine.assign(static_cast<const unsigned char*>(send), static_cast<const unsigned char*>(send) + bytes);
  if (!barrier(ms)) return 1;
  for (int r = 0; r < n_dev(ms); ++r) {
    const std::vector<unsigned char>& s = ms->slots[parity][static_cast<size_t>(r)];
    if (s.size() != bytes) {  // the ranks are not in the same exchange: a bug, never a hang
      ms->abort.store(true, std::memory_order_release);
      return 1;
    }
    std::memcpy(static_cast<unsigned char*>(recv) + static_cast<size_t>(r) * bytes, s.data(), bytes);
  }
  return 0;
}

void worker_loop(Worker* w) {
  for (;;) {
    std::function<int()> task;
    {
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [&] { return w->has_task || w->quit; });
      if (w->quit) return;
      task = std::move(w->task);
      w->has_task = false;
    }
    const int r = task();
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->result = r;
      w->done = true;
    }
    w->cv.notify_all();
  }
}

// f(rank) on every device's worker at once; the first failure's status and text are the call's
int run_all(MultiState* ms, const std::function<int(int)>& f) {
  ms->abort.store(false, std::memory_order_release);
  ms->arrived.store(0, std::memory_order_relaxed);
  ms->generation.store(0, std::memory_order_relaxed);
  for (auto& u : ms->users) u.seq = 0;
  const int n = n_dev(ms);
  std::vector<int> st(static_cast<size_t>(n), MI355_ICP_OK);
  if (n == 1) {
    st[0] = f(0);
  } else {
    for (int r = 0; r < n; ++r) {
      Worker* w = ms->workers[static_cast<size_t>(r)].get();
      {
        std::lock_guard<std::mutex> lk(w->m);
        w->task = [ms, &f, r]() {
          const int s = f(r);
          if (s != MI355_ICP_OK) ms->abort.store(true, std::memory_order_release);  // the others must not wait for this rank
          return s;
        };
        w->done = false;
        w->has_task = true;
      }
      w->cv.notify_all();
    }
    for (int r = 0; r < n; ++r) {
      Worker* w = ms->workers[static_cast<size_t>(r)].get();
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [&] { return w->done; });
      st[static_cast<size_t>(r)] = w->result;
    }
  }
  // report the failure that started it: a rank that only gave up because another one did says ERR_COMM
  int first = -1;
  for (int r = 0; r < n; ++r)
    if (st[r] != MI355_ICP_OK && (first < 0 || (st[first] == MI355_ICP_ERR_COMM && st[r] != MI355_ICP_ERR_COMM))) first = r;
  if (first < 0) return MI355_ICP_OK;
  ms->err = std::string("device slot ") + std::to_string(first) + ": " + mi355_icp_last_error(ms->sub[static_cast<size_t>(first)]);
  return st[first];
}

int fail(MultiState* ms, int code, const std::string& msg) {
  ms->err = msg;
  return code;
}

// brings the source clouds onto the devices in the layout the next call needs
int ensure_layout(MultiState* ms, MultiState::Layout want) {
  if (ms->layout == want) return MI355_ICP_OK;
  if (ms->src_rows.empty()) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "no source clouds uploaded (mi355_icp_set_sources)");
  const int n = n_dev(ms);
  const size_t fstride = ms->src_stride_bytes / 4;
  const int st = run_all(ms, [&](int r) {
    mi355_icp_ctx* c = ms->sub[static_cast<size_t>(r)];
    int s = mi355_icp_comm_detach(c);
    if (s != MI355_ICP_OK) return s;
    std::vector<const float*> ptr;
    std::vector<size_t> cnt;
    for (size_t k = 0; k < ms->src_rows.size(); ++k) {
      size_t begin = 0, count = ms->src_n[k];
      if (want == MultiState::kSlab) (void)mi355_icp_slab_range(ms->src_n[k], r, n, &begin, &count);
      ptr.push_back(count ? ms->src_rows[k].data() + begin * fstride : nullptr);
      cnt.push_back(count);
    }
    s = mi355_icp_set_sources(c, static_cast<int>(ptr.size()), ptr.data(), cnt.data(), ms->src_stride_bytes);
    if (s != MI355_ICP_OK) return s;
    if (want == MultiState::kSlab) s = mi355_icp_comm_attach_external(c, r, n, multi_allgather, &ms->users[static_cast<size_t>(r)]);
    return s;
  });
  ms->layout = st == MI355_ICP_OK ? want : MultiState::kNone;
  return st;
}

// contiguous runs of jobs per device: device r takes [first, first + count)
void job_range(int n_jobs, int r, int n, int* first, int* count) {
  const int base = n_jobs / n, extra = n_jobs % n;
  *first = r * base + (r < extra ? r : extra);
  *count = base + (r < extra ? 1 : 0);
}

}  // namespace

int multi_create(const int* device_ids, int n_devices, MultiState** out, std::string* err) {
  auto ms = std::make_unique<MultiState>();
  for (int r = 0; r < n_devices; ++r) {
    mi355_icp_ctx* c = nullptr;
    const int st = mi355_icp_create(device_ids[r], &c);
    if (st != MI355_ICP_OK) {
      *err = std::string("device slot ") + std::to_string(r) + " (device " + std::to_string(device_ids[r]) + "): " + mi355_icp_last_error(nullptr);
      for (mi355_icp_ctx* s : ms->sub) mi355_icp_destroy(s);
      return st;
    }
    ms->sub.push_back(c);
  }
  ms->users.resize(static_cast<size_t>(n_devices));
  for (int r = 0; r < n_devices; ++r) ms->users[static_cast<size_t>(r)] = MultiState::CommUser{ms.get(), r, 0};
  for (auto& s : ms->slots) s.resize(static_cast<size_t>(n_devices));
  if (n_devices > 1)
    for (int r = 0; r < n_devices; ++r) {
      ms->workers.push_back(std::make_unique<Worker>());
      Worker* w = ms->workers.back().get();
      w->th = std::thread(worker_loop, w);
    }
  *out = ms.release();
  return MI355_ICP_OK;
}

void multi_destroy(MultiState* ms) {
  if (!ms) return;
  for (auto& w : ms->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
  for (mi355_icp_ctx* c : ms->sub) mi355_icp_destroy(c);
  delete ms;
}

const char* multi_last_error(const MultiState* ms) { return ms->err.c_str(); }
int multi_devices(const MultiState* ms) { return n_dev(ms); }
mi355_icp_ctx* multi_sub(MultiState* ms, int slot) { return ms->sub[static_cast<size_t>(slot)]; }

// a call every device makes for itself, results taken from device slot 0 (uploads of the scene, scene preparation, knobs)
int multi_broadcast(MultiState* ms, const std::function<int(mi355_icp_ctx*, int)>& f) {
  ms->err.clear();
  return run_all(ms, [&](int r) { return f(ms->sub[static_cast<size_t>(r)], r); });
}

int multi_set_targets(MultiState* ms, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes) {
  ms->err.clear();
  const int st = run_all(ms, [&](int r) { return mi355_icp_set_targets(ms->sub[static_cast<size_t>(r)], n_clouds, xyz, n_points, stride_bytes); });
  if (st == MI355_ICP_OK) ms->n_targets = static_cast<size_t>(n_clouds);
  return st;
}

int multi_set_sources(MultiState* ms, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes) {
  ms->err.clear();
  if (n_clouds < 1 || !xyz || !n_points) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "set_sources: no clouds");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "set_sources: stride_bytes must be a multiple of 4 and >= 12");
  for (int k = 0; k < n_clouds; ++k)
    if (n_points[k] && !xyz[k]) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "set_sources: null cloud pointer");
  // every device gets the whole clouds first (this is also where non-finite input is refused); the host copy lets a later
  // call re-lay them out as slabs
  ms->layout = MultiState::kNone;
  const int st = run_all(ms, [&](int r) {
    const int s = mi355_icp_comm_detach(ms->sub[static_cast<size_t>(r)]);
    return s != MI355_ICP_OK ? s : mi355_icp_set_sources(ms->sub[static_cast<size_t>(r)], n_clouds, xyz, n_points, stride_bytes);
  });
  if (st != MI355_ICP_OK) {
    ms->src_rows.clear();
    ms->src_n.clear();
    return st;
  }
  const size_t fstride = stride_bytes / 4;
  ms->src_rows.assign(static_cast<size_t>(n_clouds), {});
  ms->src_n.assign(n_points, n_points + n_clouds);
  ms->src_stride_bytes = stride_bytes;
  for (int k = 0; k < n_clouds; ++k)
    if (n_points[k]) ms->src_rows[static_cast<size_t>(k)].assign(xyz[k], xyz[k] + n_points[k] * fstride);
  ms->layout = MultiState::kFull;
  return MI355_ICP_OK;
}

namespace {
// by job when every device gets at least one
bool shard_by_job(const MultiState* ms, int n_jobs) { return n_jobs >= n_dev(ms); }
}  // namespace

int multi_align(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* init_poses, const mi355_icp_params* params,
                double* out_poses, double* out_residual, int* out_iterations, int* out_status, int* out_converged) {
  ms->err.clear();
  if (n_jobs < 1 || !init_poses || !params) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "align: bad arguments");
  const int n = n_dev(ms);
  if (shard_by_job(ms, n_jobs)) {
    if (const int st = ensure_layout(ms, MultiState::kFull); st != MI355_ICP_OK) return st;
    return run_all(ms, [&](int r) {
      int first = 0, count = 0;
      job_range(n_jobs, r, n, &first, &count);
      if (!count) return static_cast<int>(MI355_ICP_OK);
      const size_t f = static_cast<size_t>(first);
      return mi355_icp_align(ms->sub[static_cast<size_t>(r)], count, source_ids ? source_ids + f : nullptr, target_ids ? target_ids + f : nullptr,
                             init_poses + 16 * f, params, out_poses ? out_poses + 16 * f : nullptr, out_residual ? out_residual + f : nullptr,
                             out_iterations ? out_iterations + f : nullptr, out_status ? out_status + f : nullptr,
                             out_converged ? out_converged + f : nullptr);
    });
  }
  if (const int st = ensure_layout(ms, MultiState::kSlab); st != MI355_ICP_OK) return st;
  return run_all(ms, [&](int r) {  // every device returns the same bits: slot 0's are the caller's
    const bool lead = r == 0;
    return mi355_icp_align(ms->sub[static_cast<size_t>(r)], n_jobs, source_ids, target_ids, init_poses, params, lead ? out_poses : nullptr,
                           lead ? out_residual : nullptr, lead ? out_iterations : nullptr, lead ? out_status : nullptr, lead ? out_converged : nullptr);
  });
}

int multi_align_reference(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* init_poses,
                          const mi355_icp_reference_params* rp, double* out_poses, double* out_residual, int* out_passes, int* out_status) {
  ms->err.clear();
  if (n_jobs < 1 || !init_poses || !rp) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: bad arguments");
  const int n = n_dev(ms);
  if (shard_by_job(ms, n_jobs)) {
    if (const int st = ensure_layout(ms, MultiState::kFull); st != MI355_ICP_OK) return st;
    return run_all(ms, [&](int r) {
      int first = 0, count = 0;
      job_range(n_jobs, r, n, &first, &count);
      if (!count) return static_cast<int>(MI355_ICP_OK);
      const size_t f = static_cast<size_t>(first);
      return mi355_icp_align_reference(ms->sub[static_cast<size_t>(r)], count, source_ids ? source_ids + f : nullptr, target_ids ? target_ids + f : nullptr,
                                       init_poses + 16 * f, rp, out_poses ? out_poses + 16 * f : nullptr, out_residual ? out_residual + f : nullptr,
                                       out_passes ? out_passes + f : nullptr, out_status ? out_status + f : nullptr);
    });
  }
  if (const int st = ensure_layout(ms, MultiState::kSlab); st != MI355_ICP_OK) return st;
  return run_all(ms, [&](int r) {
    const bool lead = r == 0;
    return mi355_icp_align_reference(ms->sub[static_cast<size_t>(r)], n_jobs, source_ids, target_ids, init_poses, rp, lead ? out_poses : nullptr,
                                     lead ? out_residual : nullptr, lead ? out_passes : nullptr, lead ? out_status : nullptr);
  });
}

// One correspondence pass.  By job: every device answers its run of jobs into the caller's arrays.  By slab: device d answers
// slab d of every job; the pieces are put back in source order and the devices' sums are added in device order.
int multi_correspond(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* poses, float max_corr,
                     int32_t* out_index, float* out_dist2, double* out_sums) {
  ms->err.clear();
  if (n_jobs < 1 || !poses) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "correspond: bad arguments");
  if (ms->src_rows.empty()) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "no source clouds uploaded (mi355_icp_set_sources)");
  const int n = n_dev(ms);
  for (int k = 0; k < n_jobs; ++k) {
    const int s = source_ids ? source_ids[k] : 0;
    if (s < 0 || static_cast<size_t>(s) >= ms->src_n.size()) return fail(ms, MI355_ICP_ERR_INVALID_ARGUMENT, "source id out of range");
  }
  auto job_points = [&](int k) { return ms->src_n[static_cast<size_t>(source_ids ? source_ids[k] : 0)]; };
  if (shard_by_job(ms, n_jobs)) {
    if (const int st = ensure_layout(ms, MultiState::kFull); st != MI355_ICP_OK) return st;
    std::vector<size_t> cursor(static_cast<size_t>(n_jobs) + 1, 0);
    for (int k = 0; k < n_jobs; ++k) cursor[static_cast<size_t>(k) + 1] = cursor[static_cast<size_t>(k)] + job_points(k);
    return run_all(ms, [&](int r) {
      int first = 0, count = 0;
      job_range(n_jobs, r, n, &first, &count);
      if (!count) return static_cast<int>(MI355_ICP_OK);
      const size_t f = static_cast<size_t>(first);
      return mi355_icp_correspond(ms->sub[static_cast<size_t>(r)], count, source_ids ? source_ids + f : nullptr, target_ids ? target_ids + f : nullptr,
                                  poses + 16 * f, max_corr, out_index ? out_index + cursor[f] : nullptr, out_dist2 ? out_dist2 + cursor[f] : nullptr,
                                  out_sums ? out_sums + MI355_ICP_NSUMS * f : nullptr);
    });
  }
  if (const int st = ensure_layout(ms, MultiState::kSlab); st != MI355_ICP_OK) return st;
  std::vector<std::vector<int32_t>> idx(static_cast<size_t>(n));
  std::vector<std::vector<float>> d2(static_cast<size_t>(n));
  std::vector<std::vector<double>> sums(static_cast<size_t>(n));
  const int st = run_all(ms, [&](int r) {
    size_t total = 0;
    for (int k = 0; k < n_jobs; ++k) {
      size_t b = 0, c = 0;
      (void)mi355_icp_slab_range(job_points(k), r, n, &b, &c);
      total += c;
    }
    if (out_index) idx[static_cast<size_t>(r)].resize(total + 1);
    if (out_dist2) d2[static_cast<size_t>(r)].resize(total + 1);
    sums[static_cast<size_t>(r)].resize(static_cast<size_t>(n_jobs) * MI355_ICP_NSUMS);
    return mi355_icp_correspond(ms->sub[static_cast<size_t>(r)], n_jobs, source_ids, target_ids, poses, max_corr,
                                out_index ? idx[static_cast<size_t>(r)].data() : nullptr, out_dist2 ? d2[static_cast<size_t>(r)].data() : nullptr,
                                sums[static_cast<size_t>(r)].data());
  });
  if (st != MI355_ICP_OK) return st;
  std::vector<size_t> piece(static_cast<size_t>(n), 0);  // read cursor into each device's outputs
  size_t out = 0;
  for (int k = 0; k < n_jobs; ++k)
    for (int r = 0; r < n; ++r) {
      size_t b = 0, c = 0;
      (void)mi355_icp_slab_range(job_points(k), r, n, &b, &c);
      if (out_index && c) std::memcpy(out_index + out, idx[static_cast<size_t>(r)].data() + piece[static_cast<size_t>(r)], c * sizeof(int32_t));
      if (out_dist2 && c) std::memcpy(out_dist2 + out, d2[static_cast<size_t>(r)].data() + piece[static_cast<size_t>(r)], c * sizeof(float));
      piece[static_cast<size_t>(r)] += c;
      out += c;
    }
  if (out_sums)
    for (size_t i = 0; i < static_cast<size_t>(n_jobs) * MI355_ICP_NSUMS; ++i) {
      double acc = sums[0][i];
      for (int r = 1; r < n; ++r) acc += sums[static_cast<size_t>(r)][i];  // device order, like the align loop's exchange
      out_sums[i] = acc;
    }
  return MI355_ICP_OK;
}

// diagnostics that look at ONE context's buffers (reference_pass, filter bounds, normalisation): device slot 0, which holds the
// whole clouds only in the by-job layout
int multi_single_context(MultiState* ms, mi355_icp_ctx** out) {
  ms->err.clear();
  if (const int st = ensure_layout(ms, MultiState::kFull); st != MI355_ICP_OK) return st;
  *out = ms->sub[0];
  return MI355_ICP_OK;
}

}  // namespace mi355

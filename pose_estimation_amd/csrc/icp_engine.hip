// icp_engine.hip — host side of the engine behind include/mi355_icp.h:
// device-resident clouds (SoA planes), the per-call planner, the ICP iteration
// loop (launch scan -> resolve/accumulate -> final reduce -> [cross-rank exchange, comm.hip]
// -> 17 doubles to the host -> Kabsch + convergence), and the C-ABI glue.
//
// Replaces, in the reference, the object constructed at
// pose_estimation/src/opencv_surface_match.cpp:85 and the call at :94.
// No CPU compute fallback exists: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "icp_internal.h"

namespace mi355 {

namespace {

thread_local std::string g_create_error;

struct CloudEntry {
  uint32_t off;     // first point in the planes
  uint32_t n;       // real points
  uint32_t padded;  // slots occupied
};

// Device buffers of the cloud copies (planes, pyramid levels, sorted copies) come from a size-classed cache.  A service
// commits a new scene per request, a few per cent larger or smaller than the last one, and every hipMalloc / hipFree pair
// costs ~0.1 ms plus a device-wide synchronisation - a request at the reference's shape went through ~40 of them.
// Everything that touches these buffers is ordered on the engine's one stream, so a buffer given back while launches that
// use it are still queued may be handed out again at once: the next user is queued behind them.
struct BufferPool {
  std::multimap<size_t, void*> idle;            // size class -> buffer
  std::unordered_map<void*, size_t> class_of;   // every buffer the pool ever handed out and still owns
  size_t idle_bytes = 0;
  static size_t size_class(size_t bytes) {      // steps of 64 KiB up to 1 MiB, of one eighth of the power of two above
    if (bytes <= (1u << 20)) return (std::max<size_t>(bytes, 1) + 0xFFFFu) & ~static_cast<size_t>(0xFFFFu);
    size_t p = 1;
    while ((p << 1) <= bytes) p <<= 1;
    const size_t step = p >> 3;
    return (bytes + step - 1) / step * step;
  }
  hipError_t take(void** out, size_t bytes) {
    const size_t cls = size_class(bytes);
    auto it = idle.find(cls);
    if (it != idle.end()) {
      *out = it->second;
      idle.erase(it);
      idle_bytes -= cls;
      return hipSuccess;
    }
    const hipError_t e = hipMalloc(out, cls);
    if (e == hipSuccess) class_of[*out] = cls;
    return e;
  }
  void give(void* ptr) {
    if (!ptr) return;
    auto it = class_of.find(ptr);
    if (it == class_of.end()) { (void)hipFree(ptr); return; }  // not ours (never happens)
    if (idle_bytes + it->second > (static_cast<size_t>(8) << 30)) {  // keep at most 8 GiB idle
      (void)hipDeviceSynchronize();
      (void)hipFree(ptr);
      class_of.erase(it);
      return;
    }
    idle.emplace(it->second, ptr);
    idle_bytes += it->second;
  }
  void release_all() {  // the caller has drained the device; buffers still held by their users are freed through give() later
    for (auto& kv : idle) { (void)hipFree(kv.second); class_of.erase(kv.second); }
    idle.clear();
    idle_bytes = 0;
  }
};
// Expansion-filter copy of a group of target clouds (built on demand, xtarget_kernels.hip): Morton-sorted, centred per
// 1024-point tile.  planes: tcx | tcy | tcz | tt | sox | soy | soz, each `stride` floats; perm: sorted position ->
// original index; table: per tile (centre x, y, z, r^2); bf: bf16 operand image of the matrix-core filter
// (kXbTileVecs uint4 per tile).
struct XCopy {
  float* planes = nullptr;
  uint32_t* perm = nullptr;
  float* table = nullptr;
  void* bf = nullptr;
  float* bf_table = nullptr;  // per tile: (centre x, y, z, r^2) the bf16 operand image was written around - shared by kXbCentreGroup tiles
  float* sub = nullptr;  // per 64-point run (centre x, y, z, r^2) on the uncentred coordinates: the pruned scans
  float* super = nullptr;  // per 16 tiles: a sphere around their spheres (first level of the pruned scans' tile search)
  bool valid = false;
};

struct LevelPlanes {  // compacted copy of every cloud of a set that keeps every step-th point (rows 0, step, 2*step, ...), step >= 2
  uint32_t step = 0;   // 2^l in the round 1-3 form; cvRound(n / cvRound(n / 2^l)) of the model size in the upstream-faithful one
  float* planes = nullptr;
  size_t stride = 0;
  std::vector<CloudEntry> clouds;
  std::vector<uint32_t> first;       // per cloud: the level-0 row (of this rank's slab) the level's first point is; then every step-th
  std::vector<uint32_t> index_base;  // per cloud: index, in the WHOLE level cloud, of this rank's first level point (0 unless sharded)
  std::vector<uint32_t> n_total;     // per cloud: points of the whole level cloud over all ranks (= clouds[k].n unless sharded)
  XCopy xt;  // targets only
  void* rows = nullptr;  // sets with normals: the 32-byte records (x, y, z, nx | ny, nz, 0, 0) the point-to-plane accumulate reads -
                         // the target's always, the source's in the upstream-faithful variant (its normals enter fval)
  DevicePlanes view() const { return DevicePlanes{planes, planes + stride, planes + 2 * stride}; }
};

struct CloudSet {
  float* planes = nullptr;   // x | y | z, each `stride` floats
  float* normals = nullptr;  // nx | ny | nz in the same layout, only when uploaded with rows of >= 6 floats
  size_t stride = 0;
  std::vector<CloudEntry> clouds;
  std::vector<LevelPlanes> levels;  // subsampled copies by step, built on demand by reference mode; "slot" 0 is the set itself
                                    // (step 1), slot k + 1 is levels[k]
  std::vector<double> sum_xyz;      // per cloud: sum of x, y, z over its points (this rank's slab), taken at upload - the
                                    // means of upstream's pre-normalisation (MI355_ICP_FLAG_UPSTREAM)
  // slab-sharded reference mode: the slab this rank holds of every cloud - [begin, begin + n) of `total` points - which
  // decides WHICH of its points belong to a pyramid level (global indices 0, step, 2 step, ...).  Empty = not sharded.
  std::vector<uint64_t> shard_begin, shard_total;
  XCopy xt;  // level-0 expansion-filter copy (targets only)
  void* rows = nullptr;  // level-0 records for the point-to-plane accumulate (sets with normals; built on demand)
  DevicePlanes view() const { return DevicePlanes{planes, planes + stride, planes + 2 * stride}; }
  DevicePlanes normal_view() const { return DevicePlanes{normals, normals + stride, normals + 2 * stride}; }
};

struct Plan {
  std::vector<JobDesc> jobs;
  uint32_t out_total = 0;      // padded output slots over all jobs (= key stride)
  uint32_t partial_total = 0;  // per-block partial rows over all jobs
  uint32_t max_split = 1;
  uint32_t max_blocks_x = 0;   // scan grid.x
  uint32_t max_src_blocks = 0; // resolve grid.x
  int scan_ppl = 8;            // source points per lane the scan kernel is launched with
  bool fused = false;          // fused-filter scan + exact resolve (large launches) instead of the all-exact scan
  bool xfilter = false;        // expansion-filter scan on the sorted/centred target copy (largest launches); implies fused
  int xkind = 0;               // 0 VALU kernel, 1 f32 MFMA kernel, 2 both side by side, 3 split-bf16 MFMA kernel
  bool pruned = false;         // OPTIONAL exact pruned scan (MI355_ICP_SCAN_MODE=pruned), never chosen by the planner
  bool merged_keys = false;    // small all-exact launches (JobDesc::sub_split): one atomically merged key row, cleared per pass
  bool classed = false;        // one-round matrix-core launches: class-weighted runs of tiles (JobDesc::first_class)
  bool queued = false;         // matrix-core launches of a few rounds: teams of workgroups drawing groups of tiles (kXbLayoutQueue)
  uint32_t grid3_x = 0, grid3_z = 0;  // the 3-D grid of the classed / queued layouts: source blocks, splits
  bool sorted_src = false;     // the matrix-core scan and its resolve walk the SOURCE cloud in Morton order too (below)
  uint32_t key_rows() const { return (xkind == 3 ? 2u : 1u) * max_split; }  // key rows per rank (wave halves publish separately)
  uint32_t key_ranks() const { return xfilter && xkind != 3 ? 4u : 3u; }       // the VALU expansion kernels also publish U
};

void free_xcopy(BufferPool& pool_ref, XCopy& x) {  // back to the pool (stream-ordered: no drain needed)
  BufferPool* pool = &pool_ref;
  pool->give(x.planes);
  pool->give(x.perm);
  pool->give(x.table);
  pool->give(x.bf);
  pool->give(x.bf_table);
  pool->give(x.sub);
  pool->give(x.super);
  x = XCopy{};
}

uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = std::getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  unsigned long x = std::strtoul(v, &end, 10);
  return (end && *end == 0 && x > 0) ? static_cast<uint32_t>(x) : dflt;
}

}  // namespace

}  // namespace mi355

using namespace mi355;

constexpr size_t kReadyWords = 1024;  // jobs of one pass that can publish for themselves (PassTail::ready); more take the copy path

struct mi355_icp_ctx {
  mi355::MultiState* multi = nullptr;  // mi355_icp_create_multi: this handle only fans calls out to its per-device contexts (multi.cpp)
  BufferPool pool;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  CloudSet targets, sources;

  // per-call work buffers (grown on demand, never shrunk)
  unsigned long long* keys = nullptr; size_t keys_cap = 0;
  int32_t* out_index = nullptr;       size_t out_cap = 0;
  float* out_dist2 = nullptr;
  double* partials = nullptr;         size_t partials_cap = 0;
  double* sums_dev = nullptr;         size_t jobs_cap = 0;
  JobDesc* jobs_dev = nullptr;
  uint32_t* error_flag = nullptr;
  JobDesc* jobs_host = nullptr;  // pinned
  double* sums_host = nullptr;   // pinned, mapped: the last workgroup of a single-GPU pass writes the result here itself (PassTail)
  double* sums_mapped = nullptr; // sums_host as the device addresses it
  uint32_t* p2p_done = nullptr;  // device, [jobs_cap], zero between passes (PassTail::done of the point-to-point pass)
  uint32_t* flag_host = nullptr; // pinned

  // reference-mode work buffers
  unsigned long long* owner = nullptr; size_t owner_cap = 0;
  unsigned long long* offer_block = nullptr; unsigned long long* offer_all = nullptr; size_t offer_words_cap = 0;  // slab-sharded: this rank's offer list / all ranks'
  bool picky_offers = true;            // MI355_ICP_PICKY_OFFERS=0: always min-reduce the whole owner table across ranks (A/B)
  uint32_t owner_epoch = 0;  // pass tag of the picky owner keys (picky_key): 1..63, 0 = the table must be cleared first
  uint8_t* selected = nullptr;         size_t selected_cap = 0;
  double* ref_partials = nullptr;      size_t ref_partials_cap = 0;
  double* ref_sums_dev = nullptr;      size_t ref_jobs_cap = 0;
  float* thr_dev = nullptr;
  uint32_t* select_work = nullptr;  // per job: 2048-bin histogram + selection state of the multi-workgroup rank select
  RefJobDesc* rjobs_dev = nullptr;
  RefJobDesc* rjobs_host = nullptr;    // pinned
  double* ref_sums_host = nullptr;     // pinned, mapped: the last workgroup of a single-GPU pass writes the result here itself
  float* thr_host = nullptr;           // pinned
  double* ref_sums_mapped = nullptr;   // ref_sums_host as the device addresses it
  uint32_t* pass_done = nullptr;       // device, [ref_jobs_cap], zero between passes (PassTail::done)
  uint32_t* ready_host = nullptr;      // pinned, mapped: one word per job, where a pass publishes (sequence number << 1 | error bit)
  uint32_t* ready_mapped = nullptr;
  uint32_t pass_seq = 0;
  uint32_t sync_seq = 0;               // sync_pass: the last value published to ready_host[kReadyWords]
  unsigned char* stage_host = nullptr;  // pinned, two halves of kStageHalf bytes: results travel through it to the caller's memory (copy_to_user)
  bool error_word_known = false;       // the last pass brought the call's error word along (PassTail::error_flag): no copy needed
  uint32_t max_sub_split = 32;        // finest cut of a tile for small all-exact launches (a divisor of the 32 chunks of a tile)
  uint32_t sub_split_wgs = 2048;      // ... cut finer only while the launch has fewer workgroups than this
  bool want_selected = false;          // this call wants the per-point winner mask written (mi355_icp_reference_pass)
  bool error_flag_clean = true;        // the device word is known to be zero (the last call read it as zero): no fill launch at the next call's start
  uint32_t error_word = 0;
  bool pass_tail = true;               // MI355_ICP_PASS_TAIL=0: final-reduce launch + device-to-host copy instead (A/B)
  size_t keys_clean = 0;               // leading slots of `keys` known to hold ~0 (merged key rows of reference mode are
                                       // handed back cleared by the resolve; anything else that writes keys zeroes this)

  // scene preparation (packed AoS xyz, ping-pong)
  float* scene[2] = {nullptr, nullptr}; size_t scene_cap = 0;
  uint32_t scene_n = 0; int scene_cur = 0; bool scene_loaded = false;
  float* upload_raw = nullptr; size_t upload_raw_cap = 0;        // staging for the caller's rows (one cloud at a time)
  uint32_t* upload_flags = nullptr; size_t upload_flags_cap = 0; // per cloud: a non-finite value was met
  float* scene_planes = nullptr; size_t scene_planes_cap = 0;    // SoA copy of the scene for the k-NN scan
  XCopy scene_xt; size_t scene_xt_stride = 0;                    // its Morton-sorted copy with per-tile spheres (pruned k-NN)
  bool knn_brute = false;                                        // MI355_ICP_KNN_MODE=brute: the exhaustive scan (A/B, tests)
  float* scene_normal_planes = nullptr; size_t scene_normal_planes_cap = 0;
  float* scene_normals_aos = nullptr; size_t scene_normals_aos_cap = 0;
  float* knn_d = nullptr; size_t knn_d_cap = 0;
  int32_t* knn_i = nullptr; size_t knn_i_cap = 0;
  int32_t* knn_out = nullptr; size_t knn_out_cap = 0;
  bool scene_normals_valid = false; uint32_t scene_normals_padded = 0;
  unsigned long long* scene_state = nullptr; size_t scene_state_cap = 0;
  float* plane_hyp = nullptr; uint32_t* plane_counts = nullptr; double* plane_partials = nullptr; size_t plane_partials_cap = 0;  // RANSAC plane fit
  // ... its read-backs land in pinned memory: hypotheses and counters (ONE device block, one copy), then the moment rows
  unsigned char* plane_host = nullptr; size_t plane_host_cap = 0;
  uint32_t* scene_counters = nullptr;       // device: ticket, kept total, error bits
  uint32_t* scene_counters_host = nullptr;  // pinned

  // upstream's pre-normalisation (MI355_ICP_FLAG_UPSTREAM): chunk sums of a cloud's coordinates at upload, and per call the
  // candidates' distance sums (upstream_kernels.hip), read back into pinned memory behind an event
  double* cloud_sum_dev = nullptr; size_t cloud_sum_cap = 0;
  double* norm_dev = nullptr; double* norm_host = nullptr; size_t norm_cap = 0;  // [job][model | scene][kCloudSumBlocks]
  hipEvent_t norm_event = nullptr;
  double scene_sum[3] = {0, 0, 0}; bool scene_sum_valid = false;  // coordinate sums of the scene mi355_icp_scene_normals laid out

  CommState comm;  // slab-sharded runs: the cross-rank exchange (comm.hip); not attached = single GPU

  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  mi355_icp_profile prof{};
  bool profile_always = false;

  int scan_ppl = 0;                  // scan points-per-lane override (2, 4, 8); 0 = planner decides per call
  bool scan_sgpr = false;            // experimental scalar-load scan (MI355_ICP_SCAN_MODE=sgpr)
  uint32_t* xt_scratch = nullptr; size_t xt_scratch_cap = 0;  // Morton codes / indices while building the sorted target copy
  float* xt_tile_box = nullptr; size_t xt_tile_box_cap = 0;   // ... and the tiles' bounding boxes between its two kernels (8 floats per tile)
  uint32_t* xt_meta_host = nullptr; size_t xt_meta_cap = 0; hipEvent_t xt_meta_event = nullptr; bool xt_meta_pending = false;  // pinned (offset, count) table of ensure_xcopy
  uint32_t* xt_box = nullptr;     size_t xt_box_cap = 0;  // per cloud: bounding box as order-preserving keys
  unsigned char* xt_temp = nullptr; size_t xt_temp_cap = 0;   // rocPRIM sort scratch
  uint32_t mfma_share_pct = 65;      // hybrid: per cent of the scan workgroups that run on the matrix cores
  hipStream_t stream2 = nullptr;     // hybrid: the MFMA workgroups are launched here, beside the VALU ones
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int scan_fused = -1;               // -1 auto (by pairs per launch: >= 2.5e8 the split-bf16 expansion filter), 0 exact, 1 fused, 2 xfilter,
                                     // 3 xmfma, 4 xhybrid, 5 xbf16 (MI355_ICP_SCAN_MODE)
  // fused-filter path
  unsigned long long* keys3 = nullptr; size_t keys3_cap = 0;
  uint32_t* overflow_list = nullptr;   size_t overflow_list_cap = 0;
  uint32_t* overflow_count = nullptr;  uint32_t* overflow_count_host = nullptr; size_t overflow_jobs_cap = 0;
  float* fb_planes = nullptr;          size_t fb_planes_cap = 0;   // gathered overflow points (SoA)
  int32_t* fb_index = nullptr;         size_t fb_index_cap = 0;
  float* fb_dist2 = nullptr;           size_t fb_dist2_cap = 0;
  JobDesc* jobs2_dev = nullptr;        JobDesc* jobs2_host = nullptr; size_t jobs2_cap = 0;
  uint64_t overflow_points_total = 0;  // statistics
  uint32_t target_blocks = 12288;    // planner: aim for about this many scan workgroups per launch
  bool target_blocks_mfma_forced = false;
  uint32_t target_blocks_mfma = 4096; // ... for the matrix-core filter: its workgroups span 512 points and publish two key
                                     // rows per split, so fewer, longer ones cost less key traffic (100k<->100k: resolve
                                     // 82 -> 35 us at equal scan time; profiles/r01o_target_blocks.txt)
  bool call_pruned = false;          // this call asked for the pruned scan (MI355_ICP_FLAG_PRUNED_SCAN)
  bool call_auto_prune = false;      // this call (mi355_icp_align_reference without a scan flag) lets the planner take the exact pruned
                                     // scan for the passes where it is the faster way to the same bits (service_prune_pays)
  int service_prune = -1;            // MI355_ICP_SERVICE_PRUNE: -1 auto (default), 0 never, 1 every pass of such a call
  int xbf16_shape = 0;               // 0: <4 sets, 256 threads> workgroups, 1: <2 sets, 512 threads> (MI355_ICP_XBF16_SHAPE)
  uint32_t min_tiles_per_split = 2;  // planner: never cut the target finer than this many tiles per workgroup
  int resolve_wave_points = 0;       // MI355_ICP_RESOLVE_WAVE_POINTS=16|32|64: points per wave of the filter's resolve (0: by launch size)
  double filter_pairs = 2.5e8;        // launches from this many pairs on take the filter scan (MI355_ICP_FILTER_PAIRS: A/B)
  bool class_split = true;           // MI355_ICP_CLASS_SPLIT=0: one-round matrix-core launches with equal runs of tiles (A/B)
  bool plan_debug = false;           // MI355_ICP_PLAN_DEBUG=1: one line per plan on stderr
  double split_model_pairs = 5e10;    // ... below this many pairs per launch (MI355_ICP_SPLIT_MODEL_PAIRS: A/B)
  bool split_model = true;           // small matrix-core launches: pick the split that fills whole rounds of resident workgroups (MI355_ICP_SPLIT_MODEL=0: A/B)
  uint32_t scan_slots = 512;         // matrix-core scan workgroups resident at once: 2 per CU (256 VGPRs x 4 waves each)
  bool scan_queue = false;           // MI355_ICP_SCAN_QUEUE=1: matrix-core launches below queue_pairs draw their tiles from tickets (kXbLayoutQueue).
                                     // OFF by default: measured SLOWER than the static cuts on every small shape (round 4: 100k<->100k 0.35 -> 0.59 ms per
                                     // iteration, the request 1.86 -> 2.2 ms; profiles/r04g_queue_ab.txt, r04g_xb_timeline.txt) - a group of 4 tiles in the
                                     // hands of the younger wave of a SIMD takes ~50 us, which is the launch's tail, and every draw stalls wave 0 on the
                                     // ticket's round trip.  Kept as a tested layout (bit-identical results) for the next attempt: docs/LAB_NOTES.md 1.6
  double queue_pairs = 5e10;          // ... launches below this many pairs draw their tiles from per-source-block tickets (MI355_ICP_QUEUE_PAIRS)
  uint32_t queue_members = 0;        // MI355_ICP_QUEUE_MEMBERS: workgroups per source block in such a launch (0: by the launch's size)
  uint32_t* xb_queue = nullptr; size_t xb_queue_cap = 0;  // [job][source block] x (ticket, members done): zero between launches
  bool level_sort = false;           // MI355_ICP_LEVEL_SORT=1: every pyramid level's sorted copy from a radix sort of its own (A/B)
  bool sort_sources = true;          // matrix-core scan: walk the sources in Morton order (MI355_ICP_SORT_SOURCES=0 turns it off: A/B)
};

namespace {

int fail(mi355_icp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                          \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail((ctx), MI355_ICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

template <typename T>
int grow(mi355_icp_ctx* c, T*& ptr, size_t& cap, size_t need);

// The one synchronisation of a pass that does not end itself (PassTail): a one-thread launch behind everything on the stream
// stores a sequence number to a word of pinned memory, and the host polls THAT for a bounded while (a short pass ends within it)
// before it blocks in hipStreamSynchronize (long scans: no core is burnt waiting for them; a faulted launch: the error comes out
// there).  Rounds 2-3 polled with hipStreamQuery instead.  On this runtime a query hands the stream's completion to the
// runtime's signal-handler thread, and a hipStreamSynchronize behind it then waits for that thread: 24 ms after a 2.6 ms pass
// on the boxes of round 4 (profiles/r04w_sync_pass.txt) - every mi355_icp_correspond of a large cloud and every pass of the
// rank-per-process loop paid it.  A synchronise that no query preceded polls the signal itself and returns within microseconds.
int sync_pass(mi355_icp_ctx* c) {
  c->sync_seq = c->sync_seq == 0xFFFFFFFFu ? 1u : c->sync_seq + 1u;  // never the word's initial 0
  uint32_t* const word = c->ready_host + kReadyWords;                  // the word behind the PassTail's
  launch_publish_word(c->stream, c->ready_mapped + kReadyWords, c->sync_seq);
  HIP_TRY(c, hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 1;; ++spin) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == c->sync_seq) return MI355_ICP_OK;
    if ((spin & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != c->sync_seq) return fail(c, MI355_ICP_ERR_INTERNAL, "a pass ended without publishing its sequence number");
  return MI355_ICP_OK;
}

// Device memory -> the caller's buffer, through the context's own pinned block.  A blocking hipMemcpy into pageable memory
// makes the runtime pin the caller's pages for the copy: into a buffer nobody has touched yet (a fresh std::vector, numpy.empty)
// that cost 14 ms per 2.8 MB copy on the boxes of round 4 - mi355_icp_correspond of 700k points took 30 ms for 2.7 ms of kernels
// (profiles/r04x_result_copies.txt).  Here the device writes into memory that is pinned once, and the host copies from there
// while the next piece is in flight.
constexpr size_t kStageHalf = size_t(4) << 20;
int copy_to_user(mi355_icp_ctx* c, void* dst, const void* src_dev, size_t bytes) {
  if (!bytes) return MI355_ICP_OK;
  if (!c->stage_host) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->stage_host), 2 * kStageHalf, hipHostMallocDefault));
  const unsigned char* src = static_cast<const unsigned char*>(src_dev);
  unsigned char* out = static_cast<unsigned char*>(dst);
  const size_t pieces = (bytes + kStageHalf - 1) / kStageHalf;
  auto piece_bytes = [&](size_t k) { return std::min(kStageHalf, bytes - k * kStageHalf); };
  HIP_TRY(c, hipMemcpyAsync(c->stage_host, src, piece_bytes(0), hipMemcpyDeviceToHost, c->stream));
  for (size_t k = 0; k < pieces; ++k) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // piece k has landed in half k & 1
    if (k + 1 < pieces)
      HIP_TRY(c, hipMemcpyAsync(c->stage_host + ((k + 1) & 1) * kStageHalf, src + (k + 1) * kStageHalf, piece_bytes(k + 1), hipMemcpyDeviceToHost, c->stream));
    std::memcpy(out + k * kStageHalf, c->stage_host + (k & 1) * kStageHalf, piece_bytes(k));
  }
  return MI355_ICP_OK;
}

// The same for a pass whose last workgroup publishes `seq` in pinned memory behind its results (PassTail): no copy command
// to wait for, the word is there as soon as the rows are.
uint32_t overflow_cap(const mi355_icp_ctx* c, int n_active);
uint32_t pass_overflow_cap(const mi355_icp_ctx* c, const Plan& plan, int n_active, const float* target_planes);

// The PassTail of the next pass, if its accumulation launch may end it (single GPU; every job has a workgroup to be its
// last one; no more jobs than publish words); done / sums: the loop's own counter array and mapped result block.
PassTail make_pass_tail(mi355_icp_ctx* c, const Plan& plan, int n_active, uint32_t sbx, bool allowed, bool defer, uint32_t* done, double* sums, const float* target_planes) {
  bool on = allowed && c->pass_tail && sbx > 0 && static_cast<size_t>(n_active) <= kReadyWords;
  for (int k = 0; k < n_active; ++k) on = on && c->jobs_host[k].src_blocks > 0;
  PassTail tail;
  if (on) {
    tail.done = done;
    tail.sums = sums;
    tail.overflow_count = plan.fused ? c->overflow_count : nullptr;
    tail.overflow_cap = defer ? pass_overflow_cap(c, plan, n_active, target_planes) : 0xFFFFFFFFu;
    c->pass_seq = (c->pass_seq + 1) & 0x7FFFFFFFu;
    tail.seq = c->pass_seq;
    tail.ready = c->ready_mapped;
    tail.error_flag = c->error_flag;
  }
  c->error_word_known = false;
  return tail;
}

// Every job publishes (seq << 1 | error bit) in its own word.
int wait_pass(mi355_icp_ctx* c, uint32_t seq, int n_jobs, uint32_t* error_bit) {
  const auto t0 = std::chrono::steady_clock::now();
  auto all_there = [&]() {
    uint32_t err = 0;
    for (int k = 0; k < n_jobs; ++k) {
      const uint32_t w = __atomic_load_n(c->ready_host + k, __ATOMIC_ACQUIRE);
      if ((w >> 1) != (seq & 0x7FFFFFFFu)) return false;
      err |= w & 1u;
    }
    *error_bit = err;
    return true;
  };
  for (uint32_t spin = 1;; ++spin) {
    if (all_there()) return MI355_ICP_OK;
    if ((spin & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // long scans: sleep; a faulted launch: the error comes out here
  if (!all_there()) return fail(c, MI355_ICP_ERR_INTERNAL, "a reference-mode pass ended without publishing its result");
  return MI355_ICP_OK;
}

// Hands every device buffer of a cloud set (planes, normals, pyramid levels, sorted copies) back to the pool.
void release_cloud_buffers(mi355_icp_ctx* c, CloudSet& set) {
  c->pool.give(set.planes);
  c->pool.give(set.normals);
  c->pool.give(set.rows);
  set.planes = set.normals = nullptr;
  set.rows = nullptr;
  for (auto& lv : set.levels) {
    c->pool.give(lv.planes);
    c->pool.give(lv.rows);
    free_xcopy(c->pool, lv.xt);
  }
  set.levels.clear();
  set.shard_begin.clear();
  set.shard_total.clear();
  set.sum_xyz.clear();
  free_xcopy(c->pool, set.xt);
}

// Coordinate sums of clouds laid out in `view` (upstream's computeMeanCols, times the row count): every cloud's kCloudSumBlocks
// chunk sums are enqueued into `host` (3 * kCloudSumBlocks doubles per cloud); the caller synchronises the stream - every
// upload path does anyway - and then adds the chunks up in chunk order with finish_cloud_sums.
int enqueue_cloud_sums(mi355_icp_ctx* c, DevicePlanes view, const std::vector<CloudEntry>& clouds, std::vector<double>& host) {
  const size_t per = 3 * static_cast<size_t>(kCloudSumBlocks);
  host.assign(per * clouds.size(), 0.0);
  if (clouds.empty()) return MI355_ICP_OK;
  if (const int st = grow(c, c->cloud_sum_dev, c->cloud_sum_cap, per * clouds.size()); st != MI355_ICP_OK) return st;
  for (size_t k = 0; k < clouds.size(); ++k) launch_cloud_sums(c->stream, view, clouds[k].off, clouds[k].n, c->cloud_sum_dev + per * k);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(host.data(), c->cloud_sum_dev, host.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return MI355_ICP_OK;
}
void finish_cloud_sums(const std::vector<double>& host, size_t n_clouds, std::vector<double>& sum_xyz) {
  sum_xyz.assign(3 * n_clouds, 0.0);
  for (size_t k = 0; k < n_clouds; ++k)
    for (int b = 0; b < kCloudSumBlocks; ++b)
      for (int a = 0; a < 3; ++a) sum_xyz[3 * k + a] += host[(k * kCloudSumBlocks + static_cast<size_t>(b)) * 3 + a];
}

int upload_clouds(mi355_icp_ctx* c, CloudSet& set, int n_clouds, const float* const* xyz, const size_t* n_points,
                  size_t stride_bytes, uint32_t pad_multiple, float pad_value, bool allow_empty, const char* what) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (n_clouds < 1 || !xyz || !n_points) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, std::string(what) + ": no clouds");
  if (stride_bytes < 12 || stride_bytes % 4 != 0)
    return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, std::string(what) + ": stride_bytes must be a multiple of 4 and >= 12");
  std::vector<CloudEntry> entries(static_cast<size_t>(n_clouds));
  uint64_t total = 0;
  for (int k = 0; k < n_clouds; ++k) {
    const size_t n = n_points[k];
    if (n == 0 && !allow_empty) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, std::string(what) + ": empty cloud");
    if (n && !xyz[k]) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, std::string(what) + ": null cloud pointer");
    const uint64_t padded = (static_cast<uint64_t>(n) + pad_multiple - 1) / pad_multiple * pad_multiple;
    if (total + padded > 0xFFFF0000ull) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, std::string(what) + ": too many points");
    entries[k] = CloudEntry{static_cast<uint32_t>(total), static_cast<uint32_t>(n), static_cast<uint32_t>(padded)};
    total += padded;
  }
  const size_t stride = static_cast<size_t>(std::max<uint64_t>(total, pad_multiple));
  const uint32_t fstride = static_cast<uint32_t>(stride_bytes / 4);
  const bool with_normals = fstride >= 6;  // rows of xyz + normal (N x 6), opencv_surface_match.cpp:22,57-59
  // The caller's rows go to the device as they are and are re-laid out to SoA planes, padded and checked for NaN/Inf
  // THERE (rows_to_planes_kernel): 12 B per point over PCIe and one pass at HBM speed, instead of a scalar host loop
  // over every point plus a second copy.  The new planes replace the old ones only if every cloud was finite.
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  size_t raw_floats = 1;
  for (int k = 0; k < n_clouds; ++k) raw_floats = std::max(raw_floats, static_cast<size_t>(entries[k].n) * fstride);
  int st;
  if ((st = grow(c, c->upload_raw, c->upload_raw_cap, raw_floats)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->upload_flags, c->upload_flags_cap, static_cast<size_t>(n_clouds))) != MI355_ICP_OK) return st;
  float *planes = nullptr, *normals = nullptr;
  HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&planes), 3 * stride * sizeof(float)));
  if (with_normals && c->pool.take(reinterpret_cast<void**>(&normals), 3 * stride * sizeof(float)) != hipSuccess) {
    c->pool.give(planes);
    return fail(c, MI355_ICP_ERR_HIP, "hipMalloc (normal planes) failed");
  }
  auto drop_new = [&]() { c->pool.give(planes); c->pool.give(normals); };
  bool ok = hipMemsetAsync(c->upload_flags, 0, static_cast<size_t>(n_clouds) * sizeof(uint32_t), c->stream) == hipSuccess;
  if (total == 0) {  // only empty clouds: planes of padding
    ok = ok && hipMemsetAsync(planes, 0, 3 * stride * sizeof(float), c->stream) == hipSuccess;
    if (normals) ok = ok && hipMemsetAsync(normals, 0, 3 * stride * sizeof(float), c->stream) == hipSuccess;
  }
  for (int k = 0; k < n_clouds && ok; ++k) {
    if (entries[k].n)  // pageable source: returns once the rows have left the caller's buffer, ordered on the stream
      ok = hipMemcpyAsync(c->upload_raw, xyz[k], static_cast<size_t>(entries[k].n) * stride_bytes, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    launch_rows_to_planes(c->stream, c->upload_raw, fstride, entries[k].n, planes, normals, stride, entries[k].off, entries[k].padded,
                          pad_value, c->upload_flags, static_cast<uint32_t>(k));
  }
  std::vector<uint32_t> flags(static_cast<size_t>(n_clouds), 0u);
  std::vector<double> sum_chunks;
  ok = ok && hipGetLastError() == hipSuccess &&
       enqueue_cloud_sums(c, DevicePlanes{planes, planes + stride, planes + 2 * stride}, entries, sum_chunks) == MI355_ICP_OK &&
       hipMemcpyAsync(flags.data(), c->upload_flags, flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
       hipStreamSynchronize(c->stream) == hipSuccess;
  if (!ok) {
    drop_new();
    return fail(c, MI355_ICP_ERR_HIP, std::string(what) + ": upload failed: " + hipGetErrorString(hipGetLastError()));
  }
  for (int k = 0; k < n_clouds; ++k)
    if (flags[k]) {
      drop_new();
      return fail(c, MI355_ICP_ERR_NONFINITE_INPUT,
                  std::string(what) + ": cloud " + std::to_string(k) + " holds NaN/Inf (strip them first, as the reference does at pose_estimation.cpp:248)");
    }
  release_cloud_buffers(c, set);
  set.planes = planes;
  set.normals = normals;
  set.stride = stride;
  set.clouds = std::move(entries);
  finish_cloud_sums(sum_chunks, set.clouds.size(), set.sum_xyz);
  return MI355_ICP_OK;
}

template <typename T>
int grow(mi355_icp_ctx* c, T*& ptr, size_t& cap, size_t need) {
  if (need <= cap) return MI355_ICP_OK;
  if (ptr) { HIP_TRY(c, hipFree(ptr)); ptr = nullptr; cap = 0; }
  const size_t want = need + need / 4;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&ptr), want * sizeof(T)));
  cap = want;
  return MI355_ICP_OK;
}

int ensure_buffers(mi355_icp_ctx* c, const Plan& plan, size_t n_jobs, bool* overflow_cleared = nullptr) {
  int st;
  {
    const size_t before = c->keys_cap;
    if (!plan.fused && (st = grow(c, c->keys, c->keys_cap, static_cast<size_t>(plan.max_split) * plan.out_total)) != MI355_ICP_OK) return st;
    if (c->keys_cap != before) c->keys_clean = 0;  // a new buffer holds anything
  }
  if (plan.out_total > c->out_cap) {
    if (c->out_index) { HIP_TRY(c, hipFree(c->out_index)); c->out_index = nullptr; }
    if (c->out_dist2) { HIP_TRY(c, hipFree(c->out_dist2)); c->out_dist2 = nullptr; }
    c->out_cap = 0;
    const size_t want = plan.out_total + plan.out_total / 4;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->out_index), want * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->out_dist2), want * sizeof(float)));
    c->out_cap = want;
  }
  if ((st = grow(c, c->partials, c->partials_cap, static_cast<size_t>(std::max<uint32_t>(plan.partial_total, 1)) * MI355_ICP_NSUMS)) != MI355_ICP_OK) return st;
  if (plan.fused) {
    if ((st = grow(c, c->keys3, c->keys3_cap, plan.key_ranks() * static_cast<size_t>(plan.key_rows()) * plan.out_total)) != MI355_ICP_OK) return st;
    if ((st = grow(c, c->overflow_list, c->overflow_list_cap, std::max<size_t>(plan.out_total, 1))) != MI355_ICP_OK) return st;
    if (n_jobs > c->overflow_jobs_cap) {
      if (c->overflow_count) { HIP_TRY(c, hipFree(c->overflow_count)); c->overflow_count = nullptr; }
      if (c->overflow_count_host) { HIP_TRY(c, hipHostFree(c->overflow_count_host)); c->overflow_count_host = nullptr; }
      const size_t want = std::max<size_t>(n_jobs, 8) * 2;
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->overflow_count), want * sizeof(uint32_t)));
      if (overflow_cleared) *overflow_cleared = false;  // a new buffer holds anything
      HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->overflow_count_host), want * sizeof(uint32_t), hipHostMallocDefault));
      c->overflow_jobs_cap = want;
    }
  }
  if (n_jobs > c->jobs_cap) {
    if (c->sums_dev) { HIP_TRY(c, hipFree(c->sums_dev)); c->sums_dev = nullptr; }
    if (c->jobs_dev) { HIP_TRY(c, hipFree(c->jobs_dev)); c->jobs_dev = nullptr; }
    if (c->jobs_host) { HIP_TRY(c, hipHostFree(c->jobs_host)); c->jobs_host = nullptr; }
    if (c->sums_host) { HIP_TRY(c, hipHostFree(c->sums_host)); c->sums_host = nullptr; }
    c->jobs_cap = 0;
    const size_t want = std::max<size_t>(n_jobs, 8) * 2;
    // + two doubles per job behind the rows: the overflow-excess flag (exchanged with the sums in a slab-sharded pass) and
    // the overflow count (final_reduce_kernel)
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->sums_dev), want * (MI355_ICP_NSUMS + 2) * sizeof(double)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->jobs_dev), want * sizeof(JobDesc)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->jobs_host), want * sizeof(JobDesc), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->sums_host), want * (MI355_ICP_NSUMS + 2) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->sums_mapped), c->sums_host, 0));
    if (c->p2p_done) { HIP_TRY(c, hipFree(c->p2p_done)); c->p2p_done = nullptr; }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->p2p_done), want * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->p2p_done, 0, want * sizeof(uint32_t), c->stream));
    c->jobs_cap = want;
  }
  // the per-job overflow counters are zero between passes (the final-reduce kernels reset them); make that true at the
  // start of every call too, whatever an earlier call left behind when it failed half way
  // (reference mode calls this once per pyramid level: one clear per CALL is enough - `overflow_cleared` remembers)
  if (plan.fused && !(overflow_cleared && *overflow_cleared)) {
    HIP_TRY(c, hipMemsetAsync(c->overflow_count, 0, c->overflow_jobs_cap * sizeof(uint32_t), c->stream));
    if (overflow_cleared) *overflow_cleared = true;
  }
  return MI355_ICP_OK;
}

// XCD-aware split count: workgroups b and b+8 share an XCD (and its L2); with n_split in {1,2,4} or a multiple of 8
// they also share the run of target tiles they stream.
uint32_t xcd_round(uint32_t split) {
  if (split >= 8) return split - split % 8;
  if (split >= 4) return 4;
  if (split >= 2) return 2;
  return 1;
}

// The service path's choice between the brute-force scan and the exact PRUNED scan for one pass of a request (reference mode
// only: the path the node calls; mi355_icp_align - what bench.py measures - is brute force unless asked otherwise).  Both give
// the same indices and distances bit for bit; this only estimates which is faster.  Fitted to profiles/r04e_pruned_vs_brute_sweep.jsonl
// (6 candidates of an n-point model against an m-point scene, one pass; scan + resolve + reduce, milliseconds):
//   brute force  ~ pairs / 3.4e10 + 0.03 + 2e-7 x source points       (3.4e13 pairs/s at these launch sizes)
//   pruned       ~ 0.30 + 2e-6 n + m (2e-6 + 0.02 / n)                 (a dense scene under a sparse model prunes badly)
// the pruned scan is taken when it wins by more than 10 % - below that the brute-force pass, whose time does not depend on
// how far the candidate still is from the scene, is the safer one.
bool service_prune_pays(double jobs, double n_model, double n_scene) {
  if (n_model < 1.0 || n_scene < 1.0) return false;
  const double width = jobs > 6.0 ? jobs / 6.0 : 1.0;  // measured with 6 candidates; more of them are more workgroups
  const double brute = jobs * n_model * n_scene / 3.4e10 + 0.03 + 2e-7 * jobs * n_model;
  const double pruned = 0.30 + (2e-6 * n_model + n_scene * (2e-6 + 0.02 / n_model)) * width;
  return pruned * 1.1 < brute;
}

// Lays the jobs out and decides how finely to cut each target: enough scan workgroups to keep
// 256 CUs x 4 resident workgroups busy for several rounds, never finer than min_tiles_per_split.
int make_plan_on(mi355_icp_ctx* c, const std::vector<CloudEntry>& src_clouds, const std::vector<CloudEntry>& tgt_clouds,
                 int n_jobs, const int* source_ids, const int* target_ids, float max_corr, Plan& plan) {
  if (tgt_clouds.empty()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "no target clouds uploaded (mi355_icp_set_targets)");
  if (src_clouds.empty()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "no source clouds uploaded (mi355_icp_set_sources)");
  plan = Plan{};
  plan.jobs.resize(static_cast<size_t>(n_jobs));
  double total_pairs = 0.0;
  for (int k = 0; k < n_jobs; ++k) {
    const int s = source_ids ? source_ids[k] : 0, t = target_ids ? target_ids[k] : 0;
    if (s < 0 || static_cast<size_t>(s) >= src_clouds.size()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "source id out of range");
    if (t < 0 || static_cast<size_t>(t) >= tgt_clouds.size()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "target id out of range");
    total_pairs += static_cast<double>(src_clouds[s].n) * static_cast<double>(tgt_clouds[t].n);
  }
  // Register blocking of the scan: 8 points per lane is the fastest steady state (fewest LDS reads per pair) but
  // makes the coarsest workgroups; below ~4e11 pairs per launch the finer 2-point workgroups balance the 256 CUs
  // better and win (profiles/r01b_scan_ab_sizes.txt: 1M<->1M 135 vs 141 ms; 64x50k<->50k 24.3 vs 21.5 ms).
  plan.scan_ppl = c->scan_ppl ? c->scan_ppl : (total_pairs >= 4e11 ? 8 : 2);
  // A filter scan in front of the exact test (exact results by construction: icp_fused_kernels.hip) wins wherever the
  // scan dominates the pass; tiny launches keep the all-exact scan, which needs no sorted copy of the targets and no
  // overflow-count round trip to the host.  Threshold from profiles/r01o_planner_threshold.txt: with the matrix-core
  // filter the pass is already 23 % shorter at 16k<->16k (2.7e8 pairs), 42 % at 32k<->32k.
  const double kFilterPairs = c->filter_pairs;
  plan.fused = c->scan_fused < 0 ? total_pairs >= kFilterPairs : c->scan_fused >= 1;
  // The expansion filters need a sorted, tile-centred copy of the targets, built on the device once per upload
  // (~0.2 ms per 100k points): worth it as soon as the filter itself is.
  plan.xfilter = c->scan_fused < 0 ? total_pairs >= kFilterPairs : c->scan_fused >= 2;
  // ... and the split-bf16 matrix-core form of it (xkind 3: one issue slot per pair, 2.3x the VALU filter end to end at
  // 100k<->100k, 1M<->1M and 64 x 50k<->50k: profiles/r01n_baseline_configs.jsonl) is the default wherever it applies
  plan.xkind = c->scan_fused < 0 ? 3 : (c->scan_fused >= 2 ? c->scan_fused - 2 : 0);
  bool auto_pruned = false;
  if (c->call_auto_prune && c->scan_fused < 0 && !c->call_pruned) {  // a request's pass: whichever exact scan is faster
    double n_model = 0.0, n_scene = 0.0;
    uint32_t max_tiles = 0;
    for (int k = 0; k < n_jobs; ++k) {
      n_model += src_clouds[source_ids ? source_ids[k] : 0].n;
      n_scene += tgt_clouds[target_ids ? target_ids[k] : 0].n;
      max_tiles = std::max(max_tiles, tgt_clouds[target_ids ? target_ids[k] : 0].padded / kTile);
    }
    auto_pruned = max_tiles <= kPrunedMaxTiles &&
                  (c->service_prune == 1 || (c->service_prune < 0 && service_prune_pays(n_jobs, n_model / n_jobs, n_scene / n_jobs)));
  }
  if (c->scan_fused == 6 || c->call_pruned || auto_pruned) {  // the optional pruned scan: no keys, no filters, no splits
    plan.pruned = true;
    plan.fused = plan.xfilter = false;
    plan.xkind = 0;
  }
  if (plan.xfilter) plan.scan_ppl = c->scan_ppl ? c->scan_ppl : 4;
  if (plan.xfilter && plan.xkind && plan.scan_ppl != 2) plan.scan_ppl = 4;  // the MFMA kernel spans 128 * {4, 8} points
  if (plan.xfilter && plan.xkind == 3) plan.scan_ppl = 2;                     // split-bf16 kernel: 512 points per workgroup
  // The matrix-core scan keeps, per lane, the three best granules of its point and takes a wave-uniform branch whenever
  // ANY of the wave's 32 points improves one.  In upload order those 32 points are anywhere on the model, each improving
  // at its own places of the (Morton-ordered) target walk: the branch is taken in most chunks.  Walking the sources in
  // Morton order as well puts neighbours into one wave - they improve in the same few chunks: 31.6 -> 28.8 ms at 1M<->1M
  // (profiles/r02c_sorted_source_what_if.txt).  Same pairs, same arithmetic, same results; only which lane holds which
  // point changes, and the resolve writes each answer back at the point's original index.
  plan.sorted_src = plan.xfilter && plan.xkind == 3 && c->sort_sources;
  const uint32_t scan_src = static_cast<uint32_t>(kBlockThreads * plan.scan_ppl);
  uint64_t total_src_blocks = 0;
  for (int k = 0; k < n_jobs; ++k)
    total_src_blocks += (src_clouds[source_ids ? source_ids[k] : 0].n + scan_src - 1) / scan_src;
  float max_d2 = std::numeric_limits<float>::infinity();
  if (max_corr > 0.0f) max_d2 = max_corr * max_corr;  // float32 product, as the oracle
  // matrix-core scan: launches below ~5e10 pairs (a request's candidates against one scene) do better with a quarter of the
  // workgroups, each walking more tiles - fewer key rows for the resolve to merge, less prologue per pair (6 x 20k<->100k,
  // reference mode: 2.39 -> 2.31 ms per request; 1M<->1M is 0.6 % FASTER with the full count: profiles/r03c_target_blocks.txt)
  const uint64_t mfma_goal = c->target_blocks_mfma_forced || total_pairs >= 5e10 ? c->target_blocks_mfma : std::max<uint32_t>(c->target_blocks_mfma / 4, 1);
  const uint64_t blocks_goal = plan.xfilter && plan.xkind == 3 ? mfma_goal : c->target_blocks;
  uint32_t want_split = static_cast<uint32_t>(
      std::max<uint64_t>(1, (blocks_goal + std::max<uint64_t>(total_src_blocks, 1) - 1) / std::max<uint64_t>(total_src_blocks, 1)));
  // Small matrix-core launches run for one to four rounds of resident workgroups (2 per CU), so the LAST round decides:
  // 100k<->100k cut in 4 is 784 workgroups = 1.53 rounds of 512, the second round half empty.  Pick the cut that
  // minimises rounds x (tiles per workgroup + a workgroup's fixed cost of about one tile) instead; any cut is valid
  // (balanced runs of tiles), and at these sizes the whole operand image sits in every XCD's L2 whichever runs its
  // workgroups walk, so the XCD rounding below has nothing to protect.
  bool model_split = false;
  if (plan.xfilter && plan.xkind == 3 && c->split_model && !c->target_blocks_mfma_forced && total_pairs < c->split_model_pairs && total_src_blocks) {
    uint32_t tiles = 1;
    for (int k = 0; k < n_jobs; ++k) tiles = std::max(tiles, tgt_clouds[target_ids ? target_ids[k] : 0].padded / kTile);
    const uint32_t cap = std::max<uint32_t>(1, tiles / std::max<uint32_t>(1, c->min_tiles_per_split));
    double best_cost = std::numeric_limits<double>::infinity();
    for (uint32_t s = 1; s <= std::min<uint32_t>(cap, 64); ++s) {
      const uint64_t rounds = (total_src_blocks * s + c->scan_slots - 1) / c->scan_slots;
      // per-workgroup time in tiles (+1: prologue and key rows); the last round's stragglers (workgroups differ by how often
      // their points improve); the resolve merges one more key row per cut.  Coefficients fitted to three shapes on one box
      // (profiles/r03q_split_model_ab.txt): 100k<->100k, 6 x 20k<->100k, 6 x 20k<->285k
      const double w = (tiles + s - 1) / s + 1.0;
      const double cost = static_cast<double>(rounds) * w + 0.1 * w + 0.5 * s;
      if (cost < best_cost) { best_cost = cost; want_split = s; }
    }
    model_split = true;
  }
  // Launches of one to four rounds of resident workgroups: instead of static runs, the workgroups of a source block form a team
  // that draws groups of tiles from the block's ticket (nn_scan_xbf16_kernel, kXbLayoutQueue) - about two workgroups per slot over
  // the whole launch, never more members than a target has tile groups.
  bool queue_split = false;
  if (plan.xfilter && plan.xkind == 3 && c->scan_queue && c->xbf16_shape != 1 && !c->target_blocks_mfma_forced && total_pairs < c->queue_pairs && total_src_blocks) {
    uint32_t groups = 0xFFFFFFFFu;
    for (int k = 0; k < n_jobs; ++k) groups = std::min(groups, (tgt_clouds[target_ids ? target_ids[k] : 0].padded / kTile + kXbCentreGroup - 1) / kXbCentreGroup);
    uint32_t members = c->queue_members ? c->queue_members : static_cast<uint32_t>((2ull * c->scan_slots + total_src_blocks - 1) / total_src_blocks);
    members = std::min<uint32_t>(std::min<uint32_t>(members, 8u), groups);
    if (members >= 2) {
      want_split = members;
      model_split = true;  // no XCD rounding of the count
      queue_split = true;
    }
  }
  uint64_t out_off = 0, partial_off = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const CloudEntry& S = src_clouds[source_ids ? source_ids[k] : 0];
    const CloudEntry& T = tgt_clouds[target_ids ? target_ids[k] : 0];
    JobDesc& J = plan.jobs[k];
    std::memset(&J, 0, sizeof(J));
    J.src_off = S.off;
    J.src_n = S.n;
    J.src_blocks = (S.n + kSrcPerBlock - 1) / kSrcPerBlock;
    J.scan_blocks = (S.n + scan_src - 1) / scan_src;
    J.tgt_off = T.off;
    J.tgt_n = T.n;
    J.tgt_tiles = T.padded / kTile;
    const uint32_t cap = std::max<uint32_t>(1, J.tgt_tiles / std::max<uint32_t>(1, c->min_tiles_per_split));
    uint32_t split = std::min(std::max<uint32_t>(want_split, 1), cap);
    // XCD-aware: workgroups b and b+8 share an XCD (and its L2); with n_split in {1,2,4} or a multiple of 8 they
    // also share the run of target tiles they stream
    if (!model_split) split = xcd_round(split);
    J.n_split = split;
    J.split_q = J.tgt_tiles / split;
    J.split_r = J.tgt_tiles % split;
    J.out_off = static_cast<uint32_t>(out_off);
    J.partial_off = static_cast<uint32_t>(partial_off);
    J.max_d2 = max_d2;
    out_off += S.padded;
    partial_off += J.src_blocks;
    if (out_off > 0xFFFF0000ull) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "batch too large");
    plan.max_split = std::max(plan.max_split, J.n_split);
    plan.max_blocks_x = std::max(plan.max_blocks_x, J.scan_blocks * J.n_split);
    plan.max_src_blocks = std::max(plan.max_src_blocks, J.src_blocks);
  }
  plan.out_total = static_cast<uint32_t>(out_off);
  plan.partial_total = static_cast<uint32_t>(partial_off);
  // Small all-exact launches (the coarse pyramid levels of reference mode, small models): even one workgroup per
  // (512 source slots, tile) leaves most of the chip idle while each of them works through 1024 x 512 pairs on one
  // CU - ~37 us whatever the cloud holds.  Cut every tile into 2, 4 ... 32 runs of chunks, one workgroup each, until the
  // launch has one or two thousand of them (6 x 20k<->100k with the reference's parameters: 11 such passes per request).
  // The merged key row is handed back cleared by the resolve, which covers ceil(n / kSrcPerBlock) * kSrcPerBlock slots per
  // job; the scan merges into ceil(n / scan_src) * scan_src of them.  Only while a scan workgroup spans no more than a
  // resolve workgroup (2 or 4 points per lane; the A/B knob MI355_ICP_SCAN_PPL=8 spans 2048) is every slot the scan
  // dirtied also cleared - otherwise padding slots would keep finite keys that a later plan could mistake for a point's.
  if (!plan.fused && !plan.pruned && !c->scan_sgpr && scan_src <= static_cast<uint32_t>(kSrcPerBlock)) {
    uint64_t wgs = 0;
    for (const JobDesc& J : plan.jobs) wgs += static_cast<uint64_t>(J.scan_blocks) * J.tgt_tiles;
    uint32_t sub = 1;
    // measured on the request shape (gpurun_out/subsplit, round 3): up to 16 runs while the launch stays under ~2048
    // workgroups; the last halving (32 runs of ONE chunk: every workgroup still pays descriptor, transforms and a tile
    // load for 32 x 512 pairs) only pays for launches that would otherwise not even cover the CUs
    while (sub < c->max_sub_split && wgs * sub < (sub >= 16 ? c->sub_split_wgs / 4 : c->sub_split_wgs)) sub *= 2;
    if (sub > 1) {
      plan.merged_keys = true;
      plan.max_split = 1;  // key rows: the workgroups of a point merge into one (atomicMin)
      plan.max_blocks_x = 0;
      for (JobDesc& J : plan.jobs) {
        J.sub_split = sub;
        J.n_split = J.tgt_tiles * sub;
        J.split_q = 1;  // unused by the kernel in this mode; kept consistent for readers of the descriptor
        J.split_r = 0;
        plan.max_blocks_x = std::max(plan.max_blocks_x, J.scan_blocks * J.n_split);
      }
    }
  }
  // One-round matrix-core launches of a uniform batch (a request's candidates): between one and two workgroups per CU.  The
  // first round's workgroups are served first by every SIMD's arbiter and walk a tile in less than half the time of the ones
  // that join them; give them 13/6 of the tiles (MI355_ICP_CLASS_SPLIT=0: equal runs, A/B).
  if (queue_split) {
    plan.queued = true;
    for (const JobDesc& J : plan.jobs) {
      plan.grid3_x = std::max(plan.grid3_x, J.scan_blocks);
      plan.grid3_z = std::max(plan.grid3_z, J.n_split);
    }
  }
  if (model_split && !queue_split && c->class_split && n_jobs <= 8) {
    const JobDesc& J0 = plan.jobs[0];
    bool uniform = J0.n_split >= 2 && J0.tgt_tiles >= 4 * J0.n_split;
    for (const JobDesc& J : plan.jobs) uniform = uniform && J.scan_blocks == J0.scan_blocks && J.n_split == J0.n_split && J.tgt_tiles == J0.tgt_tiles;
    const uint64_t wgs = static_cast<uint64_t>(J0.scan_blocks) * J0.n_split * static_cast<uint64_t>(n_jobs);
    if (uniform && wgs > c->scan_slots / 2 && wgs <= c->scan_slots) {
      plan.classed = true;
      plan.grid3_x = J0.scan_blocks;
      plan.grid3_z = J0.n_split;
      for (JobDesc& J : plan.jobs) J.first_class = c->scan_slots / 2;
    }
  }
  if (plan.pruned)  // the pruned kernel keeps a bitmap of needed tiles in LDS
    for (const JobDesc& J : plan.jobs)
      if (J.tgt_tiles > kPrunedMaxTiles) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "MI355_ICP_SCAN_MODE=pruned: a target cloud has more than 16M points");

  if (static_cast<uint64_t>(plan.key_rows()) * plan.out_total > 0xFFFF0000ull) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "batch too large");
  if (c->plan_debug)
    std::fprintf(stderr, "mi355_icp plan: %d jobs, %.3g pairs, pruned %d fused %d xkind %d ppl %d, source blocks %llu, want_split %u%s -> split %u, %u workgroups x %d, sub_split %u\n",
                 n_jobs, total_pairs, int(plan.pruned), int(plan.fused), plan.xkind, plan.scan_ppl, static_cast<unsigned long long>(total_src_blocks), want_split,
                 queue_split ? " (queue)" : model_split ? " (model)" : "", plan.jobs[0].n_split, plan.max_blocks_x, n_jobs, plan.jobs[0].sub_split);
  return MI355_ICP_OK;
}

int make_plan(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids, float max_corr, Plan& plan) {
  return make_plan_on(c, c->sources.clouds, c->targets.clouds, n_jobs, source_ids, target_ids, max_corr, plan);
}

bool pose_to_f32(const double* pose16, float* out12) {
  bool ok = true;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) {
      const float v = static_cast<float>(pose16[4 * r + col]);
      ok = ok && std::isfinite(v);
      out12[4 * r + col] = v;
    }
  return ok;
}

// Fused path, overflow handling: points whose candidate chunks did not fit the filter's band are answered by the
// all-exact scan on a gathered copy.  counts[] (host) holds the number of such points per active job.
int run_overflow_fallback(mi355_icp_ctx* c, const Plan& plan, int n_active, const uint32_t* counts, DevicePlanes S, DevicePlanes T) {
  std::vector<int> who;
  size_t padded_total = 0, blocks_total = 0, scan_blocks_total = 0;
  for (int k = 0; k < n_active; ++k)
    if (counts[k]) {
      who.push_back(k);
      padded_total += (static_cast<size_t>(counts[k]) + kSrcPad - 1) / kSrcPad * kSrcPad;
      blocks_total += (counts[k] + kSrcPerBlock - 1) / kSrcPerBlock;
      scan_blocks_total += (counts[k] + 2 * kBlockThreads - 1) / (2 * kBlockThreads);
    }
  int st;
  if ((st = grow(c, c->fb_planes, c->fb_planes_cap, 3 * padded_total)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->fb_index, c->fb_index_cap, padded_total)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->fb_dist2, c->fb_dist2_cap, padded_total)) != MI355_ICP_OK) return st;
  if (who.size() > c->jobs2_cap) {
    if (c->jobs2_dev) { HIP_TRY(c, hipFree(c->jobs2_dev)); c->jobs2_dev = nullptr; }
    if (c->jobs2_host) { HIP_TRY(c, hipHostFree(c->jobs2_host)); c->jobs2_host = nullptr; }
    const size_t want = std::max<size_t>(who.size(), 8) * 2;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->jobs2_dev), want * sizeof(JobDesc)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->jobs2_host), want * sizeof(JobDesc), hipHostMallocDefault));
    c->jobs2_cap = want;
  }
  const uint32_t want_split = static_cast<uint32_t>(std::max<size_t>(1, c->target_blocks / std::max<size_t>(scan_blocks_total, 1)));
  uint32_t max_split = 1, bx = 0, sbx = 0;
  size_t off = 0, prow = 0;
  for (size_t q = 0; q < who.size(); ++q) {
    const JobDesc& M = c->jobs_host[who[q]];
    const uint32_t n = counts[who[q]];
    const uint32_t padded = (n + kSrcPad - 1) / kSrcPad * kSrcPad;
    launch_gather_points(c->stream, S, M.src_off, c->overflow_list + M.out_off, n, c->fb_planes + off, padded_total, padded);
    JobDesc J = M;
    J.sub_split = 0;
    J.src_off = static_cast<uint32_t>(off);
    J.src_n = n;
    J.src_blocks = (n + kSrcPerBlock - 1) / kSrcPerBlock;
    J.scan_blocks = (n + 2 * kBlockThreads - 1) / (2 * kBlockThreads);
    const uint32_t cap = std::max<uint32_t>(1, J.tgt_tiles / std::max<uint32_t>(1, c->min_tiles_per_split));
    J.n_split = xcd_round(std::min(std::max<uint32_t>(want_split, 1), cap));
    J.split_q = J.tgt_tiles / J.n_split;
    J.split_r = J.tgt_tiles % J.n_split;
    J.out_off = static_cast<uint32_t>(off);
    J.partial_off = static_cast<uint32_t>(prow);
    c->jobs2_host[q] = J;
    max_split = std::max(max_split, J.n_split);
    bx = std::max(bx, J.scan_blocks * J.n_split);
    sbx = std::max(sbx, J.src_blocks);
    off += padded;
    prow += J.src_blocks;
  }
  if ((st = grow(c, c->keys, c->keys_cap, static_cast<size_t>(max_split) * padded_total)) != MI355_ICP_OK) return st;
  c->keys_clean = 0;
  if ((st = grow(c, c->partials, c->partials_cap, std::max<size_t>(blocks_total, plan.partial_total) * MI355_ICP_NSUMS)) != MI355_ICP_OK) return st;
  HIP_TRY(c, hipMemcpyAsync(c->jobs2_dev, c->jobs2_host, who.size() * sizeof(JobDesc), hipMemcpyHostToDevice, c->stream));
  const DevicePlanes F{c->fb_planes, c->fb_planes + padded_total, c->fb_planes + 2 * padded_total};
  launch_nn_scan(c->stream, 2, c->jobs2_dev, static_cast<int>(who.size()), bx, F, T, c->keys, static_cast<uint32_t>(padded_total));
  launch_resolve(c->stream, c->jobs2_dev, static_cast<int>(who.size()), sbx, F, T, c->keys, static_cast<uint32_t>(padded_total), c->fb_index,
                 c->fb_dist2, c->partials, c->error_flag);
  for (size_t q = 0; q < who.size(); ++q) {
    const JobDesc& M = c->jobs_host[who[q]];
    const JobDesc& J = c->jobs2_host[q];
    launch_scatter_results(c->stream, c->overflow_list + M.out_off, J.src_n, c->fb_index + J.out_off, c->fb_dist2 + J.out_off,
                           c->out_index, c->out_dist2, M.out_off);
    c->overflow_points_total += J.src_n;
    c->prof.overflow_points += J.src_n;
  }
  c->prof.overflow_host_passes += 1;
  HIP_TRY(c, hipGetLastError());
  return MI355_ICP_OK;
}

// Builds, on the device, the expansion-filter copy of a group of target clouds (XCopy, xtarget_kernels.hip).
// Everything is enqueued on the engine's stream; the scan that follows on the same stream sees it complete.
// the buffers of a sorted copy of `stride` slots, padding written (a copy keeps its buffers while its stride stays the same)
int alloc_xcopy(mi355_icp_ctx* c, XCopy& x, size_t stride) {
  if (!x.planes) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.planes), 7 * stride * sizeof(float)));
  if (!x.perm) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.perm), stride * sizeof(uint32_t)));
  if (!x.table) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.table), 4 * (stride / kTile) * sizeof(float)));
  if (!x.bf) HIP_TRY(c, c->pool.take(&x.bf, (stride / kTile) * static_cast<size_t>(kXbTileVecs) * 16));
  if (!x.bf_table) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.bf_table), 4 * (stride / kTile) * sizeof(float)));
  if (!x.sub) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.sub), (stride / 64) * 4 * sizeof(float)));
  if (!x.super) HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&x.super), (stride / kTile / kSuperTiles + 1) * 4 * sizeof(float)));
  launch_xt_fill(c->stream, x.planes, stride, x.perm, x.bf, x.table, x.bf_table);  // padding of every plane; both per-tile tables cleared
  return MI355_ICP_OK;
}

int ensure_xcopy(mi355_icp_ctx* c, DevicePlanes view, size_t stride, const std::vector<CloudEntry>& clouds, XCopy& x) {
  if (x.valid) return MI355_ICP_OK;
  const int n_clouds = static_cast<int>(clouds.size());
  uint32_t max_n = 1;
  // the (offset, count) table goes up from a pinned buffer of the engine's, without a synchronisation of its own: the buffer
  // is rewritten only after the previous build's copy has left it (an event, long since signalled in practice)
  if (c->xt_meta_pending) { HIP_TRY(c, hipEventSynchronize(c->xt_meta_event)); c->xt_meta_pending = false; }
  if (2 * static_cast<size_t>(n_clouds) > c->xt_meta_cap) {
    if (c->xt_meta_host) { HIP_TRY(c, hipHostFree(c->xt_meta_host)); c->xt_meta_host = nullptr; c->xt_meta_cap = 0; }
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->xt_meta_host), 4 * static_cast<size_t>(n_clouds) * sizeof(uint32_t), hipHostMallocDefault));
    c->xt_meta_cap = 4 * static_cast<size_t>(n_clouds);
  }
  if (!c->xt_meta_event) HIP_TRY(c, hipEventCreateWithFlags(&c->xt_meta_event, hipEventDisableTiming));
  uint32_t* meta = c->xt_meta_host;
  for (int k = 0; k < n_clouds; ++k) {
    meta[k] = clouds[k].off;
    meta[n_clouds + k] = clouds[k].n;
    max_n = std::max(max_n, clouds[k].n);
  }
  size_t temp_bytes = 0;
  if (xtarget_sort_temp_bytes(max_n, &temp_bytes) != 0) return fail(c, MI355_ICP_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
  int st;
  if ((st = grow(c, c->xt_scratch, c->xt_scratch_cap, 4 * static_cast<size_t>(max_n) + 2 * static_cast<size_t>(n_clouds))) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->xt_box, c->xt_box_cap, 6 * static_cast<size_t>(n_clouds))) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->xt_temp, c->xt_temp_cap, temp_bytes + 16)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->xt_tile_box, c->xt_tile_box_cap, 8 * (stride / kTile + 1))) != MI355_ICP_OK) return st;
  if ((st = alloc_xcopy(c, x, stride)) != MI355_ICP_OK) return st;
  uint32_t* code = c->xt_scratch;
  uint32_t* index = code + max_n;
  uint32_t* code2 = index + max_n;
  uint32_t* order = code2 + max_n;
  uint32_t* meta_dev = order + max_n;
  HIP_TRY(c, hipMemcpyAsync(meta_dev, meta, 2 * static_cast<size_t>(n_clouds) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->xt_meta_event, c->stream));
  c->xt_meta_pending = true;
  launch_bbox(c->stream, view, meta_dev, meta_dev + n_clouds, n_clouds, max_n, c->xt_box);
  for (int k = 0; k < n_clouds; ++k)
    if (xtarget_build_cloud(c->stream, view, clouds[k].off, clouds[k].n, c->xt_box + 6 * k, code, index, code2, order, c->xt_temp,
                            temp_bytes, x.planes, stride, x.perm, x.table, x.bf, x.sub, x.bf_table, c->xt_tile_box) != 0)
      return fail(c, MI355_ICP_ERR_HIP, "rocprim::radix_sort_pairs failed");
  launch_super_tiles(c->stream, x.table, static_cast<uint32_t>(stride / kTile), x.super);
  HIP_TRY(c, hipGetLastError());
  x.valid = true;
  return MI355_ICP_OK;
}

int ensure_xtarget(mi355_icp_ctx* c, CloudSet& set) { return ensure_xcopy(c, set.view(), set.stride, set.clouds, set.xt); }

// the sorted copy that was built from exactly these planes (level 0 or a pyramid level of `set`), if any
const XCopy* find_xcopy(const CloudSet& set, const float* planes, size_t* stride) {
  if (planes == set.planes) { *stride = set.stride; return &set.xt; }
  for (const LevelPlanes& lv : set.levels)
    if (planes == lv.planes) { *stride = lv.stride; return &lv.xt; }
  return nullptr;
}

// OPTIONAL pruned scan: index and dist2 of every point, straight from the sorted copies of both clouds
int run_pruned_index_phase(mi355_icp_ctx* c, int n_active, uint32_t sbx, DevicePlanes S, DevicePlanes T) {
  size_t sstride = 0, tstride = 0;
  const XCopy* xs = find_xcopy(c->sources, S.x, &sstride);
  const XCopy* xt = find_xcopy(c->targets, T.x, &tstride);
  if (!xs || !xt || !xs->valid || !xt->valid) return fail(c, MI355_ICP_ERR_INTERNAL, "pruned scan: sorted copies missing");
  const float* sp = xs->planes;
  const float* tp = xt->planes;
  launch_nn_pruned(c->stream, c->jobs_dev, n_active, sbx, DevicePlanes{sp + 4 * sstride, sp + 5 * sstride, sp + 6 * sstride}, xs->perm,
                   DevicePlanes{tp + 4 * tstride, tp + 5 * tstride, tp + 6 * tstride}, xt->perm, xt->table, xt->sub, xt->super,
                   c->out_index, c->out_dist2, c->error_flag);
  HIP_TRY(c, hipGetLastError());
  return MI355_ICP_OK;
}

// Overflow points per job the device-side pass takes before the host path has to: kOverflowCap, or 1/128 of the largest
// active source cloud where that is more (the matrix-core filter leaves ~0.25 % of a cloud unsettled per pass whatever
// its size - the band over the nearest-neighbour distance is scale-free; a fixed 1024 would send every pass of a
// 10^6-point model through the host).  The launch grid is this many workgroups per
// job; the ones past a job's count leave at once.
uint32_t overflow_cap(const mi355_icp_ctx* c, int n_active) {
  uint32_t n = 0;
  for (int k = 0; k < n_active; ++k) n = std::max(n, c->jobs_host[k].src_n);
  return std::max<uint32_t>(kOverflowCap, n / 128u);
}

// the sorted copy (expansion filters) that was built from exactly these target planes - level 0 or a pyramid level - if any
const XCopy* sorted_copy_of(const mi355_icp_ctx* c, const float* target_planes, size_t* stride) {
  const XCopy* found = nullptr;
  if (target_planes == c->targets.planes) { found = &c->targets.xt; if (stride) *stride = c->targets.stride; }
  for (const LevelPlanes& lv : c->targets.levels)
    if (target_planes == lv.planes) { found = &lv.xt; if (stride) *stride = lv.stride; }
  return found;
}

// What a deferred pass tells its tail: how many listed points per job the device-side overflow pass settles.  With a sorted copy
// of the targets that pass (overflow_tiles_kernel) strides over the WHOLE list of a job - there is nothing it leaves to the host,
// so no count may send the pass round again.  (Until round 4 the cap of the other pass was applied here too: harmless while the
// matrix-core filter listed 0.25 % of a cloud, a repeated 50 ms pass at 700k<->3M once the wider band listed 1.1 % -
// profiles/r04w_overflow_cap.txt.)  Without a sorted copy (the fused filter) overflow_exact_kernel takes `overflow_cap` points.
uint32_t pass_overflow_cap(const mi355_icp_ctx* c, const Plan& plan, int n_active, const float* target_planes) {
  const XCopy* x = plan.xfilter ? sorted_copy_of(c, target_planes, nullptr) : nullptr;
  return x && x->valid ? 0xFFFFFFFFu : overflow_cap(c, n_active);
}

// Fused-filter path up to "index and dist2 of every point are final": filter scan -> exact resolve of the candidate
// chunks -> (rarely) the all-exact scan for points whose candidates overflowed the band.  Records ev[1] after the scan.
// `defer`: the points the resolve could not settle are answered on the device (overflow_exact_kernel, up to kOverflowCap
// per job) and the counts are only copied back, to be looked at after the pass's final synchronisation
// (overflow_settled); otherwise the host reads the counts here and runs the gathered all-exact scan for them.
// staged != nullptr: the caller has NOT staged the job descriptors; the matrix-core scan does it itself (its kernel arguments),
// any other scan of this phase gets them from stage_jobs_kernel first.
int run_fused_index_phase(mi355_icp_ctx* c, const Plan& plan, int n_active, uint32_t bx, uint32_t sbx, DevicePlanes S,
                          DevicePlanes T, bool profile, bool defer, const StageArgs* staged = nullptr) {
  const uint32_t rank_stride = plan.key_rows() * plan.out_total;
  // the expansion-filter copy that was built from exactly these target planes (level 0 or a pyramid level), if any
  size_t xstride = 0;
  const XCopy* Xp = sorted_copy_of(c, T.x, &xstride);
  const bool xf = plan.xfilter && Xp && Xp->valid;
  const XCopy none;
  const XCopy& X = xf ? *Xp : none;
  // matrix-core scan: the Morton-ordered copy of these source planes (the scan and the resolve agree on positions; the
  // resolve writes results - and lists overflow points - at original indices, so everything downstream is unchanged)
  DevicePlanes Sscan = S;
  const uint32_t* src_perm = nullptr;
  if (xf && plan.sorted_src) {
    size_t sstride = 0;
    const XCopy* xs = find_xcopy(c->sources, S.x, &sstride);
    if (!xs || !xs->valid) return fail(c, MI355_ICP_ERR_INTERNAL, "matrix-core scan: sorted copy of the sources missing");
    Sscan = DevicePlanes{xs->planes + 4 * sstride, xs->planes + 5 * sstride, xs->planes + 6 * sstride};
    src_perm = xs->perm;
  }
  const bool self_staged = staged && xf && bx && plan.xkind == 3 && c->xbf16_shape != 1;
  if (staged && !self_staged &&
      stage_jobs(c->stream, c->jobs_host, staged->has_ref ? c->rjobs_host : nullptr, n_active, c->jobs_dev, staged->has_ref ? c->rjobs_dev : nullptr) != MI355_ICP_OK)
    return fail(c, MI355_ICP_ERR_HIP, "staging the job descriptors failed");
  if (xf) {
    const float* xp = X.planes;
    const size_t st = xstride;
    const int ppl = plan.scan_ppl == 2 ? 2 : 4;  // the MFMA kernel spans 128 * {4, 8} points per workgroup
    if (bx && plan.xkind == 3) {
      // The launch's layout is decided HERE, once, and travels as the kernel's template argument: static runs (2-D grid), or the
      // 3-D grid of the class-weighted layout (one-round launches; needs the self-staged kernel and every ACTIVE job to be one
      // of the plan's uniform batch) or of the ticket queue (any subset of the plan's jobs: a team is a source block of a job).
      int layout = kXbLayoutStatic;
      if (plan.queued && c->xbf16_shape != 1) {
        layout = kXbLayoutQueue;
        const size_t words = 2 * static_cast<size_t>(plan.grid3_x) * static_cast<size_t>(n_active);
        if (words > c->xb_queue_cap) {
          if (c->xb_queue) { HIP_TRY(c, hipFree(c->xb_queue)); c->xb_queue = nullptr; c->xb_queue_cap = 0; }
          HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->xb_queue), 2 * words * sizeof(uint32_t)));
          HIP_TRY(c, hipMemsetAsync(c->xb_queue, 0, 2 * words * sizeof(uint32_t), c->stream));  // every launch leaves its words zeroed again
          c->xb_queue_cap = 2 * words;
        }
      } else if (self_staged && plan.classed) {
        layout = kXbLayoutClassed;
        for (int k = 0; k < n_active; ++k)
          if (staged->jobs[k].first_class == 0 || staged->jobs[k].scan_blocks != plan.grid3_x || staged->jobs[k].n_split != plan.grid3_z) layout = kXbLayoutStatic;
      }
      launch_nn_scan_xbf16(c->stream, c->jobs_dev, n_active, bx, Sscan, X.bf, X.bf_table, c->keys3, plan.out_total, rank_stride, c->xbf16_shape,
                           self_staged ? staged : nullptr, c->jobs_dev, staged && staged->has_ref ? c->rjobs_dev : nullptr, layout, plan.grid3_x,
                           plan.grid3_z, c->xb_queue);
    } else if (bx && plan.xkind == 0) {
      launch_nn_scan_xfilter(c->stream, plan.scan_ppl, c->jobs_dev, n_active, bx, S, xp, xp + st, xp + 2 * st, xp + 3 * st,
                             X.table, c->keys3, plan.out_total, rank_stride);
    } else if (bx && plan.xkind == 1) {
      launch_nn_scan_xmfma(c->stream, ppl, c->jobs_dev, n_active, bx, 0, S, xp, xp + st, xp + 2 * st, xp + 3 * st, X.table,
                           c->keys3, plan.out_total, rank_stride);
    } else if (bx) {
      // both pipes at once: scan blocks [0, first) on the vector ALU, [first, ...) on the matrix cores, in two streams
      uint32_t max_scan_blocks = 0, split = 1;
      for (int k = 0; k < n_active; ++k) {
        max_scan_blocks = std::max(max_scan_blocks, c->jobs_host[k].scan_blocks);
        split = std::max(split, c->jobs_host[k].n_split);
      }
      const uint32_t first = static_cast<uint32_t>(static_cast<uint64_t>(max_scan_blocks) * (100 - c->mfma_share_pct) / 100);
      HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
      HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      if (first)
        launch_nn_scan_xfilter(c->stream, ppl, c->jobs_dev, n_active, first * split, S, xp, xp + st, xp + 2 * st, xp + 3 * st,
                               X.table, c->keys3, plan.out_total, rank_stride);
      launch_nn_scan_xmfma(c->stream2, ppl, c->jobs_dev, n_active, (max_scan_blocks - first) * split, first, S, xp, xp + st,
                           xp + 2 * st, xp + 3 * st, X.table, c->keys3, plan.out_total, rank_stride);
      HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
  } else if (bx) {
    launch_nn_scan_fused(c->stream, plan.scan_ppl, c->jobs_dev, n_active, bx, S, T, c->keys3, plan.out_total, rank_stride);
  }
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  if (sbx && xf) {
    const float* xp = X.planes;
    const size_t st = xstride;
    const bool xb = plan.xkind == 3;
    // points of the launch (padded): the resolve spreads a small launch over more waves, 16 or 32 points each instead of 64 -
    // a wave's time is one round trip per 8 points (6 x 20k points: 21.6 -> 8.2 us per wave, request 2.03 -> 1.94 ms;
    // 600k points do best with 32, from a million on the 64-point waves are as fast: profiles/r03q_resolve_wave_points.txt)
    const int wave_points = c->resolve_wave_points ? c->resolve_wave_points : (plan.out_total <= 300000u ? 16 : plan.out_total <= 1500000u ? 32 : 64);
    launch_resolve_xfilter(c->stream, xb ? kXbRun : kFusedChunk, xb ? 2u : 1u, wave_points, c->jobs_dev, n_active, sbx, xb ? Sscan : S, xb ? src_perm : nullptr,
                           DevicePlanes{xp + 4 * st, xp + 5 * st, xp + 6 * st}, X.perm,
                           c->keys3, plan.out_total, rank_stride, c->out_index, c->out_dist2, c->overflow_list, c->overflow_count,
                           c->error_flag);
  } else if (sbx) {
    launch_resolve_fused(c->stream, c->jobs_dev, n_active, sbx, S, T, c->keys3, plan.out_total, rank_stride, c->out_index,
                         c->out_dist2, c->partials, c->overflow_list, c->overflow_count, c->error_flag);
  }
  HIP_TRY(c, hipGetLastError());
  if (defer) {  // the counts reach the host with the sums (final-reduce kernels), after the pass's one synchronisation
    if (sbx && xf) {  // a sorted copy with tile spheres exists: only the tiles a point can still reach (any target size)
      const float* xp = X.planes;
      launch_overflow_tiles(c->stream, c->jobs_dev, n_active, S, DevicePlanes{xp + 4 * xstride, xp + 5 * xstride, xp + 6 * xstride},
                            X.perm, X.table, c->overflow_list, c->overflow_count, c->out_index, c->out_dist2, c->error_flag);
    } else if (sbx) {
      launch_overflow_exact(c->stream, c->jobs_dev, n_active, overflow_cap(c, n_active), S, T, c->overflow_list, c->overflow_count, c->out_index,
                            c->out_dist2, c->error_flag);
    }
    HIP_TRY(c, hipGetLastError());
    return MI355_ICP_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(c->overflow_count_host, c->overflow_count, static_cast<size_t>(n_active) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  bool any = false;
  for (int k = 0; k < n_active; ++k) any = any || c->overflow_count_host[k] != 0;
  if (any) return run_overflow_fallback(c, plan, n_active, c->overflow_count_host, S, T);
  return MI355_ICP_OK;
}

// After the final synchronisation of a pass that deferred its overflow handling: `counts` is the tail the final-reduce
// kernel wrote behind the sums (one double per job; the flags beside it say whether any exceeded what the device pass
// covers - then the caller repeats the pass with the host path and nothing is booked here).
void book_overflow(mi355_icp_ctx* c, const double* counts, int n_active) {
  for (int k = 0; k < n_active; ++k) {
    c->overflow_points_total += static_cast<uint64_t>(counts[k]);
    c->prof.overflow_points += static_cast<uint64_t>(counts[k]);
  }
}

// Is the device-side overflow pass the cheaper one?  With a sorted copy of the targets (every expansion filter) it scans
// only the tiles a point can still reach and always is.  The fused filter has no such copy: its pass streams the whole
// target cloud once per listed point (12 B per target and point, out of L2 / Infinity Cache), which beats a host round
// trip + gathered tiled scan only while the cloud is small (profiles/r01o_overflow_paths.txt).
bool defer_pays(const mi355_icp_ctx* c, const Plan& plan, int n_active) {
  if (plan.xfilter) return true;
  uint32_t tiles = 0;
  for (int k = 0; k < n_active; ++k) tiles = std::max(tiles, c->jobs_host[k].tgt_tiles);
  return tiles <= 256;
}

int run_pass_once(mi355_icp_ctx* c, const Plan& plan, int n_active, bool profile, bool local_only, bool defer, bool* retry, bool tail_allowed);

// Enqueues one correspondence pass for the first n_active descriptors of jobs_host and leaves
// their sums in sums_host (synchronises the stream).  `local_only` skips the collective.
// tail_allowed: the accumulation launch may end the pass by itself (PassTail) - the align loop; a correspondence query reads
// its outputs with blocking copies afterwards and keeps the copy + synchronisation.
int run_pass(mi355_icp_ctx* c, const Plan& plan, int n_active, bool profile, bool local_only, bool tail_allowed = false) {
  // Slab-sharded: the exchange sits between the kernels and the synchronisation, so a rank cannot decide on its own to
  // repeat a pass.  The per-job "my device-side overflow pass was too small" flags therefore travel WITH the sums;
  // every rank sees their total and all of them repeat the pass through the host path together.
  bool retry = false;
  int st = run_pass_once(c, plan, n_active, profile, local_only, plan.fused && defer_pays(c, plan, n_active), &retry, tail_allowed);
  if (st == MI355_ICP_OK && retry) st = run_pass_once(c, plan, n_active, profile, local_only, false, &retry, tail_allowed);
  return st;
}

int run_pass_once(mi355_icp_ctx* c, const Plan& plan, int n_active, bool profile, bool local_only, bool defer, bool* retry, bool tail_allowed) {
  *retry = false;
  // the all-exact scan of a small pass stages the descriptors itself (one dependent launch less)
  uint32_t bx = 0, sbx = 0;
  double pairs = 0.0;
  bool all_scan = true;
  for (int k = 0; k < n_active; ++k) {
    const JobDesc& J = c->jobs_host[k];
    bx = std::max(bx, J.scan_blocks * J.n_split);
    sbx = std::max(sbx, J.src_blocks);
    pairs += static_cast<double>(J.src_n) * static_cast<double>(J.tgt_n);
    all_scan = all_scan && J.scan_blocks * J.n_split > 0;
  }
  // small passes: the scan launch takes the descriptors as its kernel arguments and writes the device tables itself - the
  // all-exact scan always, the matrix-core scan below 5e10 pairs (above, a launch in front costs nothing that shows, and
  // the benchmark's kernel stays the instantiation it was measured as)
  StageArgs staged;
  const bool exact_path = !plan.fused && !plan.pruned && !c->scan_sgpr;
  const bool mfma_path = plan.fused && plan.xfilter && plan.xkind == 3 && !plan.pruned && pairs < 5e10;
  const bool self_staged = (exact_path || mfma_path) && all_scan && fill_stage_args(c->jobs_host, nullptr, n_active, &staged);
  if (!self_staged && stage_jobs(c->stream, c->jobs_host, nullptr, n_active, c->jobs_dev, nullptr) != MI355_ICP_OK) return fail(c, MI355_ICP_ERR_HIP, "staging the job descriptors failed");
  const DevicePlanes S = c->sources.view(), T = c->targets.view();
  // never launch against a buffer the planner did not size for this launch shape (an out-of-bounds kernel can take the GPU down)
  const size_t key_need = static_cast<size_t>(plan.fused ? plan.key_rows() : plan.max_split) * plan.out_total;
  if (plan.out_total > c->out_cap || static_cast<size_t>(plan.partial_total) * MI355_ICP_NSUMS > c->partials_cap ||
      (plan.pruned ? false : plan.fused ? (plan.key_ranks() * key_need > c->keys3_cap || plan.out_total > c->overflow_list_cap) : key_need > c->keys_cap))
    return fail(c, MI355_ICP_ERR_INTERNAL, "work buffers are smaller than the plan (ensure_buffers not run for this plan)");
  // single GPU, align loop: the accumulation launch ends the pass by itself (PassTail)
  const bool exchange = c->comm.attached() && !local_only;
  // an external transport (gloo in the tests; the in-process exchange of a multi-device handle) meets on the HOST anyway: the pass
  // may end itself and the rows are exchanged where finish_pass left them - no device-to-host-to-device round trip around the callback
  const bool host_exchange = exchange && c->comm.ext != nullptr;
  const PassTail tail = make_pass_tail(c, plan, n_active, sbx, tail_allowed && (!exchange || host_exchange), defer, c->p2p_done, c->sums_mapped, T.x);
  const bool tail_on = tail.done != nullptr;
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (plan.pruned) {
    const int fst = run_pruned_index_phase(c, n_active, sbx, S, T);
    if (fst != MI355_ICP_OK) return fst;
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
    if (sbx) launch_accumulate(c->stream, c->jobs_dev, n_active, sbx, S, T, c->out_index, c->out_dist2, c->partials, tail);
  } else if (plan.fused) {
    const int fst = run_fused_index_phase(c, plan, n_active, bx, sbx, S, T, profile, defer, self_staged ? &staged : nullptr);
    if (fst != MI355_ICP_OK) return fst;
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
    if (sbx) launch_accumulate(c->stream, c->jobs_dev, n_active, sbx, S, T, c->out_index, c->out_dist2, c->partials, tail);
  } else {
    if (plan.merged_keys && plan.out_total) HIP_TRY(c, hipMemsetAsync(c->keys, 0xFF, static_cast<size_t>(plan.out_total) * sizeof(unsigned long long), c->stream));
    c->keys_clean = 0;
    if (bx) launch_nn_scan(c->stream, plan.scan_ppl + (c->scan_sgpr ? 100 : 0), c->jobs_dev, n_active, bx, S, T, c->keys, plan.out_total,
                           self_staged ? &staged : nullptr, c->jobs_dev, nullptr);
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    if (sbx) launch_resolve(c->stream, c->jobs_dev, n_active, sbx, S, T, c->keys, plan.out_total, c->out_index, c->out_dist2, c->partials, c->error_flag, tail);
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  }
  // the pass result, sums_dev or (PassTail) sums_host directly: [n x 17 sums][n overflow-excess flags][n overflow counts]
  if (!tail_on) launch_final_reduce(c->stream, c->jobs_dev, n_active, c->partials, c->sums_dev, plan.fused ? c->overflow_count : nullptr,
                                    defer ? pass_overflow_cap(c, plan, n_active, T.x) : 0xFFFFFFFFu);
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[3], c->stream));
  HIP_TRY(c, hipGetLastError());
  const size_t n_sums = static_cast<size_t>(n_active) * MI355_ICP_NSUMS, n_act = static_cast<size_t>(n_active);
  if (exchange && !tail_on) {  // the rows and the flags: every rank learns whether ANY rank's device-side overflow pass was too small
    const char* why = nullptr;
    const int cst = comm_sum_f64_rank_order(c->comm, c->sums_dev, n_sums + n_act, c->stream, &why);
    if (cst != MI355_ICP_OK) return fail(c, cst == MI355_ICP_ERR_HIP ? MI355_ICP_ERR_HIP : MI355_ICP_ERR_COMM, std::string("cross-rank exchange: ") + (why ? why : "?"));
  }
  if (tail_on) {
    if (const int sst = wait_pass(c, tail.seq, n_active, &c->error_word); sst != MI355_ICP_OK) return sst;
    c->error_word_known = true;
    if (profile) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the events are read below
    if (exchange) {  // host_exchange: the same rows and flags, summed in rank order where the pass left them
      const char* why = nullptr;
      const int cst = comm_sum_f64_rank_order_host(c->comm, c->sums_host, n_sums + n_act, &why);
      if (cst != MI355_ICP_OK) return fail(c, cst, std::string("cross-rank exchange: ") + (why ? why : "?"));
    }
  } else {
    HIP_TRY(c, hipMemcpyAsync(c->sums_host, c->sums_dev, (n_sums + 2 * n_act) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (const int sst = sync_pass(c); sst != MI355_ICP_OK) return sst;
  }
  if (defer || exchange) {
    // The decision must be the same on every rank (the flags are summed over the ranks) - including ranks whose own
    // plan did not defer anything (another scan kernel for their slab size): they repeat the pass with the others, or the
    // next collective would wait for them for ever.
    bool settled = true;
    for (size_t k = 0; k < n_act; ++k) settled = settled && c->sums_host[n_sums + k] == 0.0;
    if (!settled) {
      *retry = true;
      return MI355_ICP_OK;
    }
    if (defer) book_overflow(c, c->sums_host + n_sums + n_act, n_active);
  }
  c->prof.iterations += 1;
  c->prof.pairs_evaluated += pairs;
  if (profile) {
    float ms = 0.f;
    if (bx) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->prof.nn_scan_ms += ms; c->prof.nn_scan_launches += 1; }
    if (sbx) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->prof.resolve_ms += ms; c->prof.resolve_launches += 1; }
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->prof.reduce_ms += ms; c->prof.reduce_launches += 1;
  }
  return MI355_ICP_OK;
}

int check_error_flag(mi355_icp_ctx* c) {
  if (c->error_word_known) {  // the last pass was a reference-mode one that ended itself: it read the word for us
    c->error_word_known = false;
    if (c->error_word) c->err = "a non-finite distance was met during the scan (overflowing pose or coordinates)";
    c->error_flag_clean = c->error_word == 0;
    return MI355_ICP_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(c->flag_host, c->error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (*c->flag_host) c->err = "a non-finite distance was met during the scan (overflowing pose or coordinates)";
  c->error_flag_clean = *c->flag_host == 0;
  return MI355_ICP_OK;
}

// Start of a call that may raise the error word: clear it unless the previous call read it as zero (a fill is a launch of its
// own in front of the call's first kernel).  From here on the word counts as dirty until check_error_flag has read it again.
int reset_error_flag(mi355_icp_ctx* c) {
  if (!c->error_flag_clean) HIP_TRY(c, hipMemsetAsync(c->error_flag, 0, sizeof(uint32_t), c->stream));
  c->error_flag_clean = false;
  return MI355_ICP_OK;
}


// =====================================================================================================
// Reference mode: pyramid + MAD rejection + picky + point-to-plane, batched over jobs.

// slot of the copy of `set` that keeps every step-th point: 0 = the set itself (step 1), k + 1 = set.levels[k], -1 = not built
int level_slot(const CloudSet& set, uint32_t step) {
  if (step == 1) return 0;
  for (size_t k = 0; k < set.levels.size(); ++k)
    if (set.levels[k].step == step) return static_cast<int>(k) + 1;
  return -1;
}

// Builds (once per upload, and again when the sharding changes) the subsampled planes of every cloud of `set` for each step of
// `steps` that is missing.  shard_begin / shard_total: this rank's slab of every cloud (nullptr = the clouds are whole).
// want_rows: a set with normals also gets its 32-byte rows from the same launch (the targets always; the sources only for the
// upstream-faithful variant - ensure_rows builds a missing level's rows on demand either way).
int ensure_levels(mi355_icp_ctx* c, CloudSet& set, const std::vector<uint32_t>& steps, uint32_t pad_multiple, float pad_value, bool want_rows,
                  const std::vector<uint64_t>* shard_begin = nullptr, const std::vector<uint64_t>* shard_total = nullptr) {
  const std::vector<uint64_t> none;
  const std::vector<uint64_t>& want_begin = shard_begin ? *shard_begin : none;
  const std::vector<uint64_t>& want_total = shard_total ? *shard_total : none;
  if (want_begin != set.shard_begin || want_total != set.shard_total) {  // levels built for another slab geometry: drop them
    for (auto& lv : set.levels) {
      c->pool.give(lv.planes);
      c->pool.give(lv.rows);
      free_xcopy(c->pool, lv.xt);
    }
    set.levels.clear();
    set.shard_begin = want_begin;
    set.shard_total = want_total;
  }
  // every missing level at once: reference mode starts at the coarsest level, so a new scene asks for all of them in its
  // first call - one launch per cloud instead of one per cloud and level (7 small dependent launches per request)
  std::vector<uint32_t> missing;
  for (const uint32_t step : steps)
    if (step > 1 && level_slot(set, step) < 0 && std::find(missing.begin(), missing.end(), step) == missing.end()) missing.push_back(step);
  if (missing.empty()) return MI355_ICP_OK;
  std::sort(missing.begin(), missing.end());
  std::vector<LevelPlanes> fresh;
  auto give_back = [&]() {
    for (auto& lv : fresh) {
      c->pool.give(lv.rows);
      c->pool.give(lv.planes);
    }
  };
  for (const uint32_t step : missing) {
    LevelPlanes lv;
    lv.step = step;
    uint64_t total = 0;
    for (size_t k = 0; k < set.clouds.size(); ++k) {
      const CloudEntry& e = set.clouds[k];
      uint64_t f = 0, n = e.n / step, base = 0, n_all = e.n / step;
      if (!set.shard_begin.empty()) level_slab(set.shard_total[k], set.shard_begin[k], e.n, step, &f, &n, &base, &n_all);
      const uint64_t padded = (n + pad_multiple - 1) / pad_multiple * pad_multiple;
      lv.clouds.push_back(CloudEntry{static_cast<uint32_t>(total), static_cast<uint32_t>(n), static_cast<uint32_t>(padded)});
      lv.index_base.push_back(static_cast<uint32_t>(base));
      lv.n_total.push_back(static_cast<uint32_t>(n_all));
      lv.first.push_back(static_cast<uint32_t>(f));
      total += padded;
    }
    lv.stride = static_cast<size_t>(std::max<uint64_t>(total, pad_multiple));
    hipError_t e = c->pool.take(reinterpret_cast<void**>(&lv.planes), 3 * lv.stride * sizeof(float));
    // a set with normals: the level's rows (coordinates + normal, what the point-to-plane accumulate reads) come out of the same launch
    if (e == hipSuccess && set.normals && want_rows) e = c->pool.take(&lv.rows, lv.stride * 32);
    if (e == hipSuccess && total == 0) e = hipMemsetAsync(lv.planes, 0, 3 * lv.stride * sizeof(float), c->stream);
    fresh.push_back(lv);
    if (e != hipSuccess) {
      give_back();
      return fail(c, MI355_ICP_ERR_HIP, std::string("building pyramid level: ") + hipGetErrorString(e));
    }
  }
  for (size_t k = 0; k < set.clouds.size(); ++k)
    for (size_t q0 = 0; q0 < fresh.size(); q0 += kSubsampleMaxLevels) {
      SubsampleArgs A{};
      A.n_levels = static_cast<int>(std::min<size_t>(kSubsampleMaxLevels, fresh.size() - q0));
      for (int q = 0; q < A.n_levels; ++q) {
        const LevelPlanes& lv = fresh[q0 + static_cast<size_t>(q)];
        A.lv[q] = SubsampleLevel{lv.planes, lv.rows, static_cast<uint32_t>(lv.stride), lv.clouds[k].off, set.clouds[k].off + lv.first[k],
                                 lv.clouds[k].n, lv.clouds[k].padded, lv.step};
      }
      launch_subsample_levels(c->stream, set.planes, set.stride, A, pad_value, set.normals);
    }
  if (const hipError_t e = hipGetLastError(); e != hipSuccess) {  // the levels are not adopted: give their planes back
    give_back();
    return fail(c, MI355_ICP_ERR_HIP, std::string("building pyramid levels: ") + hipGetErrorString(e));
  }
  for (auto& lv : fresh) set.levels.push_back(lv);
  return MI355_ICP_OK;
}

// the expansion-filter copy of slot `slot` of the targets (slot 0: the uploaded clouds), built once per upload
// The sorted copy of a pyramid level is DERIVED from the level-0 copy (xtarget_build_cloud_from_parent: the level's points in
// the parent's Morton order, an order-preserving compaction instead of a radix sort) - MI355_ICP_LEVEL_SORT=1 sorts each
// level by itself as rounds 1-2 did (A/B; same results either way, the order of a sorted copy never reaches an output).
int ensure_level_xcopy(mi355_icp_ctx* c, CloudSet& set, int slot) {
  if (slot == 0) return ensure_xtarget(c, set);
  LevelPlanes& lv = set.levels[static_cast<size_t>(slot) - 1];
  if (lv.xt.valid) return MI355_ICP_OK;
  if (c->level_sort) return ensure_xcopy(c, lv.view(), lv.stride, lv.clouds, lv.xt);
  int st = ensure_xtarget(c, set);
  if (st != MI355_ICP_OK) return st;
  size_t scratch = 1;
  for (size_t k = 0; k < lv.clouds.size(); ++k)
    scratch = std::max(scratch, static_cast<size_t>(lv.clouds[k].n) + set.clouds[k].n / 4096 + 2);
  if ((st = grow(c, c->xt_scratch, c->xt_scratch_cap, scratch)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->xt_tile_box, c->xt_tile_box_cap, 8 * (lv.stride / kTile + 1))) != MI355_ICP_OK) return st;
  if ((st = alloc_xcopy(c, lv.xt, lv.stride)) != MI355_ICP_OK) return st;
  const DevicePlanes view = lv.view();
  for (size_t k = 0; k < lv.clouds.size(); ++k)
    if (xtarget_build_cloud_from_parent(c->stream, view, lv.clouds[k].off, lv.clouds[k].n, set.xt.perm + set.clouds[k].off, set.clouds[k].n,
                                        lv.first[k], lv.step, c->xt_scratch, lv.xt.planes, lv.stride, lv.xt.perm, lv.xt.table, lv.xt.bf,
                                        lv.xt.sub, lv.xt.bf_table, c->xt_tile_box) != 0)
      return fail(c, MI355_ICP_ERR_HIP, "deriving a pyramid level's sorted copy failed");
  launch_super_tiles(c->stream, lv.xt.table, static_cast<uint32_t>(lv.stride / kTile), lv.xt.super);
  HIP_TRY(c, hipGetLastError());
  lv.xt.valid = true;
  return MI355_ICP_OK;
}

// The 32-byte records (coordinates + normal) of slot `slot` of a set uploaded with normals (ensure_levels has run): what the
// point-to-plane accumulate gathers per pair from the targets and - upstream-faithful variant - reads per point from the sources.
// Built once per upload from the level's planes and the level-0 normals (levels get theirs from the subsample launch).
int ensure_rows(mi355_icp_ctx* c, CloudSet& set, int slot, const void** rows_out) {
  void*& rows = slot ? set.levels[static_cast<size_t>(slot) - 1].rows : set.rows;
  if (!rows) {
    const size_t stride = slot ? set.levels[static_cast<size_t>(slot) - 1].stride : set.stride;
    const std::vector<CloudEntry>& clouds = slot ? set.levels[static_cast<size_t>(slot) - 1].clouds : set.clouds;
    const DevicePlanes view = slot ? set.levels[static_cast<size_t>(slot) - 1].view() : set.view();
    const uint32_t step = slot ? set.levels[static_cast<size_t>(slot) - 1].step : 1u;
    void* fresh = nullptr;
    HIP_TRY(c, c->pool.take(&fresh, std::max<size_t>(stride, 1) * 32));
    for (size_t k = 0; k < clouds.size(); ++k) {
      const uint32_t first = slot ? set.levels[static_cast<size_t>(slot) - 1].first[k] : 0u;
      launch_target_rows(c->stream, view, clouds[k].off, clouds[k].n, set.normal_view(), set.clouds[k].off + first, step, fresh);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      c->pool.give(fresh);
      return fail(c, MI355_ICP_ERR_HIP, std::string("building the point rows: ") + hipGetErrorString(e));
    }
    rows = fresh;
  }
  *rows_out = rows;
  return MI355_ICP_OK;
}

const std::vector<CloudEntry>& level_clouds(const CloudSet& set, int slot) { return slot ? set.levels[slot - 1].clouds : set.clouds; }
DevicePlanes level_view(const CloudSet& set, int slot) { return slot ? set.levels[slot - 1].view() : set.view(); }

// Which copies of the two cloud sets a reference-mode pass runs on, and in which form.
struct RefLevel {
  int sslot = 0, tslot = 0;  // level_slot of the sources / of the targets for the level's step
  bool upstream = false;     // MI355_ICP_FLAG_UPSTREAM: model rows with normals, plane equations about RefJobDesc::centre
  uint32_t offer_cap = 0;    // slab-sharded: the most level points any rank holds over the jobs of this level (every rank computes the
                             // same number from the slab geometry): the size of a rank's offer list, see run_ref_pass_once
};

int ensure_ref_buffers(mi355_icp_ctx* c, const Plan& plan, size_t n_jobs, size_t owner_total) {
  int st;
  {
    // a re-grown table holds garbage: clear it to "never written" right here.  The pass tag is NOT touched: it must be a
    // function of the call's pass sequence alone, or ranks whose engines grew their tables at different times (different
    // histories) would tag the same pass differently and the min-reduction of a slab-sharded run would pick by tag.
    const size_t before = c->owner_cap;
    if ((st = grow(c, c->owner, c->owner_cap, std::max<size_t>(owner_total, 1))) != MI355_ICP_OK) return st;
    if (c->owner_cap != before) HIP_TRY(c, hipMemsetAsync(c->owner, 0xFF, c->owner_cap * sizeof(unsigned long long), c->stream));
  }
  if ((st = grow(c, c->selected, c->selected_cap, std::max<size_t>(plan.out_total, 1))) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->ref_partials, c->ref_partials_cap, static_cast<size_t>(std::max<uint32_t>(plan.partial_total, 1)) * kRefSums)) != MI355_ICP_OK) return st;
  if (n_jobs > c->ref_jobs_cap) {
    if (c->ref_sums_dev) { HIP_TRY(c, hipFree(c->ref_sums_dev)); c->ref_sums_dev = nullptr; }
    if (c->thr_dev) { HIP_TRY(c, hipFree(c->thr_dev)); c->thr_dev = nullptr; }
    if (c->rjobs_dev) { HIP_TRY(c, hipFree(c->rjobs_dev)); c->rjobs_dev = nullptr; }
    if (c->rjobs_host) { HIP_TRY(c, hipHostFree(c->rjobs_host)); c->rjobs_host = nullptr; }
    if (c->ref_sums_host) { HIP_TRY(c, hipHostFree(c->ref_sums_host)); c->ref_sums_host = nullptr; }
    if (c->thr_host) { HIP_TRY(c, hipHostFree(c->thr_host)); c->thr_host = nullptr; }
    c->ref_jobs_cap = 0;
    const size_t want = std::max<size_t>(n_jobs, 8) * 2;
    // + three doubles per job behind the rows: overflow-excess flag, rejection threshold, overflow count (ref_final_reduce_kernel)
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->ref_sums_dev), want * (kRefSums + 3) * sizeof(double)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->thr_dev), want * sizeof(float)));
    if (c->select_work) { HIP_TRY(c, hipFree(c->select_work)); c->select_work = nullptr; }
    // [want x 2048 histogram bins][want x 4 words of selection state]: the histograms form one block a slab-sharded pass
    // adds up across ranks
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->select_work), want * (2048 + 4) * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->select_work, 0, want * (2048 + 4) * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->rjobs_dev), want * sizeof(RefJobDesc)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->rjobs_host), want * sizeof(RefJobDesc), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->ref_sums_host), want * (kRefSums + 3) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->ref_sums_mapped), c->ref_sums_host, 0));
    if (c->pass_done) { HIP_TRY(c, hipFree(c->pass_done)); c->pass_done = nullptr; }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->pass_done), want * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->pass_done, 0, want * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->thr_host), want * sizeof(float), hipHostMallocDefault));
    c->ref_jobs_cap = want;
  }
  return MI355_ICP_OK;
}

// One reference-mode pass for the first n_active descriptors of jobs_host / rjobs_host on pyramid level `level`.
// Leaves the 29 sums per job in ref_sums_host and the thresholds in thr_host (synchronises the stream).
int run_ref_pass_once(mi355_icp_ctx* c, const Plan& plan, const RefLevel& level, int n_active, float rejection_scale, size_t owner_total,
                      bool profile, bool defer, bool* retry);

int run_ref_pass(mi355_icp_ctx* c, const Plan& plan, const RefLevel& level, int n_active, float rejection_scale, size_t owner_total,
                 bool profile) {
  bool retry = false;
  int st = run_ref_pass_once(c, plan, level, n_active, rejection_scale, owner_total, profile, plan.fused && defer_pays(c, plan, n_active), &retry);
  if (st == MI355_ICP_OK && retry) st = run_ref_pass_once(c, plan, level, n_active, rejection_scale, owner_total, profile, false, &retry);
  return st;
}

int run_ref_pass_once(mi355_icp_ctx* c, const Plan& plan, const RefLevel& level, int n_active, float rejection_scale, size_t owner_total,
                      bool profile, bool defer, bool* retry) {
  *retry = false;
  // this pass's tag of the owner keys: one smaller prefix per pass, the table cleared once per 63 passes (picky_key)
  if (c->owner_epoch == 0 || c->owner_epoch >= 63) {
    HIP_TRY(c, hipMemsetAsync(c->owner, 0xFF, c->owner_cap * sizeof(unsigned long long), c->stream));
    c->owner_epoch = 0;
  }
  c->owner_epoch += 1;
  for (int k = 0; k < n_active; ++k) c->rjobs_host[k].owner_prefix = 63u - c->owner_epoch;
  uint32_t bx = 0, sbx = 0;
  double pairs = 0.0;
  bool all_scan = true;
  for (int k = 0; k < n_active; ++k) {
    const JobDesc& J = c->jobs_host[k];
    bx = std::max(bx, J.scan_blocks * J.n_split);
    sbx = std::max(sbx, J.src_blocks);
    pairs += static_cast<double>(J.src_n) * static_cast<double>(J.tgt_n);
    all_scan = all_scan && J.scan_blocks * J.n_split > 0;  // the scan must launch for every job
  }
  // small passes: the scan launch stages the descriptors itself (one dependent launch less; see run_pass_once)
  StageArgs staged;
  const bool exact_path = !plan.fused && !plan.pruned && !c->scan_sgpr;
  const bool mfma_path = plan.fused && plan.xfilter && plan.xkind == 3 && !plan.pruned && pairs < 5e10;
  const bool self_staged = (exact_path || mfma_path) && all_scan && fill_stage_args(c->jobs_host, c->rjobs_host, n_active, &staged);
  if (!self_staged && stage_jobs(c->stream, c->jobs_host, c->rjobs_host, n_active, c->jobs_dev, c->rjobs_dev) != MI355_ICP_OK) return fail(c, MI355_ICP_ERR_HIP, "staging the job descriptors failed");
  const DevicePlanes S = level_view(c->sources, level.sslot), T = level_view(c->targets, level.tslot);
  const void *target_rows = nullptr, *source_rows = nullptr;  // built by the level's first pass after an upload
  if (const int rst = ensure_rows(c, c->targets, level.tslot, &target_rows); rst != MI355_ICP_OK) return rst;
  if (level.upstream)
    if (const int rst = ensure_rows(c, c->sources, level.sslot, &source_rows); rst != MI355_ICP_OK) return rst;
  const size_t ref_key_need = static_cast<size_t>(plan.fused ? plan.key_rows() : plan.max_split) * plan.out_total;
  if ((plan.pruned ? false : plan.fused ? (plan.key_ranks() * ref_key_need > c->keys3_cap || plan.out_total > c->overflow_list_cap) : ref_key_need > c->keys_cap) ||
      plan.out_total > c->out_cap || plan.out_total > c->selected_cap || owner_total > c->owner_cap ||
      static_cast<size_t>(plan.partial_total) * kRefSums > c->ref_partials_cap ||
      static_cast<size_t>(plan.partial_total) * MI355_ICP_NSUMS > c->partials_cap)
    return fail(c, MI355_ICP_ERR_INTERNAL, "work buffers are smaller than the plan (reference mode)");
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (plan.pruned) {
    const int fst = run_pruned_index_phase(c, n_active, sbx, S, T);
    if (fst != MI355_ICP_OK) return fst;
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  } else if (plan.fused) {
    const int fst = run_fused_index_phase(c, plan, n_active, bx, sbx, S, T, profile, defer, self_staged ? &staged : nullptr);
    if (fst != MI355_ICP_OK) return fst;
  } else {
    // merged key rows come back cleared from the resolve (which runs whenever the scan does): one memset per plan, not per pass
    if (plan.merged_keys && plan.out_total && c->keys_clean < plan.out_total) {
      HIP_TRY(c, hipMemsetAsync(c->keys, 0xFF, static_cast<size_t>(plan.out_total) * sizeof(unsigned long long), c->stream));
      c->keys_clean = plan.out_total;
    }
    if (!plan.merged_keys) c->keys_clean = 0;
    if (bx) launch_nn_scan(c->stream, plan.scan_ppl + (c->scan_sgpr ? 100 : 0), c->jobs_dev, n_active, bx, S, T, c->keys, plan.out_total,
                           self_staged ? &staged : nullptr, c->jobs_dev, c->rjobs_dev);
    if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    if (sbx) launch_resolve(c->stream, c->jobs_dev, n_active, sbx, S, T, c->keys, plan.out_total, c->out_index, c->out_dist2, nullptr, c->error_flag);
    else if (bx) c->keys_clean = 0;
  }
  const bool sharded = c->comm.attached();
  const char* why = nullptr;
  bool scattered = false;  // the selection launch offered the accepted pairs to the owner table as well
  if (rejection_scale > 0.0f) {
    uint32_t max_n = 0;
    for (int k = 0; k < n_active; ++k) max_n = std::max(max_n, c->jobs_host[k].src_n);
    if (!sharded) {
      scattered = launch_rejection_select(c->stream, c->jobs_dev, n_active, max_n, c->out_dist2, rejection_scale, c->thr_dev, c->select_work,
                                          c->select_work + c->ref_jobs_cap * 2048, c->rjobs_dev, c->out_index, c->owner);
    } else {
      // slab-sharded: the median and the MAD are those of ALL ranks' distances - every rank histograms its slab, the
      // 2048-bin counts are added across ranks (integers: exact in any order), every rank picks the same digit
      for (int pass = 0; pass < 6; ++pass) {
        launch_select_hist(c->stream, c->jobs_dev, n_active, max_n, c->out_dist2, c->select_work, c->select_work + c->ref_jobs_cap * 2048, pass);
        const int cst = comm_sum_u32(c->comm, c->select_work, static_cast<size_t>(n_active) * 2048, c->stream, &why);
        if (cst != MI355_ICP_OK) return fail(c, cst, std::string("cross-rank histogram: ") + (why ? why : "?"));
        launch_select_pick(c->stream, c->jobs_dev, n_active, c->select_work, c->select_work + c->ref_jobs_cap * 2048, pass, rejection_scale, c->thr_dev);
      }
    }
  } else {
    for (int k = 0; k < n_active; ++k) c->thr_host[k] = std::numeric_limits<float>::infinity();
    HIP_TRY(c, hipMemcpyAsync(c->thr_dev, c->thr_host, static_cast<size_t>(n_active) * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  // single GPU: the accumulate launch ends the pass by itself (PassTail)
  const bool host_exchange = sharded && c->comm.ext != nullptr;  // see run_pass_once
  const PassTail tail = make_pass_tail(c, plan, n_active, sbx, !sharded || host_exchange, defer, c->pass_done, c->ref_sums_mapped, T.x);
  const bool tail_on = tail.done != nullptr;
  if (sbx || sharded) {
    // Slab-sharded: targets are replicated, the owner of a target is the smallest key ANY rank offered.  Two exchanges give the
    // same table: min-reducing the whole table (8 B x level targets), or all-gathering the ranks' OFFERS (16 B x the level points
    // of the whole model) and applying them locally - bytes in proportion to the accepted points instead of the scene.  The
    // cheaper one by bytes is taken; every rank computes the same sizes, hence the same choice.
    const size_t offer_words = 1 + 2 * static_cast<size_t>(level.offer_cap);
    const bool offers = sharded && c->picky_offers && level.offer_cap > 0 &&
                        offer_words * sizeof(unsigned long long) * static_cast<size_t>(c->comm.size) < std::max<size_t>(owner_total, 1) * sizeof(unsigned long long);
    if (offers) {
      const size_t need = offer_words * (static_cast<size_t>(c->comm.size) + 1);
      if (need > c->offer_words_cap) {
        if (c->offer_block) { HIP_TRY(c, hipFree(c->offer_block)); c->offer_block = nullptr; c->offer_words_cap = 0; }
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->offer_block), 2 * need * sizeof(unsigned long long)));
        c->offer_words_cap = 2 * need;
      }
      c->offer_all = c->offer_block + offer_words;
      HIP_TRY(c, hipMemsetAsync(c->offer_block, 0, sizeof(unsigned long long), c->stream));
      if (sbx) launch_picky_offer(c->stream, c->jobs_dev, c->rjobs_dev, n_active, sbx, c->out_index, c->out_dist2, c->thr_dev, c->owner, c->offer_block, level.offer_cap);
      const int cst = comm_allgather_device(c->comm, c->offer_block, c->offer_all, offer_words * sizeof(unsigned long long), c->stream, &why);
      if (cst != MI355_ICP_OK) return fail(c, cst, std::string("cross-rank picky offers: ") + (why ? why : "?"));
      launch_picky_apply(c->stream, c->offer_all, static_cast<uint32_t>(offer_words), c->comm.size, c->comm.rank, level.offer_cap, c->owner);
    } else {
      if (sbx && !scattered) launch_picky_scatter(c->stream, c->jobs_dev, c->rjobs_dev, n_active, sbx, c->out_index, c->out_dist2, c->thr_dev, c->owner);
      if (sharded) {
        const int cst = comm_min_u64(c->comm, c->owner, std::max<size_t>(owner_total, 1), c->stream, &why);
        if (cst != MI355_ICP_OK) return fail(c, cst, std::string("cross-rank owner table: ") + (why ? why : "?"));
      }
    }
    if (sbx) launch_ref_accumulate(c->stream, c->jobs_dev, c->rjobs_dev, n_active, sbx, S, source_rows, target_rows, c->out_index,
                                   c->out_dist2, c->thr_dev, c->owner, c->want_selected ? c->selected : nullptr, c->ref_partials, tail);
  }
  // the pass result, ref_sums_dev or (PassTail) ref_sums_host directly: [n x 29 sums][n overflow-excess flags][n thresholds][n overflow counts]
  if (!tail_on) launch_ref_final_reduce(c->stream, c->jobs_dev, n_active, c->ref_partials, c->ref_sums_dev, c->thr_dev, plan.fused ? c->overflow_count : nullptr,
                                        defer ? pass_overflow_cap(c, plan, n_active, T.x) : 0xFFFFFFFFu);
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipGetLastError());
  const size_t n_sums = static_cast<size_t>(n_active) * kRefSums, n_act = static_cast<size_t>(n_active);
  if (sharded && !tail_on) {
    const int cst = comm_sum_f64_rank_order(c->comm, c->ref_sums_dev, n_sums + n_act, c->stream, &why);
    if (cst != MI355_ICP_OK) return fail(c, cst == MI355_ICP_ERR_HIP ? MI355_ICP_ERR_HIP : MI355_ICP_ERR_COMM, std::string("cross-rank exchange: ") + (why ? why : "?"));
  }
  if (tail_on) {
    if (const int sst = wait_pass(c, tail.seq, n_active, &c->error_word); sst != MI355_ICP_OK) return sst;
    c->error_word_known = true;
    if (profile) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the events are read below
    if (sharded) {
      const int cst = comm_sum_f64_rank_order_host(c->comm, c->ref_sums_host, n_sums + n_act, &why);
      if (cst != MI355_ICP_OK) return fail(c, cst, std::string("cross-rank exchange: ") + (why ? why : "?"));
    }
  } else {
    HIP_TRY(c, hipMemcpyAsync(c->ref_sums_host, c->ref_sums_dev, (n_sums + 3 * n_act) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (const int sst = sync_pass(c); sst != MI355_ICP_OK) return sst;
  }
  for (size_t k = 0; k < n_act; ++k) c->thr_host[k] = static_cast<float>(c->ref_sums_host[n_sums + n_act + k]);
  if (defer || sharded) {  // every rank takes the same decision (see run_pass_once)
    bool settled = true;
    for (size_t k = 0; k < n_act; ++k) settled = settled && c->ref_sums_host[n_sums + k] == 0.0;
    if (!settled) {
      *retry = true;
      return MI355_ICP_OK;
    }
    if (defer) book_overflow(c, c->ref_sums_host + n_sums + 2 * n_act, n_active);
  }
  c->prof.iterations += 1;
  c->prof.pairs_evaluated += pairs;
  if (profile) {
    float ms = 0.f;
    if (bx) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->prof.nn_scan_ms += ms; c->prof.nn_scan_launches += 1; }
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->prof.resolve_ms += ms; c->prof.resolve_launches += 1;
  }
  return MI355_ICP_OK;
}

int check_reference_inputs(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids) {
  if (c->targets.clouds.empty() || c->sources.clouds.empty()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference mode: upload clouds first");
  if (!c->targets.normals)
    return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT,
                "reference mode needs target normals: upload the scene as rows of >= 6 float32 (xyz + normal, stride_bytes >= 24)");
  for (int k = 0; k < n_jobs; ++k) {
    const int s = source_ids ? source_ids[k] : 0, t = target_ids ? target_ids[k] : 0;
    if (s < 0 || static_cast<size_t>(s) >= c->sources.clouds.size()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "source id out of range");
    if (t < 0 || static_cast<size_t>(t) >= c->targets.clouds.size()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "target id out of range");
  }
  return MI355_ICP_OK;
}

// ---- upstream's pre-normalisation (MI355_ICP_FLAG_UPSTREAM): centre and scale per candidate ------------------------------
// pose in cloud coordinates of the centred-frame pose Xc: Tr(c) Xc Tr(-c) T0 (oracle_from_centred)
void from_centred(const double* Xc, const float* centre, const double* T0, double* out) {
  double M[16];
  std::memcpy(M, Xc, sizeof(M));
  for (int i = 0; i < 3; ++i) {
    const double rc = Xc[4 * i] * static_cast<double>(centre[0]) + Xc[4 * i + 1] * static_cast<double>(centre[1]) + Xc[4 * i + 2] * static_cast<double>(centre[2]);
    M[4 * i + 3] = (Xc[4 * i + 3] + static_cast<double>(centre[i])) - rc;
  }
  mat4_mul(M, T0, out);
}

struct Normalisation {
  std::vector<float> centre;     // 3 per job: (float)meanAvg
  std::vector<double> scale;     // per job, valid after finish()
  std::vector<uint64_t> n_model; // per job: rows of the WHOLE model (all ranks)
  bool pending = false;
};

// Enqueues what the candidates' centres and scales need: centres on the host at once (cloud sums were taken at upload), the
// distance sums as one small launch per 8 jobs whose partial rows come back behind an event (normalisation_finish).
int normalisation_begin(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids, const double* poses, Normalisation& N) {
  const size_t n_src_clouds = c->sources.clouds.size();
  if (c->sources.sum_xyz.size() != 3 * n_src_clouds || c->targets.sum_xyz.size() != 3 * c->targets.clouds.size())
    return fail(c, MI355_ICP_ERR_INTERNAL, "upstream normalisation: cloud sums missing");
  // model sums and sizes over ALL ranks (slab-sharded: every rank holds a slab of each model)
  std::vector<double> msum(c->sources.sum_xyz);
  std::vector<uint64_t> mrows(n_src_clouds);
  for (size_t k = 0; k < n_src_clouds; ++k) mrows[k] = c->sources.clouds[k].n;
  if (c->comm.attached()) {
    std::vector<double> mine(4 * n_src_clouds), all(4 * n_src_clouds * static_cast<size_t>(c->comm.size));
    for (size_t k = 0; k < n_src_clouds; ++k) {
      for (int a = 0; a < 3; ++a) mine[4 * k + a] = c->sources.sum_xyz[3 * k + a];
      mine[4 * k + 3] = static_cast<double>(c->sources.clouds[k].n);
    }
    const char* why = nullptr;
    const int cst = comm_allgather_host(c->comm, mine.data(), all.data(), mine.size() * sizeof(double), c->stream, &why);
    if (cst != MI355_ICP_OK) return fail(c, cst, std::string("upstream normalisation: exchanging the model sums: ") + (why ? why : "?"));
    for (size_t k = 0; k < n_src_clouds; ++k) {
      double acc[4] = {0, 0, 0, 0};
      for (int r = 0; r < c->comm.size; ++r)  // rank order: identical bits on every rank
        for (int a = 0; a < 4; ++a) acc[a] += all[static_cast<size_t>(r) * 4 * n_src_clouds + 4 * k + a];
      for (int a = 0; a < 3; ++a) msum[3 * k + a] = acc[a];
      mrows[k] = static_cast<uint64_t>(acc[3]);
    }
  }
  N.centre.assign(3 * static_cast<size_t>(n_jobs), 0.f);
  N.scale.assign(static_cast<size_t>(n_jobs), 0.0);
  N.n_model.assign(static_cast<size_t>(n_jobs), 0);
  const size_t per_job = 2 * static_cast<size_t>(kCloudSumBlocks);
  if (per_job * static_cast<size_t>(n_jobs) > c->norm_cap) {
    if (c->norm_dev) { HIP_TRY(c, hipFree(c->norm_dev)); c->norm_dev = nullptr; }
    if (c->norm_host) { HIP_TRY(c, hipHostFree(c->norm_host)); c->norm_host = nullptr; }
    c->norm_cap = 0;
    const size_t want = per_job * std::max<size_t>(static_cast<size_t>(n_jobs), 8) * 2;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->norm_dev), want * sizeof(double)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->norm_host), want * sizeof(double), hipHostMallocDefault));
    c->norm_cap = want;
  }
  if (!c->norm_event) HIP_TRY(c, hipEventCreateWithFlags(&c->norm_event, hipEventDisableTiming));
  const DevicePlanes S = c->sources.view(), T = c->targets.view();
  for (int first = 0; first < n_jobs; first += 8) {
    NormArgs A{};
    A.n = static_cast<uint32_t>(std::min(8, n_jobs - first));
    A.first = static_cast<uint32_t>(first);
    for (uint32_t q = 0; q < A.n; ++q) {
      const int k = first + static_cast<int>(q);
      const int s = source_ids ? source_ids[k] : 0, t = target_ids ? target_ids[k] : 0;
      const double* P = poses + 16 * static_cast<size_t>(k);
      const CloudEntry& SE = c->sources.clouds[static_cast<size_t>(s)];
      const CloudEntry& TE = c->targets.clouds[static_cast<size_t>(t)];
      N.n_model[static_cast<size_t>(k)] = mrows[static_cast<size_t>(s)];
      if (mrows[static_cast<size_t>(s)] == 0 || TE.n == 0) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "upstream normalisation: empty cloud");
      NormJob& J = A.jobs[q];
      for (int a = 0; a < 3; ++a) {
        const double ms[3] = {msum[3 * s] / static_cast<double>(mrows[s]), msum[3 * s + 1] / static_cast<double>(mrows[s]), msum[3 * s + 2] / static_cast<double>(mrows[s])};
        const double m = ((P[4 * a] * ms[0] + P[4 * a + 1] * ms[1]) + P[4 * a + 2] * ms[2]) + P[4 * a + 3];
        const double mt = c->targets.sum_xyz[3 * static_cast<size_t>(t) + a] / static_cast<double>(TE.n);
        N.centre[3 * static_cast<size_t>(k) + a] = static_cast<float>(0.5 * (m + mt));
        J.centre[a] = N.centre[3 * static_cast<size_t>(k) + a];
      }
      if (!pose_to_f32(P, J.pose)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "upstream normalisation: non-finite pose");
      J.src_off = SE.off; J.src_n = SE.n; J.tgt_off = TE.off; J.tgt_n = TE.n;
    }
    launch_centre_dists(c->stream, A, S, T, c->norm_dev);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->norm_host, c->norm_dev, per_job * static_cast<size_t>(n_jobs) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->norm_event, c->stream));
  N.pending = true;
  return MI355_ICP_OK;
}

// scale = n / ((sum |model'_i - c| + sum |scene_j - c|) / 2); the chunk sums are added in chunk order, the ranks' in rank order
int normalisation_finish(mi355_icp_ctx* c, int n_jobs, Normalisation& N) {
  if (!N.pending) return MI355_ICP_OK;
  HIP_TRY(c, hipEventSynchronize(c->norm_event));
  N.pending = false;
  std::vector<double> ds(static_cast<size_t>(n_jobs), 0.0), dt(static_cast<size_t>(n_jobs), 0.0);
  for (int k = 0; k < n_jobs; ++k)
    for (int b = 0; b < kCloudSumBlocks; ++b) {
      ds[k] += c->norm_host[(static_cast<size_t>(k) * 2) * kCloudSumBlocks + b];
      dt[k] += c->norm_host[(static_cast<size_t>(k) * 2 + 1) * kCloudSumBlocks + b];
    }
  if (c->comm.attached()) {  // the model is sharded, the scene replicated
    std::vector<double> all(static_cast<size_t>(n_jobs) * static_cast<size_t>(c->comm.size));
    const char* why = nullptr;
    const int cst = comm_allgather_host(c->comm, ds.data(), all.data(), ds.size() * sizeof(double), c->stream, &why);
    if (cst != MI355_ICP_OK) return fail(c, cst, std::string("upstream normalisation: exchanging the distance sums: ") + (why ? why : "?"));
    for (int k = 0; k < n_jobs; ++k) {
      double acc = 0.0;
      for (int r = 0; r < c->comm.size; ++r) acc += all[static_cast<size_t>(r) * n_jobs + k];
      ds[k] = acc;
    }
  }
  for (int k = 0; k < n_jobs; ++k) N.scale[k] = static_cast<double>(N.n_model[k]) / ((ds[k] + dt[k]) * 0.5);
  return MI355_ICP_OK;
}

}  // namespace

extern "C" {

const char* mi355_icp_last_error(const mi355_icp_ctx* ctx) {
  if (ctx && ctx->multi && ctx->err.empty()) return multi_last_error(ctx->multi);
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int mi355_icp_create_multi(const int* device_ids, int n_devices, mi355_icp_ctx** out) {
  if (!out) return MI355_ICP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) {
    g_create_error = "create_multi: 1 <= n_devices <= 64 device ids required";
    return MI355_ICP_ERR_INVALID_ARGUMENT;
  }
  mi355_icp_ctx* c = new (std::nothrow) mi355_icp_ctx();
  if (!c) return MI355_ICP_ERR_INTERNAL;
  c->device = device_ids[0];
  std::string why;
  const int st = multi_create(device_ids, n_devices, &c->multi, &why);
  if (st != MI355_ICP_OK) {
    g_create_error = "create_multi: " + why;
    delete c;
    return st;
  }
  *out = c;
  return MI355_ICP_OK;
}

int mi355_icp_create(int device_id, mi355_icp_ctx** out) {
  if (!out) return MI355_ICP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1) {
    g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                     " (this engine has no CPU fallback)";
    return MI355_ICP_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= count) {
    g_create_error = "device_id out of range";
    return MI355_ICP_ERR_INVALID_ARGUMENT;
  }
  mi355_icp_ctx* c = new (std::nothrow) mi355_icp_ctx();
  if (!c) return MI355_ICP_ERR_INTERNAL;
  c->device = device_id;
  {
    const uint32_t ppl = env_u32("MI355_ICP_SCAN_PPL", 0);
    if (ppl == 2 || ppl == 4 || ppl == 8) c->scan_ppl = static_cast<int>(ppl);
  }
  {
    const char* mode = std::getenv("MI355_ICP_SCAN_MODE");
    c->scan_sgpr = mode && std::string(mode) == "sgpr";
    if (mode && std::string(mode) == "exact") c->scan_fused = 0;
    if (mode && std::string(mode) == "fused") c->scan_fused = 1;
    if (mode && std::string(mode) == "xfilter") c->scan_fused = 2;
    if (mode && std::string(mode) == "xmfma") c->scan_fused = 3;   // expansion filter on the matrix cores only
    if (mode && std::string(mode) == "xhybrid") c->scan_fused = 4; // VALU and MFMA workgroups side by side
    if (mode && std::string(mode) == "xbf16") c->scan_fused = 5;   // expansion filter on the matrix cores, split-bf16 operands
    if (mode && std::string(mode) == "pruned") c->scan_fused = 6;  // exact scan with tile/run pruning (not brute force)
    c->mfma_share_pct = env_u32("MI355_ICP_MFMA_SHARE", c->mfma_share_pct);
    if (c->scan_sgpr) c->scan_fused = 0;
  }
  {
    const char* km = std::getenv("MI355_ICP_KNN_MODE");
    c->knn_brute = km && std::string(km) == "brute";
  }
  c->xbf16_shape = static_cast<int>(env_u32("MI355_ICP_XBF16_SHAPE", static_cast<uint32_t>(c->xbf16_shape)));
  c->target_blocks = env_u32("MI355_ICP_TARGET_BLOCKS", c->target_blocks);
  c->target_blocks_mfma = env_u32("MI355_ICP_TARGET_BLOCKS", c->target_blocks_mfma);  // the same override serves A/B runs of either
  c->target_blocks_mfma = env_u32("MI355_ICP_TARGET_BLOCKS_MFMA", c->target_blocks_mfma);  // ... or of the matrix-core scan alone
  c->target_blocks_mfma_forced = std::getenv("MI355_ICP_TARGET_BLOCKS") || std::getenv("MI355_ICP_TARGET_BLOCKS_MFMA");  // then for every size
  c->min_tiles_per_split = env_u32("MI355_ICP_MIN_TILES_PER_SPLIT", c->min_tiles_per_split);
  if (const char* v = std::getenv("MI355_ICP_SPLIT_MODEL")) c->split_model = std::atoi(v) != 0;
  if (const char* v = std::getenv("MI355_ICP_SPLIT_MODEL_PAIRS")) c->split_model_pairs = std::atof(v);
  c->plan_debug = env_u32("MI355_ICP_PLAN_DEBUG", 0) != 0;
  if (const char* v = std::getenv("MI355_ICP_CLASS_SPLIT")) c->class_split = std::atoi(v) != 0;
  if (const char* v = std::getenv("MI355_ICP_SCAN_QUEUE")) c->scan_queue = std::atoi(v) != 0;
  if (const char* v = std::getenv("MI355_ICP_QUEUE_PAIRS")) c->queue_pairs = std::atof(v);
  c->queue_members = env_u32("MI355_ICP_QUEUE_MEMBERS", 0);
  if (const char* v = std::getenv("MI355_ICP_FILTER_PAIRS")) c->filter_pairs = std::atof(v);
  {
    const uint32_t wp = env_u32("MI355_ICP_RESOLVE_WAVE_POINTS", 0);
    if (wp == 16 || wp == 32 || wp == 64) c->resolve_wave_points = static_cast<int>(wp);
  }
  c->max_sub_split = std::min<uint32_t>(32u, std::max<uint32_t>(1u, env_u32("MI355_ICP_MAX_SUB_SPLIT", c->max_sub_split)));
  while (c->max_sub_split & (c->max_sub_split - 1)) c->max_sub_split &= c->max_sub_split - 1;  // a power of two
  c->sub_split_wgs = std::max<uint32_t>(1u, env_u32("MI355_ICP_SUB_SPLIT_WGS", c->sub_split_wgs));
  {
    const char* ss = std::getenv("MI355_ICP_SORT_SOURCES");
    if (ss && std::string(ss) == "0") c->sort_sources = false;
  }
  auto bail = [&](const char* what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    mi355_icp_destroy(c);
    return MI355_ICP_ERR_HIP;
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
  {
    int cus = 0;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id)) != hipSuccess) return bail("hipDeviceGetAttribute", e);
    if (cus > 0) c->scan_slots = std::max<uint32_t>(1u, env_u32("MI355_ICP_SCAN_SLOTS", 2u * static_cast<uint32_t>(cus)));
  }
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  for (auto& ev : c->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->error_flag), sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(c->error_flag, 0, sizeof(uint32_t))) != hipSuccess) return bail("hipMemset", e);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->flag_host), sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->ready_host), (kReadyWords + 1) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->ready_mapped), c->ready_host, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
  std::memset(c->ready_host, 0, (kReadyWords + 1) * sizeof(uint32_t));  // [kReadyWords]: sync_pass's word
  if (const char* v = std::getenv("MI355_ICP_PASS_TAIL")) c->pass_tail = std::atoi(v) != 0;
  if (const char* v = std::getenv("MI355_ICP_PICKY_OFFERS")) c->picky_offers = std::atoi(v) != 0;
  if (const char* v = std::getenv("MI355_ICP_SERVICE_PRUNE")) c->service_prune = std::string(v) == "auto" ? -1 : (std::atoi(v) != 0 ? 1 : 0);
  if (const char* v = std::getenv("MI355_ICP_LEVEL_SORT")) c->level_sort = std::atoi(v) != 0;
  *out = c;
  return MI355_ICP_OK;
}

void mi355_icp_destroy(mi355_icp_ctx* c) {
  if (!c) return;
  if (c->multi) {  // a multi-device handle owns nothing but its per-device contexts
    multi_destroy(c->multi);
    delete c;
    return;
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  comm_release(c->comm);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  for (CloudSet* set : {&c->targets, &c->sources}) release_cloud_buffers(c, *set);
  (void)hipFree(c->scene[0]);
  (void)hipFree(c->scene[1]);
  (void)hipFree(c->scene_planes);
  (void)hipFree(c->upload_raw);
  (void)hipFree(c->upload_flags);
  (void)hipFree(c->cloud_sum_dev);
  (void)hipFree(c->norm_dev);
  if (c->norm_host) (void)hipHostFree(c->norm_host);
  if (c->norm_event) (void)hipEventDestroy(c->norm_event);
  free_xcopy(c->pool, c->scene_xt);
  c->pool.release_all();  // the stream was drained above
  if (c->xt_meta_host) (void)hipHostFree(c->xt_meta_host);
  if (c->xt_meta_event) (void)hipEventDestroy(c->xt_meta_event);
  (void)hipFree(c->scene_normal_planes);
  (void)hipFree(c->scene_normals_aos);
  (void)hipFree(c->knn_d);
  (void)hipFree(c->knn_i);
  (void)hipFree(c->knn_out);
  (void)hipFree(c->scene_state);
  (void)hipFree(c->plane_hyp);  // the counters live behind the hypotheses in the same block
  (void)hipFree(c->plane_partials);
  (void)hipHostFree(c->plane_host);
  (void)hipFree(c->scene_counters);
  if (c->scene_counters_host) (void)hipHostFree(c->scene_counters_host);
  (void)hipFree(c->xt_scratch);
  (void)hipFree(c->xt_tile_box);
  (void)hipFree(c->xt_box);
  (void)hipFree(c->xt_temp);
  (void)hipFree(c->keys3);
  (void)hipFree(c->xb_queue);
  (void)hipFree(c->overflow_list);
  (void)hipFree(c->overflow_count);
  if (c->overflow_count_host) (void)hipHostFree(c->overflow_count_host);
  (void)hipFree(c->fb_planes);
  (void)hipFree(c->fb_index);
  (void)hipFree(c->fb_dist2);
  (void)hipFree(c->jobs2_dev);
  if (c->jobs2_host) (void)hipHostFree(c->jobs2_host);
  (void)hipFree(c->owner);
  (void)hipFree(c->offer_block);
  (void)hipFree(c->selected);
  (void)hipFree(c->ref_partials);
  (void)hipFree(c->ref_sums_dev);
  (void)hipFree(c->thr_dev);
  (void)hipFree(c->select_work);
  (void)hipFree(c->rjobs_dev);
  if (c->rjobs_host) (void)hipHostFree(c->rjobs_host);
  if (c->ref_sums_host) (void)hipHostFree(c->ref_sums_host);
  if (c->pass_done) (void)hipFree(c->pass_done);
  if (c->p2p_done) (void)hipFree(c->p2p_done);
  if (c->ready_host) (void)hipHostFree(c->ready_host);
  if (c->stage_host) (void)hipHostFree(c->stage_host);
  if (c->thr_host) (void)hipHostFree(c->thr_host);
  (void)hipFree(c->keys);
  (void)hipFree(c->out_index);
  (void)hipFree(c->out_dist2);
  (void)hipFree(c->partials);
  (void)hipFree(c->sums_dev);
  (void)hipFree(c->jobs_dev);
  (void)hipFree(c->error_flag);
  if (c->jobs_host) (void)hipHostFree(c->jobs_host);
  if (c->sums_host) (void)hipHostFree(c->sums_host);
  if (c->flag_host) (void)hipHostFree(c->flag_host);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mi355_icp_set_targets(mi355_icp_ctx* ctx, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes) {
  if (!ctx) return MI355_ICP_ERR_INVALID_ARGUMENT;
  ctx->err.clear();
  if (ctx->multi) return multi_set_targets(ctx->multi, n_clouds, xyz, n_points, stride_bytes);
  return upload_clouds(ctx, ctx->targets, n_clouds, xyz, n_points, stride_bytes, kTile, kTargetPad, false, "set_targets");
}

int mi355_icp_set_sources(mi355_icp_ctx* ctx, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes) {
  if (!ctx) return MI355_ICP_ERR_INVALID_ARGUMENT;
  ctx->err.clear();
  if (ctx->multi) return multi_set_sources(ctx->multi, n_clouds, xyz, n_points, stride_bytes);
  return upload_clouds(ctx, ctx->sources, n_clouds, xyz, n_points, stride_bytes, kSrcPad, 0.0f, true, "set_sources");
}

int mi355_icp_align(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids, const double* init_poses,
                    const mi355_icp_params* params, double* out_poses, double* out_residual, int* out_iterations,
                    int* out_status, int* out_converged) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return multi_align(c->multi, n_jobs, source_ids, target_ids, init_poses, params, out_poses, out_residual, out_iterations, out_status, out_converged);
  if (n_jobs < 1 || !init_poses || !params) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align: bad arguments");
  if (params->max_iterations < 1) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align: max_iterations must be >= 1");
  HIP_TRY(c, hipSetDevice(c->device));
  Plan plan;
  c->call_pruned = (params->flags & MI355_ICP_FLAG_PRUNED_SCAN) != 0;
  int st = make_plan(c, n_jobs, source_ids, target_ids, params->max_correspondence_distance, plan);
  if (st != MI355_ICP_OK) return st;
  if ((st = ensure_buffers(c, plan, static_cast<size_t>(n_jobs))) != MI355_ICP_OK) return st;
  if ((plan.xfilter || plan.pruned) && (st = ensure_xtarget(c, c->targets)) != MI355_ICP_OK) return st;
  if ((plan.pruned || plan.sorted_src) && (st = ensure_xtarget(c, c->sources)) != MI355_ICP_OK) return st;
  if (const int est = reset_error_flag(c); est != MI355_ICP_OK) return est;
  c->error_word_known = false;

  struct JobState { double pose[16]; double prev_mse = 0.0; double residual = 0.0; int iters = 0; int status = MI355_ICP_OK; int converged = 0; };
  std::vector<JobState> js(static_cast<size_t>(n_jobs));
  std::vector<int> active;
  for (int k = 0; k < n_jobs; ++k) {
    std::memcpy(js[k].pose, init_poses + 16 * static_cast<size_t>(k), sizeof(js[k].pose));
    float tmp[12];
    if (!pose_to_f32(js[k].pose, tmp)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "align: non-finite initial pose");
    active.push_back(k);
  }
  const bool profile = c->profile_always || (params->flags & MI355_ICP_FLAG_PROFILE) != 0;

  for (int it = 0; it < params->max_iterations && !active.empty(); ++it) {
    for (size_t a = 0; a < active.size(); ++a) {
      JobDesc J = plan.jobs[static_cast<size_t>(active[a])];
      if (!pose_to_f32(js[active[a]].pose, J.pose)) return fail(c, MI355_ICP_ERR_NUMERIC, "align: pose overflowed float32");
      c->jobs_host[a] = J;
    }
    if ((st = run_pass(c, plan, static_cast<int>(active.size()), profile, false, true)) != MI355_ICP_OK) return st;
    std::vector<int> next;
    for (size_t a = 0; a < active.size(); ++a) {
      JobState& s = js[active[a]];
      int done = 0, conv = 0;
      const int hs = host_update(params, c->sums_host + a * MI355_ICP_NSUMS, it, s.pose, &s.prev_mse, &done, &conv);
      if (hs != MI355_ICP_OK) {
        s.status = hs;  // pose left at the last valid value: candidate stays unrefined
        continue;
      }
      s.iters = it + 1;
      s.residual = std::sqrt(s.prev_mse);
      s.converged = conv;
      if (!done) next.push_back(active[a]);
    }
    active.swap(next);
  }
  if ((st = check_error_flag(c)) != MI355_ICP_OK) return st;
  for (int k = 0; k < n_jobs; ++k) {
    if (out_poses) std::memcpy(out_poses + 16 * static_cast<size_t>(k), js[k].pose, sizeof(js[k].pose));
    if (out_residual) out_residual[k] = js[k].residual;
    if (out_iterations) out_iterations[k] = js[k].iters;
    if (out_status) out_status[k] = js[k].status;
    if (out_converged) out_converged[k] = js[k].converged;
  }
  return MI355_ICP_OK;
}

int mi355_icp_correspond(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids, const double* poses,
                         float max_correspondence_distance, int32_t* out_index, float* out_dist2, double* out_sums) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return multi_correspond(c->multi, n_jobs, source_ids, target_ids, poses, max_correspondence_distance, out_index, out_dist2, out_sums);
  c->call_pruned = false;
  if (n_jobs < 1 || !poses) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "correspond: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  Plan plan;
  int st = make_plan(c, n_jobs, source_ids, target_ids, max_correspondence_distance, plan);
  if (st != MI355_ICP_OK) return st;
  if ((st = ensure_buffers(c, plan, static_cast<size_t>(n_jobs))) != MI355_ICP_OK) return st;
  if ((plan.xfilter || plan.pruned) && (st = ensure_xtarget(c, c->targets)) != MI355_ICP_OK) return st;
  if ((plan.pruned || plan.sorted_src) && (st = ensure_xtarget(c, c->sources)) != MI355_ICP_OK) return st;
  if (const int est = reset_error_flag(c); est != MI355_ICP_OK) return est;
  c->error_word_known = false;
  for (int k = 0; k < n_jobs; ++k) {
    JobDesc J = plan.jobs[static_cast<size_t>(k)];
    if (!pose_to_f32(poses + 16 * static_cast<size_t>(k), J.pose)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "correspond: non-finite pose");
    c->jobs_host[k] = J;
  }
  if ((st = run_pass(c, plan, n_jobs, c->profile_always, true)) != MI355_ICP_OK) return st;
  if (out_sums) std::memcpy(out_sums, c->sums_host, static_cast<size_t>(n_jobs) * MI355_ICP_NSUMS * sizeof(double));
  size_t cursor = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const JobDesc& J = plan.jobs[static_cast<size_t>(k)];
    if (J.src_n) {
      int cst = MI355_ICP_OK;
      if (out_index && (cst = copy_to_user(c, out_index + cursor, c->out_index + J.out_off, J.src_n * sizeof(int32_t))) != MI355_ICP_OK) return cst;
      if (out_dist2 && (cst = copy_to_user(c, out_dist2 + cursor, c->out_dist2 + J.out_off, J.src_n * sizeof(float))) != MI355_ICP_OK) return cst;
    }
    cursor += J.src_n;
  }
  return check_error_flag(c);
}

int mi355_icp_debug_filter_bounds(mi355_icp_ctx* c, int source_id, int target_id, const double* pose, float* out_lower,
                                  int32_t* out_run_targets, size_t* inout_n_runs) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) {
    mi355_icp_ctx* one = nullptr;
    const int mst = multi_single_context(c->multi, &one);
    return mst != MI355_ICP_OK ? mst : mi355_icp_debug_filter_bounds(one, source_id, target_id, pose, out_lower, out_run_targets, inout_n_runs);
  }
  c->call_pruned = false;
  if (!pose || !inout_n_runs) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "debug_filter_bounds: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  Plan plan;
  int st = make_plan(c, 1, &source_id, &target_id, 0.0f, plan);
  if (st != MI355_ICP_OK) return st;
  JobDesc J = plan.jobs[0];
  const CloudEntry& T = c->targets.clouds[static_cast<size_t>(target_id)];
  const CloudEntry& S = c->sources.clouds[static_cast<size_t>(source_id)];
  const size_t n_runs = T.padded / kXbRun;
  const size_t capacity = *inout_n_runs;
  *inout_n_runs = n_runs;
  if (!out_lower && !out_run_targets) return MI355_ICP_OK;  // size query
  if (capacity < n_runs) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "debug_filter_bounds: buffers hold fewer runs than the target has");
  if (n_runs * static_cast<size_t>(S.padded) > (1ull << 28)) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "debug_filter_bounds: runs x points too large for a diagnostic dump");
  if (!pose_to_f32(pose, J.pose)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "debug_filter_bounds: non-finite pose");
  // one unsplit job whose output slots start at 0, walked in upload order (the dump is indexed by point position)
  J.n_split = 1;
  J.split_q = J.tgt_tiles;
  J.split_r = 0;
  J.out_off = 0;
  J.scan_blocks = (S.n + 511u) / 512u;
  plan.jobs[0] = J;
  plan.max_split = 1;
  plan.out_total = S.padded;
  plan.fused = plan.xfilter = true;
  plan.xkind = 3;
  plan.pruned = false;
  if ((st = ensure_buffers(c, plan, 1)) != MI355_ICP_OK) return st;
  if ((st = ensure_xtarget(c, c->targets)) != MI355_ICP_OK) return st;
  float* dump = nullptr;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&dump), n_runs * S.padded * sizeof(float)));
  auto finish = [&](int code) { (void)hipStreamSynchronize(c->stream); (void)hipFree(dump); return code; };
  c->jobs_host[0] = J;
  if (hipMemsetAsync(dump, 0, n_runs * S.padded * sizeof(float), c->stream) != hipSuccess ||
      hipMemcpyAsync(c->jobs_dev, c->jobs_host, sizeof(JobDesc), hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return finish(fail(c, MI355_ICP_ERR_HIP, "debug_filter_bounds: staging failed"));
  const uint32_t rank_stride = plan.key_rows() * plan.out_total;
  launch_nn_scan_xbf16_dump(c->stream, c->jobs_dev, J.scan_blocks, c->sources.view(), c->targets.xt.bf, c->targets.xt.bf_table, c->keys3,
                            plan.out_total, rank_stride, dump, S.padded);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
    return finish(fail(c, MI355_ICP_ERR_HIP, "debug_filter_bounds: the dump launch failed"));
  if (out_lower)
    for (size_t g = 0; g < n_runs; ++g)
      if (S.n && hipMemcpy(out_lower + g * S.n, dump + g * S.padded, S.n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return finish(fail(c, MI355_ICP_ERR_HIP, "debug_filter_bounds: copy back failed"));
  if (out_run_targets) {
    std::vector<uint32_t> perm(T.padded);
    if (hipMemcpy(perm.data(), c->targets.xt.perm + T.off, T.padded * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
      return finish(fail(c, MI355_ICP_ERR_HIP, "debug_filter_bounds: copy back failed"));
    for (size_t p = 0; p < T.padded; ++p) out_run_targets[p] = p < T.n ? static_cast<int32_t>(perm[p]) : -1;
  }
  return finish(MI355_ICP_OK);
}

int mi355_icp_upstream_normalisation(mi355_icp_ctx* c, int source_id, int target_id, const double* pose, float* out_centre, double* out_scale) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) {
    mi355_icp_ctx* one = nullptr;
    const int mst = multi_single_context(c->multi, &one);
    return mst != MI355_ICP_OK ? mst : mi355_icp_upstream_normalisation(one, source_id, target_id, pose, out_centre, out_scale);
  }
  if (!pose) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "upstream_normalisation: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->targets.clouds.empty() || c->sources.clouds.empty()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "upstream_normalisation: upload clouds first");
  if (source_id < 0 || static_cast<size_t>(source_id) >= c->sources.clouds.size() || target_id < 0 || static_cast<size_t>(target_id) >= c->targets.clouds.size())
    return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "upstream_normalisation: cloud id out of range");
  Normalisation N;
  int st = normalisation_begin(c, 1, &source_id, &target_id, pose, N);
  if (st == MI355_ICP_OK) st = normalisation_finish(c, 1, N);
  if (st != MI355_ICP_OK) return st;
  if (out_centre) std::memcpy(out_centre, N.centre.data(), 3 * sizeof(float));
  if (out_scale) *out_scale = N.scale[0];
  return MI355_ICP_OK;
}

int mi355_icp_align_reference(mi355_icp_ctx* c, int n_jobs, const int* source_ids, const int* target_ids,
                              const double* init_poses, const mi355_icp_reference_params* rp, double* out_poses,
                              double* out_residual, int* out_passes, int* out_status) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return multi_align_reference(c->multi, n_jobs, source_ids, target_ids, init_poses, rp, out_poses, out_residual, out_passes, out_status);
  if (n_jobs < 1 || !init_poses || !rp) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: bad arguments");
  if (rp->iterations < 1 || rp->num_levels < 1 || rp->num_levels > 24) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: iterations >= 1 and 1 <= num_levels <= 24 required");
  HIP_TRY(c, hipSetDevice(c->device));
  int st = check_reference_inputs(c, n_jobs, source_ids, target_ids);
  if (st != MI355_ICP_OK) return st;
  // MI355_ICP_FLAG_UPSTREAM: what upstream's cv::ppf_match_3d::ICP does where the round 1-3 form left its control flow (model
  // normals in fval, pre-normalisation, cvRound sampling step and pass limit): oracle/icp_oracle.c, "UPSTREAM-FAITHFUL variant"
  const bool upstream = (rp->flags & MI355_ICP_FLAG_UPSTREAM) != 0;
  if (upstream && !c->sources.normals)
    return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT,
                "align_reference: MI355_ICP_FLAG_UPSTREAM needs the model's normals: upload the sources as rows of >= 6 float32 (xyz + normal, "
                "as the reference's models are: opencv_surface_match.cpp:22), or drop the flag for the round 1-3 form");
  // Slab-sharded (a communicator is attached): every rank holds a contiguous slab of each source cloud.  The ranks tell
  // each other their slab sizes once per call; from them every rank knows where its slab begins in the whole cloud,
  // hence which of its points belong to a pyramid level (global indices 0, step, 2 step, ...) and their global level
  // indices (the tie rule of the picky dedupe is "lowest source index": it must be the index in the WHOLE cloud).
  const bool sharded = c->comm.attached();
  std::vector<uint64_t> shard_begin, shard_total;
  std::vector<uint32_t> all;  // sharded: every rank's (cloud count, slab sizes...)
  if (sharded) {

    const size_t n_clouds = c->sources.clouds.size();
    std::vector<uint32_t> mine(n_clouds + 1);
    all.assign((n_clouds + 1) * static_cast<size_t>(c->comm.size), 0u);
    mine[0] = static_cast<uint32_t>(n_clouds);
    for (size_t k = 0; k < n_clouds; ++k) mine[k + 1] = c->sources.clouds[k].n;
    const char* why = nullptr;
    const int cst = comm_allgather_host(c->comm, mine.data(), all.data(), mine.size() * sizeof(uint32_t), c->stream, &why);
    if (cst != MI355_ICP_OK) return fail(c, cst, std::string("align_reference: exchanging the slab sizes: ") + (why ? why : "?"));
    shard_begin.assign(n_clouds, 0);
    shard_total.assign(n_clouds, 0);
    for (int r = 0; r < c->comm.size; ++r) {
      const uint32_t* row = all.data() + static_cast<size_t>(r) * (n_clouds + 1);
      if (row[0] != n_clouds) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: the ranks hold different numbers of source clouds");
      for (size_t k = 0; k < n_clouds; ++k) {
        if (r < c->comm.rank) shard_begin[k] += row[k + 1];
        shard_total[k] += row[k + 1];
      }
    }
    for (size_t k = 0; k < n_clouds; ++k)
      if (shard_total[k] >= (1ull << 32)) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: a sharded source cloud has 2^32 points or more");
  }
  auto sid = [&](int k) { return source_ids ? source_ids[k] : 0; };
  auto tid = [&](int k) { return target_ids ? target_ids[k] : 0; };
  // rows of the whole (all ranks) source cloud `s`, and the step with which pyramid level `level` samples BOTH clouds of a job
  // on that model (0 = the level does not exist for it)
  auto model_rows = [&](int s) -> uint64_t { return sharded ? shard_total[static_cast<size_t>(s)] : c->sources.clouds[static_cast<size_t>(s)].n; };
  auto step_of = [&](int s, int level) -> uint32_t { return upstream ? upstream_step(model_rows(s), level) : (1u << level); };
  if (const int est = reset_error_flag(c); est != MI355_ICP_OK) return est;
  c->error_word_known = false;
  // Slab-sharded: the picky pass tags restart with every call (table cleared at its first pass) - a function of the call's pass
  // sequence only, hence the same on every rank whatever the ranks' engines did before.  One GPU: the tags simply run on across
  // calls (a key of ANY later pass beats the keys of earlier ones), so the 4.8 MB table of a request's six candidates is cleared
  // once per 63 passes - every third or fourth request - instead of at the start of each.  Whoever replaces or grows c->owner
  // clears it (ensure_ref_buffers); whoever resets owner_epoch to 0 gets it cleared by the next pass (run_ref_pass_once).
  if (sharded) c->owner_epoch = 0;
  c->call_pruned = (rp->flags & MI355_ICP_FLAG_PRUNED_SCAN) != 0;
  // no scan flag: the planner takes the exact pruned scan for the passes where it is faster (large frames; never slab-sharded:
  // the ranks' plans stay independent of each other's slab sizes)
  c->call_auto_prune = !sharded && (rp->flags & (MI355_ICP_FLAG_PRUNED_SCAN | MI355_ICP_FLAG_BRUTE_SCAN)) == 0;
  struct AutoPruneOff { mi355_icp_ctx* c; ~AutoPruneOff() { c->call_auto_prune = false; } } auto_prune_off{c};
  const bool profile = c->profile_always || (rp->flags & MI355_ICP_FLAG_PROFILE) != 0;

  std::vector<double> solve_x;
  std::vector<int> solve_st;
  // frame: the pose composed so far in the frame the level loop works in - cloud coordinates starting from the candidate pose
  // (round 1-3 form), or upstream's centred frame starting from the identity (the candidate pose is applied first, as
  // upstream's wrapper does with transformPCPose, and the result is appended to it)
  struct JobState { double init[16]; double frame[16]; double residual = 0.0; int passes = 0; int status = MI355_ICP_OK; };
  struct LevelState { double posex[16]; double start[16]; double fval_old, fval_perc, fval_min; int i; bool done; };
  std::vector<JobState> js(static_cast<size_t>(n_jobs));
  const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int k = 0; k < n_jobs; ++k) {
    std::memcpy(js[k].init, init_poses + 16 * static_cast<size_t>(k), sizeof(js[k].init));
    std::memcpy(js[k].frame, upstream ? I16 : js[k].init, sizeof(js[k].frame));
    float tmp[12];
    if (!pose_to_f32(js[k].init, tmp)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "align_reference: non-finite initial pose");
  }
  Normalisation norm;
  if (upstream && (st = normalisation_begin(c, n_jobs, source_ids, target_ids, init_poses, norm)) != MI355_ICP_OK) return st;
  // the pose in cloud coordinates of a pose X of job k's working frame
  auto to_cloud = [&](int k, const double* X, double* out) {
    if (upstream) from_centred(X, norm.centre.data() + 3 * static_cast<size_t>(k), js[k].init, out);
    else std::memcpy(out, X, 16 * sizeof(double));
  };

  // every subsampled copy the call will need, built in one go (a new scene: all of its levels from one launch)
  auto level_runs = [&](int k, int level, uint32_t* step_out) {
    const uint32_t step = step_of(sid(k), level);
    *step_out = step;
    if (!step) return false;
    return model_rows(sid(k)) / step >= 6 && c->targets.clouds[tid(k)].n / step >= 1;  // decided on the WHOLE cloud: every rank keeps the same jobs
  };
  {
    std::vector<uint32_t> steps;
    for (int level = 0; level < rp->num_levels; ++level)
      for (int k = 0; k < n_jobs; ++k) {
        uint32_t step = 0;
        if (level_runs(k, level, &step) && std::find(steps.begin(), steps.end(), step) == steps.end()) steps.push_back(step);
      }
    if ((st = ensure_levels(c, c->sources, steps, kSrcPad, 0.0f, upstream, sharded ? &shard_begin : nullptr, sharded ? &shard_total : nullptr)) != MI355_ICP_OK) return st;
    if ((st = ensure_levels(c, c->targets, steps, kTile, kTargetPad, true)) != MI355_ICP_OK) return st;
  }

  bool select_cleared = false, overflow_cleared = false;
  for (int level = rp->num_levels - 1; level >= 0; --level) {
    // the jobs of this level by sampling step (one group in the round 1-3 form, and whenever the candidates share a model)
    std::map<uint32_t, std::vector<int>> groups;
    for (int k = 0; k < n_jobs; ++k) {
      if (js[k].status != MI355_ICP_OK) continue;
      uint32_t step = 0;
      if (!level_runs(k, level, &step)) continue;
      if (model_rows(sid(k)) / step >= (1ull << kPickyIndexBits)) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "align_reference: more than 2^27 model points on a level (picky owner keys hold 27 index bits)");
      groups[step].push_back(k);
    }
   for (const auto& group : groups) {
    const uint32_t step = group.first;
    const std::vector<int>& part = group.second;
    std::vector<int> psid, ptid;
    for (const int k : part) { psid.push_back(sid(k)); ptid.push_back(tid(k)); }
    RefLevel lvl;
    lvl.sslot = level_slot(c->sources, step);
    lvl.tslot = level_slot(c->targets, step);
    lvl.upstream = upstream;
    if (lvl.sslot < 0 || lvl.tslot < 0) return fail(c, MI355_ICP_ERR_INTERNAL, "align_reference: a pyramid level is missing");
    if (sharded) {  // the most level points any rank holds over this group's jobs (from the slab geometry: the same on every rank)
      const size_t n_clouds = c->sources.clouds.size();
      for (int r = 0; r < c->comm.size; ++r) {
        uint64_t held = 0;
        for (const int s_id : psid) {
          uint64_t begin = 0;
          for (int q = 0; q < r; ++q) begin += all[static_cast<size_t>(q) * (n_clouds + 1) + 1 + static_cast<size_t>(s_id)];
          uint64_t f = 0, n_loc = 0, base = 0, n_all = 0;
          level_slab(shard_total[static_cast<size_t>(s_id)], begin, all[static_cast<size_t>(r) * (n_clouds + 1) + 1 + static_cast<size_t>(s_id)], step, &f, &n_loc, &base, &n_all);
          held += n_loc;
        }
        lvl.offer_cap = std::max<uint32_t>(lvl.offer_cap, static_cast<uint32_t>(std::min<uint64_t>(held, 0xFFFFFFF0ull)));
      }
    }
    auto level_total = [&](int s) -> uint64_t { return model_rows(s) / step; };  // points of the whole (all ranks) level cloud
    Plan plan;
    const int np = static_cast<int>(part.size());
    if ((st = make_plan_on(c, level_clouds(c->sources, lvl.sslot), level_clouds(c->targets, lvl.tslot), np, psid.data(), ptid.data(), 0.0f, plan)) != MI355_ICP_OK) return st;
    if ((st = ensure_buffers(c, plan, static_cast<size_t>(np), &overflow_cleared)) != MI355_ICP_OK) return st;
    std::vector<uint32_t> owner_off(static_cast<size_t>(np));
    size_t owner_total = 0;
    for (int a = 0; a < np; ++a) {
      owner_off[a] = static_cast<uint32_t>(owner_total);
      owner_total += level_clouds(c->targets, lvl.tslot)[ptid[a]].padded;
    }
    if ((st = ensure_ref_buffers(c, plan, static_cast<size_t>(np), owner_total)) != MI355_ICP_OK) return st;
    // the selection histograms are left clear by every completed pass; make that true whatever an earlier call left behind
    // (sharded passes only: the single-GPU selection of jobs below 64k points keeps its histograms in LDS)
    // (once per call: within a call nothing but completed passes touches them)
    uint32_t largest_job = 0;  // the many-workgroup selection (global histograms) starts at 64k points PER JOB (launch_rejection_select)
    for (const JobDesc& J : plan.jobs) largest_job = std::max(largest_job, J.src_n);
    if ((sharded || largest_job >= 65536u) && !select_cleared) {
      HIP_TRY(c, hipMemsetAsync(c->select_work, 0, c->ref_jobs_cap * (2048 + 4) * sizeof(uint32_t), c->stream));
      select_cleared = true;
    }
    if ((plan.xfilter || plan.pruned) && (st = ensure_level_xcopy(c, c->targets, lvl.tslot)) != MI355_ICP_OK) return st;
    if ((plan.pruned || plan.sorted_src) && (st = ensure_level_xcopy(c, c->sources, lvl.sslot)) != MI355_ICP_OK) return st;

    const double tolp = static_cast<double>(rp->tolerance) * static_cast<double>(level + 1) * static_cast<double>(level + 1);
    const int maxit = upstream ? upstream_max_iterations(rp->iterations, level)
                               : static_cast<int>(std::floor(static_cast<double>(rp->iterations) / static_cast<double>(level + 1) + 0.5));
    std::vector<LevelState> ls(static_cast<size_t>(np));
    for (int a = 0; a < np; ++a) {
      std::memcpy(ls[a].posex, I16, sizeof(I16));
      std::memcpy(ls[a].start, js[part[a]].frame, sizeof(I16));
      ls[a].fval_old = 9999999999.0;
      ls[a].fval_perc = 0.0;
      ls[a].fval_min = 9999999999.0;
      ls[a].i = 0;
      ls[a].done = false;
    }
    for (;;) {
      std::vector<int> active;
      for (int a = 0; a < np; ++a) {
        const LevelState& L = ls[a];
        const bool in_band = L.fval_perc < 1.0 + tolp && L.fval_perc > 1.0 - tolp;
        if (!L.done && !in_band && L.i < maxit) active.push_back(a);
      }
      if (active.empty()) break;
      for (size_t q = 0; q < active.size(); ++q) {
        const int a = active[q];
        JobDesc J = plan.jobs[static_cast<size_t>(a)];
        double cur[16], cur_cloud[16], start_cloud[16];
        mat4_mul(ls[a].posex, ls[a].start, cur);
        to_cloud(part[a], cur, cur_cloud);
        to_cloud(part[a], ls[a].start, start_cloud);
        RefJobDesc R{};
        if (!pose_to_f32(cur_cloud, J.pose) || !pose_to_f32(start_cloud, R.pose_start)) return fail(c, MI355_ICP_ERR_NUMERIC, "align_reference: pose overflowed float32");
        R.owner_off = owner_off[a];
        R.normal_off = c->targets.clouds[ptid[a]].off;
        R.step = step;
        if (upstream) std::memcpy(R.centre, norm.centre.data() + 3 * static_cast<size_t>(part[a]), sizeof(R.centre));
        if (sharded) {  // global level indices and counts (step 1: the slab itself)
          J.src_n_total = static_cast<uint32_t>(level_total(psid[a]));
          R.src_index_base = lvl.sslot ? c->sources.levels[static_cast<size_t>(lvl.sslot) - 1].index_base[static_cast<size_t>(psid[a])]
                                       : static_cast<uint32_t>(shard_begin[static_cast<size_t>(psid[a])]);
        }
        c->jobs_host[q] = J;
        c->rjobs_host[q] = R;
      }
      if ((st = run_ref_pass(c, plan, lvl, static_cast<int>(active.size()), rp->rejection_scale, owner_total, profile)) != MI355_ICP_OK) return st;
      // the distance sums behind the candidates' scales were enqueued in front of the call's first pass: there by now
      if (upstream && (st = normalisation_finish(c, n_jobs, norm)) != MI355_ICP_OK) return st;
      // the candidates' 6x6 systems in one SIMD pass (lane for lane the scalar solve's bits): the device idles meanwhile
      solve_x.resize(6 * active.size());
      solve_st.resize(active.size());
      solve_point_to_plane_batch(c->ref_sums_host, static_cast<int>(active.size()), solve_x.data(), solve_st.data(), kRefSums);
      for (size_t q = 0; q < active.size(); ++q) {
        const int a = active[q];
        LevelState& L = ls[a];
        const double* sums = c->ref_sums_host + q * kRefSums;
        js[part[a]].passes += 1;
        const double* x = solve_x.data() + 6 * q;
        if (sums[28] < 6.0 || solve_st[q] != MI355_ICP_OK) {  // upstream: fewer than 6 pairs or a NaN solve leaves the level
          L.done = true;
          continue;
        }
        euler_pose(x, L.posex);
        // all ranks' points of the level.  Upstream: the norm over all six columns of the matched rows, positions in its
        // normalised units (x scale) beside the difference of the normals
        const double scale = upstream ? norm.scale[static_cast<size_t>(part[a])] : 1.0;
        const double sq = upstream ? scale * scale * sums[27] + sums[29] : sums[27];
        const double fval = std::sqrt(sq) / static_cast<double>(level_total(psid[a]));
        L.fval_perc = fval / L.fval_old;
        L.fval_old = fval;
        if (fval < L.fval_min) L.fval_min = fval;
        L.i += 1;
      }
    }
    for (int a = 0; a < np; ++a) {
      JobState& S = js[part[a]];
      mat4_mul(ls[a].posex, ls[a].start, S.frame);
      S.residual = ls[a].fval_min;
    }
   }
  }
  if (upstream && (st = normalisation_finish(c, n_jobs, norm)) != MI355_ICP_OK) return st;  // a call whose every level was skipped
  if ((st = check_error_flag(c)) != MI355_ICP_OK) return st;
  for (int k = 0; k < n_jobs; ++k) {
    if (out_poses) to_cloud(k, js[k].frame, out_poses + 16 * static_cast<size_t>(k));
    if (out_residual) out_residual[k] = js[k].residual;
    if (out_passes) out_passes[k] = js[k].passes;
    if (out_status) out_status[k] = js[k].status;
  }
  return MI355_ICP_OK;
}

int mi355_icp_reference_pass_ex(mi355_icp_ctx* c, int source_id, int target_id, unsigned step, const double* pose_start,
                                const double* pose_cur, float rejection_scale, const float* centre, int32_t* out_index,
                                float* out_dist2, uint8_t* out_selected, float* out_threshold, double* out_sums) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) {
    mi355_icp_ctx* one = nullptr;
    const int mst = multi_single_context(c->multi, &one);
    return mst != MI355_ICP_OK ? mst : mi355_icp_reference_pass_ex(one, source_id, target_id, step, pose_start, pose_cur, rejection_scale, centre, out_index,
                                                                   out_dist2, out_selected, out_threshold, out_sums);
  }
  c->call_pruned = false;
  if (!pose_start || !pose_cur || step < 1) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference_pass: bad arguments");
  if (c->comm.attached()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference_pass: a single-context diagnostic; detach the communicator (mi355_icp_align_reference is the slab-sharded entry point)");
  HIP_TRY(c, hipSetDevice(c->device));
  int st = check_reference_inputs(c, 1, &source_id, &target_id);
  if (st != MI355_ICP_OK) return st;
  if (centre && !c->sources.normals) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference_pass: the upstream-faithful form needs the source uploaded with normals");
  const std::vector<uint32_t> steps{step};
  if ((st = ensure_levels(c, c->sources, steps, kSrcPad, 0.0f, centre != nullptr)) != MI355_ICP_OK) return st;
  if ((st = ensure_levels(c, c->targets, steps, kTile, kTargetPad, true)) != MI355_ICP_OK) return st;
  RefLevel lvl;
  lvl.sslot = level_slot(c->sources, step);
  lvl.tslot = level_slot(c->targets, step);
  lvl.upstream = centre != nullptr;
  Plan plan;
  if ((st = make_plan_on(c, level_clouds(c->sources, lvl.sslot), level_clouds(c->targets, lvl.tslot), 1, &source_id, &target_id, 0.0f, plan)) != MI355_ICP_OK) return st;
  if (plan.jobs[0].tgt_n < 1) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference_pass: the level has no target points");
  if ((st = ensure_buffers(c, plan, 1)) != MI355_ICP_OK) return st;
  const size_t owner_total = level_clouds(c->targets, lvl.tslot)[target_id].padded;
  if ((st = ensure_ref_buffers(c, plan, 1, owner_total)) != MI355_ICP_OK) return st;
  if ((plan.xfilter || plan.pruned) && (st = ensure_level_xcopy(c, c->targets, lvl.tslot)) != MI355_ICP_OK) return st;
  if ((plan.pruned || plan.sorted_src) && (st = ensure_level_xcopy(c, c->sources, lvl.sslot)) != MI355_ICP_OK) return st;
  if (const int est = reset_error_flag(c); est != MI355_ICP_OK) return est;
  c->error_word_known = false;
  c->owner_epoch = 0;
  JobDesc J = plan.jobs[0];
  RefJobDesc R{};
  if (!pose_to_f32(pose_cur, J.pose) || !pose_to_f32(pose_start, R.pose_start)) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "reference_pass: non-finite pose");
  R.owner_off = 0;
  R.normal_off = c->targets.clouds[target_id].off;
  R.step = step;
  if (centre) std::memcpy(R.centre, centre, sizeof(R.centre));
  c->jobs_host[0] = J;
  c->rjobs_host[0] = R;
  c->want_selected = out_selected != nullptr;
  st = run_ref_pass(c, plan, lvl, 1, rejection_scale, owner_total, c->profile_always);
  c->want_selected = false;
  if (st != MI355_ICP_OK) return st;
  if (out_sums) std::memcpy(out_sums, c->ref_sums_host, kRefSums * sizeof(double));
  if (out_threshold) *out_threshold = c->thr_host[0];
  // The pass may have ended itself (PassTail): the host then only saw the published words, and c->stream does not
  // synchronise with the NULL stream the copies below run on.  The kernel's plain stores (selected, index, dist2) are
  // ordered before these copies only by the end of the kernel: wait for it.
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (J.src_n) {
    int cst = MI355_ICP_OK;
    if (out_index && (cst = copy_to_user(c, out_index, c->out_index + J.out_off, J.src_n * sizeof(int32_t))) != MI355_ICP_OK) return cst;
    if (out_dist2 && (cst = copy_to_user(c, out_dist2, c->out_dist2 + J.out_off, J.src_n * sizeof(float))) != MI355_ICP_OK) return cst;
    if (out_selected && (cst = copy_to_user(c, out_selected, c->selected + J.out_off, J.src_n * sizeof(uint8_t))) != MI355_ICP_OK) return cst;
  }
  return check_error_flag(c);
}

int mi355_icp_reference_pass(mi355_icp_ctx* c, int source_id, int target_id, int level, const double* pose_start,
                             const double* pose_cur, float rejection_scale, int32_t* out_index, float* out_dist2,
                             uint8_t* out_selected, float* out_threshold, double* out_sums) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (level < 0 || level > 23) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "reference_pass: bad arguments");
  double sums[kRefSums];
  const int st = mi355_icp_reference_pass_ex(c, source_id, target_id, 1u << level, pose_start, pose_cur, rejection_scale, nullptr, out_index, out_dist2,
                                             out_selected, out_threshold, out_sums ? sums : nullptr);
  if (st == MI355_ICP_OK && out_sums) std::memcpy(out_sums, sums, kRefSolveSums * sizeof(double));
  return st;
}

int mi355_icp_solve_point_to_plane_batch(const double* sums29, int n, double* out_x, int* out_status) {
  if (!sums29 || n < 0 || !out_x || !out_status) return MI355_ICP_ERR_INVALID_ARGUMENT;
  solve_point_to_plane_batch(sums29, n, out_x, out_status, kRefSolveSums);
  return MI355_ICP_OK;
}

int mi355_icp_solve_point_to_plane(const double* sums29, double out_x[6]) {
  if (!sums29 || !out_x) return MI355_ICP_ERR_INVALID_ARGUMENT;
  return solve_point_to_plane(sums29, out_x);
}

// ---- scene preparation ---------------------------------------------------------------------------------------
int mi355_icp_scene_load(mi355_icp_ctx* c, const float* xyz, size_t n_points, size_t stride_bytes) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  // a multi-device handle prepares the scene on every device (each needs the whole target; the kernels are deterministic, so
  // every device ends up with the same scene) and reports device slot 0's outputs
  if (c->multi) return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int) { return mi355_icp_scene_load(s, xyz, n_points, stride_bytes); });
  if ((n_points && !xyz) || stride_bytes < 12 || stride_bytes % 4 != 0) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_load: bad arguments");
  if (n_points > 0xFFFF0000ull) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_load: too many points");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t need = std::max<size_t>(n_points, 1) * 3 + 16;  // + slack: the kernel reads whole float4s
  if (need > c->scene_cap) {
    for (auto& b : c->scene)
      if (b) { HIP_TRY(c, hipFree(b)); b = nullptr; }
    c->scene_cap = 0;
    const size_t want = need + need / 8;
    for (auto& b : c->scene) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&b), want * sizeof(float)));
    c->scene_cap = want;
  }
  if (!c->scene_counters) {
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->scene_counters), 4 * sizeof(uint32_t)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->scene_counters_host), 4 * sizeof(uint32_t), hipHostMallocDefault));
  }
  if (n_points) {
    if (stride_bytes == 12) {
      HIP_TRY(c, hipMemcpy(c->scene[0], xyz, n_points * 12, hipMemcpyHostToDevice));
    } else {
      std::vector<float> packed(n_points * 3);
      const size_t fs = stride_bytes / 4;
      for (size_t i = 0; i < n_points; ++i) {
        packed[3 * i] = xyz[i * fs];
        packed[3 * i + 1] = xyz[i * fs + 1];
        packed[3 * i + 2] = xyz[i * fs + 2];
      }
      HIP_TRY(c, hipMemcpy(c->scene[0], packed.data(), n_points * 12, hipMemcpyHostToDevice));
    }
  }
  c->scene_n = static_cast<uint32_t>(n_points);
  c->scene_cur = 0;
  c->scene_loaded = true;
  c->scene_normals_valid = false;
  c->scene_sum_valid = false;
  return MI355_ICP_OK;
}

int mi355_icp_scene_filter(mi355_icp_ctx* c, int n_filters, const mi355_scene_filter* filters, size_t* out_n_points) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int slot) { return mi355_icp_scene_filter(s, n_filters, filters, slot == 0 ? out_n_points : nullptr); });
  if (!c->scene_loaded) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_filter: no scene loaded (mi355_icp_scene_load)");
  if (n_filters < 0 || n_filters > MI355_SCENE_MAX_FILTERS || (n_filters && !filters)) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_filter: bad filter list");
  SceneFilterSet F{};
  F.n = n_filters;
  for (int k = 0; k < n_filters; ++k) {
    if (filters[k].kind < 0 || filters[k].kind > MI355_SCENE_FILTER_PLANE_BAND_REMOVE) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_filter: unknown filter kind");
    F.f[k] = filters[k];
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t n = c->scene_n;
  const uint32_t n_tiles = (n + kSceneTile - 1) / kSceneTile;
  if (n_tiles) {
    int st;
    // the tiles' look-back words and, behind them, this pass's counters (ticket, kept total, error bits): one clear for both
    if ((st = grow(c, c->scene_state, c->scene_state_cap, static_cast<size_t>(n_tiles) + 2)) != MI355_ICP_OK) return st;
    uint32_t* counters = reinterpret_cast<uint32_t*>(c->scene_state + n_tiles);
    HIP_TRY(c, hipMemsetAsync(c->scene_state, 0, (static_cast<size_t>(n_tiles) + 2) * sizeof(unsigned long long), c->stream));
    if (c->profile_always) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    launch_scene_compact(c->stream, c->scene[c->scene_cur], n, F, c->scene[c->scene_cur ^ 1], c->scene_state, counters, n_tiles);
    if (c->profile_always) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->scene_counters_host, counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->scene_counters_host[2] & 1u) return fail(c, MI355_ICP_ERR_INTERNAL, "scene_filter: look-back timed out");
    if (c->profile_always) {
      float ms = 0.f;
      HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
      c->prof.scene_filter_ms += ms;
      c->prof.scene_filter_launches += 1;
      c->prof.scene_points_in += n;
      c->prof.scene_points_out += c->scene_counters_host[1];
    }
    c->scene_n = c->scene_counters_host[1];
    c->scene_cur ^= 1;
    c->scene_normals_valid = false;
    c->scene_sum_valid = false;
  }
  if (out_n_points) *out_n_points = c->scene_n;
  return MI355_ICP_OK;
}

int mi355_icp_scene_download(mi355_icp_ctx* c, float* out_xyz, size_t capacity_points, size_t* out_n_points) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return mi355_icp_scene_download(multi_sub(c->multi, 0), out_xyz, capacity_points, out_n_points);
  if (!c->scene_loaded) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_download: no scene loaded");
  if (out_n_points) *out_n_points = c->scene_n;
  if (!out_xyz) return MI355_ICP_OK;
  if (capacity_points < c->scene_n) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_download: buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->scene_n)
    if (const int cst = copy_to_user(c, out_xyz, c->scene[c->scene_cur], static_cast<size_t>(c->scene_n) * 12); cst != MI355_ICP_OK) return cst;
  return MI355_ICP_OK;
}

int mi355_icp_scene_normals(mi355_icp_ctx* c, int k, int flip_viewpoint, const float viewpoint[3], float* out_normals,
                            int32_t* out_neighbors) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi)
    return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int slot) {
      return mi355_icp_scene_normals(s, k, flip_viewpoint, viewpoint, slot == 0 ? out_normals : nullptr, slot == 0 ? out_neighbors : nullptr);
    });
  if (!c->scene_loaded || c->scene_n == 0) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_normals: the scene is empty");
  if (k < 1 || k > kMaxNormalNeighbors) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_normals: 1 <= k <= 32 required");
  if (flip_viewpoint && !viewpoint) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_normals: viewpoint missing");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t n = c->scene_n;
  const uint32_t padded = (n + kTile - 1) / kTile * kTile;
  const uint32_t n_tiles = padded / kTile;
  const uint32_t k_eff = std::min<uint32_t>(static_cast<uint32_t>(k), n);
  const uint32_t kk = k <= 8 ? 8 : (k <= 20 ? 20 : 32);  // list length of the kernel instantiation
  int st;
  if ((st = grow(c, c->scene_planes, c->scene_planes_cap, static_cast<size_t>(padded) * 3)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->scene_normal_planes, c->scene_normal_planes_cap, static_cast<size_t>(padded) * 3)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->scene_normals_aos, c->scene_normals_aos_cap, static_cast<size_t>(n) * 3)) != MI355_ICP_OK) return st;
  // split the target range so that the scan fills the chip (same XCD-aware rule as the NN scan)
  const uint32_t qblocks = (n + kBlockThreads - 1) / kBlockThreads;
  uint32_t split = std::max<uint32_t>(1, c->target_blocks / std::max<uint32_t>(qblocks, 1));
  split = std::min(split, std::max<uint32_t>(1, n_tiles / std::max<uint32_t>(1, c->min_tiles_per_split)));
  split = std::min<uint32_t>(split, 16);
  if (split >= 8) split -= split % 8;
  else if (split >= 4) split = 4;
  else if (split >= 2) split = 2;
  const bool knn_brute = c->knn_brute;  // MI355_ICP_KNN_MODE=brute; the pruned scan (knn_wave_kernel) has no size limit of its own
  if (!knn_brute) split = 1;  // the pruned scan keeps one list per point
  const uint32_t q_stride = qblocks * kBlockThreads;
  const size_t list_len = static_cast<size_t>(split) * kk * q_stride;
  if ((st = grow(c, c->knn_d, c->knn_d_cap, list_len)) != MI355_ICP_OK) return st;
  if ((st = grow(c, c->knn_i, c->knn_i_cap, list_len)) != MI355_ICP_OK) return st;
  if (out_neighbors && (st = grow(c, c->knn_out, c->knn_out_cap, static_cast<size_t>(n) * k_eff)) != MI355_ICP_OK) return st;
  HIP_TRY(c, hipMemsetAsync(c->scene_counters, 0, 4 * sizeof(uint32_t), c->stream));
  launch_scene_to_planes(c->stream, c->scene[c->scene_cur], n, c->scene_planes, padded, 0, padded, kTargetPad, c->scene_counters);
  HIP_TRY(c, hipMemcpyAsync(c->scene_counters_host, c->scene_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->scene_counters_host[2] & 2u) return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "scene_normals: the scene holds NaN/Inf (apply MI355_SCENE_FILTER_FINITE first)");
  const DevicePlanes P{c->scene_planes, c->scene_planes + padded, c->scene_planes + 2 * static_cast<size_t>(padded)};
  const float vp[3] = {viewpoint ? viewpoint[0] : 0.f, viewpoint ? viewpoint[1] : 0.f, viewpoint ? viewpoint[2] : 0.f};
  const bool profile = c->profile_always;
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (knn_brute) {
    launch_knn_scan(c->stream, k, P, n, n_tiles, split, c->knn_d, c->knn_i, q_stride);
  } else {
    // Morton-sorted copy of the scene with per-tile bounding spheres (the same builder the ICP targets use), then the
    // pruned exact scan on it (normals_kernels.hip)
    if (c->scene_xt_stride != padded) {  // another size: buffers of the right class come from the pool
      free_xcopy(c->pool, c->scene_xt);
      c->scene_xt_stride = padded;
    }
    c->scene_xt.valid = false;
    const std::vector<CloudEntry> one{CloudEntry{0, n, padded}};
    if ((st = ensure_xcopy(c, P, padded, one, c->scene_xt)) != MI355_ICP_OK) return st;
    const float* xp = c->scene_xt.planes;
    launch_knn_pruned(c->stream, k, DevicePlanes{xp + 4 * static_cast<size_t>(padded), xp + 5 * static_cast<size_t>(padded), xp + 6 * static_cast<size_t>(padded)},
                      c->scene_xt.perm, c->scene_xt.table, c->scene_xt.sub, c->scene_xt.super, n, n_tiles, c->knn_d, c->knn_i, q_stride);
  }
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  launch_normals(c->stream, k, P, n, k_eff, split, c->knn_d, c->knn_i, q_stride, flip_viewpoint ? 1 : 0, vp, c->scene_normals_aos,
                 c->scene_normal_planes, padded, out_neighbors ? c->knn_out : nullptr);
  if (profile) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipGetLastError());
  // the coordinate sums of this scene (upstream's pre-normalisation needs the scene's mean): they ride on the synchronisation
  // below and go along when the scene is committed as the ICP target
  std::vector<double> sum_chunks;
  c->scene_sum_valid = false;
  if ((st = enqueue_cloud_sums(c, P, std::vector<CloudEntry>{CloudEntry{0, n, padded}}, sum_chunks)) != MI355_ICP_OK) return st;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  {
    std::vector<double> sums;
    finish_cloud_sums(sum_chunks, 1, sums);
    std::memcpy(c->scene_sum, sums.data(), sizeof(c->scene_sum));
    c->scene_sum_valid = true;
  }
  if (profile) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->prof.knn_scan_ms += ms;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->prof.normals_ms += ms;
    c->prof.normals_launches += 1;
  }
  c->scene_normals_valid = true;
  c->scene_normals_padded = padded;
  if (out_normals)
    if (const int cst = copy_to_user(c, out_normals, c->scene_normals_aos, static_cast<size_t>(n) * 12); cst != MI355_ICP_OK) return cst;
  if (out_neighbors)
    if (const int cst = copy_to_user(c, out_neighbors, c->knn_out, static_cast<size_t>(n) * k_eff * sizeof(int32_t)); cst != MI355_ICP_OK) return cst;
  return MI355_ICP_OK;
}

int mi355_icp_scene_fit_plane(mi355_icp_ctx* c, const mi355_plane_fit_params* fp, float out_coefficients[4], size_t* out_inliers,
                              int* out_iterations) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) {  // the fit is deterministic: every device finds the same plane; slot 0's is reported
    if (!fp || !out_coefficients) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_fit_plane: bad arguments");
    return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int slot) {
      float scratch[4];
      return mi355_icp_scene_fit_plane(s, fp, slot == 0 ? out_coefficients : scratch, slot == 0 ? out_inliers : nullptr, slot == 0 ? out_iterations : nullptr);
    });
  }
  if (!fp || !out_coefficients) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_fit_plane: bad arguments");
  if (!c->scene_loaded || c->scene_n < 3) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_fit_plane: the scene has fewer than 3 points");
  if (fp->max_iterations < 1 || fp->max_iterations > kPlaneMaxHypotheses || !(fp->distance_threshold >= 0.0f) ||
      !(fp->probability > 0.0 && fp->probability < 1.0))
    return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_fit_plane: 1 <= max_iterations <= 1024, distance_threshold >= 0 and 0 < probability < 1 required");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t n = c->scene_n, n_hyp = static_cast<uint32_t>(fp->max_iterations);
  const uint32_t blocks = (n + 4 * kBlockThreads - 1) / (4 * kBlockThreads);
  // the moment rows (doubles) sit behind the counters in the pinned read-back block: the counters' size is rounded up so that
  // they start on an 8-byte boundary (4100 bytes of counters would put them on a 4-mod-8 address)
  constexpr size_t kHypBytes = (kPlaneMaxHypotheses + 1) * 4 * sizeof(float);
  constexpr size_t kCountBytes = ((kPlaneMaxHypotheses + 1) * sizeof(uint32_t) + 7) & ~static_cast<size_t>(7);
  static_assert((kHypBytes + kCountBytes) % alignof(double) == 0, "the moment rows must be aligned for double");
  if (!c->plane_hyp) {  // hypotheses | counters: one block, one read-back
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->plane_hyp), kHypBytes + kCountBytes));
    c->plane_counts = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(c->plane_hyp) + kHypBytes);
  }
  int st;
  if ((st = grow(c, c->plane_partials, c->plane_partials_cap, static_cast<size_t>(blocks) * kPlaneMoments)) != MI355_ICP_OK) return st;
  const size_t host_need = kHypBytes + kCountBytes + static_cast<size_t>(blocks) * kPlaneMoments * sizeof(double);
  if (host_need > c->plane_host_cap) {
    if (c->plane_host) { HIP_TRY(c, hipHostFree(c->plane_host)); c->plane_host = nullptr; c->plane_host_cap = 0; }
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->plane_host), host_need + host_need / 4, hipHostMallocDefault));
    c->plane_host_cap = host_need + host_need / 4;
  }
  const float* scene = c->scene[c->scene_cur];
  const float* planes = reinterpret_cast<const float*>(c->plane_host);
  const uint32_t* counts = reinterpret_cast<const uint32_t*>(c->plane_host + kHypBytes);
  const double* rows = reinterpret_cast<const double*>(c->plane_host + kHypBytes + kCountBytes);
  launch_plane_hypotheses(c->stream, scene, n, fp->seed, n_hyp, c->plane_hyp, c->plane_counts);  // also clears the counters it will need
  launch_plane_support(c->stream, scene, n, c->plane_hyp, n_hyp, fp->distance_threshold, c->plane_counts);
  HIP_TRY(c, hipGetLastError());
  // only the rows in use come back (100 hypotheses: 2 KB).  The whole block is 20.5 KB, which the runtime hands to a DMA engine
  // instead of a copy kernel - and the first copy an engine sees costs its queue's creation (7-8 ms; which engine a copy gets depends
  // on what is in flight, and a service met a new one in the middle of its SECOND request: profiles/r04t_second_request_stall.txt)
  HIP_TRY(c, hipMemcpyAsync(c->plane_host, c->plane_hyp, static_cast<size_t>(n_hyp) * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->plane_host + kHypBytes, c->plane_counts, static_cast<size_t>(n_hyp) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int iterations = 0;
  const int best = plane_choose(counts, planes, n_hyp, n, fp->probability, &iterations);
  if (out_iterations) *out_iterations = iterations;
  if (best < 0) return fail(c, MI355_ICP_ERR_NUMERIC, "scene_fit_plane: no three points of the scene span a plane");
  float coeff[4] = {planes[4 * best], planes[4 * best + 1], planes[4 * best + 2], planes[4 * best + 3]};
  size_t inliers = counts[static_cast<size_t>(best)];
  if (fp->optimize) {
    launch_plane_moments(c->stream, scene, n, coeff, fp->distance_threshold, c->plane_partials, blocks);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->plane_host + kHypBytes + kCountBytes, c->plane_partials, static_cast<size_t>(blocks) * kPlaneMoments * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double m[kPlaneMoments] = {0};
    for (uint32_t b = 0; b < blocks; ++b)  // fixed order: the same scene and plane always give the same bits
      for (int k = 0; k < kPlaneMoments; ++k) m[k] += rows[static_cast<size_t>(b) * kPlaneMoments + k];
    float refined[4];
    if (plane_refine(m, coeff, refined) == MI355_ICP_OK) {  // a degenerate inlier set keeps the hypothesis (PCL: optimisation fails, model kept)
      std::memcpy(coeff, refined, sizeof(coeff));
      uint32_t* cslot = c->plane_counts + kPlaneMaxHypotheses;  // the spare counter (cleared with the others above)
      launch_plane_support_one(c->stream, scene, n, coeff, fp->distance_threshold, cslot);  // the refitted plane rides in the kernel arguments
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipMemcpyAsync(c->plane_host + kHypBytes + kPlaneMaxHypotheses * sizeof(uint32_t), cslot, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      inliers = counts[kPlaneMaxHypotheses];
    }
  }
  std::memcpy(out_coefficients, coeff, sizeof(coeff));
  if (out_inliers) *out_inliers = inliers;
  return MI355_ICP_OK;
}

int mi355_icp_scene_commit_target(mi355_icp_ctx* c) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  c->err.clear();
  if (c->multi) return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int) { return mi355_icp_scene_commit_target(s); });
  if (!c->scene_loaded || c->scene_n == 0) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "scene_commit_target: the scene is empty");
  HIP_TRY(c, hipSetDevice(c->device));
  CloudSet& set = c->targets;
  const uint32_t n = c->scene_n;
  const uint32_t padded = (n + kTile - 1) / kTile * kTile;
  const size_t plane_bytes = 3 * static_cast<size_t>(padded) * sizeof(float);
  // scene_normals left, for exactly this scene: its padded SoA planes (checked finite), the normal planes and - pruned
  // k-NN - the Morton-sorted copy with the tile tables and the operand image of the matrix-core scan
  const bool fresh = c->scene_normals_valid && c->scene_normals_padded == padded && c->scene_sum_valid;
  std::vector<double> commit_sum_chunks;
  float* planes = nullptr;
  HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&planes), plane_bytes));
  if (fresh) {
    if (hipMemcpyAsync(planes, c->scene_planes, plane_bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
      c->pool.give(planes);
      return fail(c, MI355_ICP_ERR_HIP, "scene_commit_target: copying the scene planes failed");
    }
  } else {
    bool ok = hipMemsetAsync(c->scene_counters, 0, 4 * sizeof(uint32_t), c->stream) == hipSuccess;
    launch_scene_to_planes(c->stream, c->scene[c->scene_cur], n, planes, padded, 0, padded, kTargetPad, c->scene_counters);
    ok = ok && enqueue_cloud_sums(c, DevicePlanes{planes, planes + padded, planes + 2 * static_cast<size_t>(padded)},
                                  std::vector<CloudEntry>{CloudEntry{0, n, padded}}, commit_sum_chunks) == MI355_ICP_OK;
    ok = ok && hipMemcpyAsync(c->scene_counters_host, c->scene_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) {
      c->pool.give(planes);
      return fail(c, MI355_ICP_ERR_HIP, "scene_commit_target: laying the scene out failed");
    }
    if (c->scene_counters_host[2] & 2u) {
      c->pool.give(planes);
      return fail(c, MI355_ICP_ERR_NONFINITE_INPUT, "scene_commit_target: the scene still holds NaN/Inf (apply MI355_SCENE_FILTER_FINITE)");
    }
  }
  release_cloud_buffers(c, set);
  set.planes = planes;
  set.stride = padded;
  set.clouds.assign(1, CloudEntry{0, n, padded});
  if (fresh) set.sum_xyz.assign(c->scene_sum, c->scene_sum + 3);
  else finish_cloud_sums(commit_sum_chunks, 1, set.sum_xyz);
  if (fresh) {  // normals of exactly this scene: they go along
    HIP_TRY(c, c->pool.take(reinterpret_cast<void**>(&set.normals), plane_bytes));
    HIP_TRY(c, hipMemcpyAsync(set.normals, c->scene_normal_planes, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
    // ... and so does the sorted copy (the builder, its input and its layout are the target's own: ensure_xcopy) - the
    // scene is not sorted a second time.  A second commit of the same scene builds its own.
    if (c->scene_xt.valid && c->scene_xt_stride == padded) {
      set.xt = c->scene_xt;
      c->scene_xt = XCopy{};
      c->scene_xt_stride = 0;
    }
  }
  return MI355_ICP_OK;
}

int mi355_icp_comm_unique_id(unsigned char out_id[MI355_ICP_UNIQUE_ID_BYTES]) {
  if (!out_id) return MI355_ICP_ERR_INVALID_ARGUMENT;
  const char* why = nullptr;
  const int st = rccl_unique_id(out_id, &why);
  if (st != MI355_ICP_OK) g_create_error = std::string("ncclGetUniqueId: ") + (why ? why : "?");
  return st;
}

int mi355_icp_comm_attach(mi355_icp_ctx* c, const unsigned char id[MI355_ICP_UNIQUE_ID_BYTES], int rank, int n_ranks) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach: a multi-device handle shards inside the process (mi355_icp_create_multi); attach a communicator to a single-device context");
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach: bad rank / size");
  if (c->comm.attached()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach: a communicator is already attached");
  HIP_TRY(c, hipSetDevice(c->device));
  const char* why = nullptr;
  void* comm = nullptr;
  if (rccl_comm_init(&comm, id, rank, n_ranks, &why) != MI355_ICP_OK) return fail(c, MI355_ICP_ERR_COMM, std::string("ncclCommInitRank: ") + (why ? why : "?"));
  c->comm.rccl = comm;
  c->comm.rank = rank;
  c->comm.size = n_ranks;
  return MI355_ICP_OK;
}

int mi355_icp_comm_attach_external(mi355_icp_ctx* c, int rank, int n_ranks, mi355_icp_allgather_fn allgather, void* user) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach_external: a multi-device handle shards inside the process (mi355_icp_create_multi)");
  if (!allgather || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach_external: bad rank / size / callback");
  if (c->comm.attached()) return fail(c, MI355_ICP_ERR_INVALID_ARGUMENT, "comm_attach_external: a communicator is already attached");
  c->comm.ext = allgather;
  c->comm.ext_user = user;
  c->comm.rank = rank;
  c->comm.size = n_ranks;
  return MI355_ICP_OK;
}

int mi355_icp_comm_detach(mi355_icp_ctx* c) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return MI355_ICP_OK;  // nothing of the caller's is attached
  if (c->comm.attached()) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    comm_release(c->comm);
  }
  return MI355_ICP_OK;
}

int mi355_icp_comm_info(const mi355_icp_ctx* c, int* rank, int* n_ranks, int* kind) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) {  // kind 3: the devices of one handle, sharded inside the process
    if (rank) *rank = 0;
    if (n_ranks) *n_ranks = multi_devices(c->multi);
    if (kind) *kind = 3;
    return MI355_ICP_OK;
  }
  int r = c->comm.rank, n = c->comm.size;
  if (c->comm.rccl) {  // what the communicator itself reports, not what the caller claimed at attach
    const char* why = nullptr;
    if (rccl_comm_query(c->comm.rccl, &r, &n, &why) != MI355_ICP_OK) return MI355_ICP_ERR_COMM;
  }
  if (rank) *rank = r;
  if (n_ranks) *n_ranks = n;
  if (kind) *kind = c->comm.rccl ? 1 : (c->comm.ext ? 2 : 0);
  return MI355_ICP_OK;
}

int mi355_icp_set_profiling(mi355_icp_ctx* c, int enabled) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int) { return mi355_icp_set_profiling(s, enabled); });
  c->profile_always = enabled != 0;
  return MI355_ICP_OK;
}

int mi355_icp_get_profile(const mi355_icp_ctx* c, mi355_icp_profile* out) {
  if (!c || !out) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return mi355_icp_get_profile(multi_sub(c->multi, 0), out);  // device slot 0's kernels
  *out = c->prof;
  return MI355_ICP_OK;
}

int mi355_icp_reset_profile(mi355_icp_ctx* c) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_broadcast(c->multi, [&](mi355_icp_ctx* s, int) { return mi355_icp_reset_profile(s); });
  c->prof = mi355_icp_profile{};
  return MI355_ICP_OK;
}

int mi355_icp_device_info(const mi355_icp_ctx* c, char* name, size_t name_len, int* compute_units, size_t* total_mem) {
  if (!c) return MI355_ICP_ERR_INVALID_ARGUMENT;
  if (c->multi) return mi355_icp_device_info(multi_sub(c->multi, 0), name, name_len, compute_units, total_mem);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return MI355_ICP_ERR_HIP;
  if (name && name_len) {
    std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_mem) *total_mem = prop.totalGlobalMem;
  return MI355_ICP_OK;
}

}  // extern "C"

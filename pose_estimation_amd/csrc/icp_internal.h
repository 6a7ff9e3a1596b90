// icp_internal.h — declarations shared by the engine's translation units
// (not part of the C-ABI; see include/mi355_icp.h for that).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>

#include "../../include/mi355_icp.h"

namespace mi355 {

// ---- geometry of the scan (compile-time; the kernels and the host planner agree on these) ----
constexpr int kBlockThreads = 256;                       // 4 wave64 per workgroup
constexpr int kResolvePointsPerLane = 4;                 // resolve/accumulate: source points per lane
constexpr int kSrcPerBlock = kBlockThreads * kResolvePointsPerLane;  // 1024: resolve workgroup / partial-sum granule
constexpr int kMaxScanPointsPerLane = 8;                 // the scan register-blocks 2, 4 or 8 source points per lane
constexpr int kSrcPad = kBlockThreads * kMaxScanPointsPerLane;       // source clouds are padded to 2048 points
constexpr int kTile = 1024;                              // target points per LDS tile (12 KiB SoA, x2 buffers)
constexpr int kChunk = 32;                               // granule at which the scan remembers "where the min fell"
constexpr int kChunksPerTile = kTile / kChunk;
constexpr float kTargetPad = 3.0e38f;                    // coordinate of padding targets: d2 overflows to +inf
constexpr int kFusedChunk = 64;                          // granule of the fused-filter scan (icp_fused_kernels.hip)
constexpr int kFusedChunksPerTile = kTile / kFusedChunk;
// split-bf16 matrix-core filter (nn_scan_xbf16_kernel): a 256-target chunk is 8 MFMA blocks of 32 rows; each half of a
// wave owns the rows of one contiguous 128-target run, which is the granule the scan remembers and the resolve re-scans
constexpr int kXbChunk = 256;
constexpr int kXbChunksPerTile = kTile / kXbChunk;
constexpr int kXbRun = 128;
constexpr int kXbTileVecs = 2 * kTile;                   // uint4 per tile of the bf16 operand image: [k half][row slot]
// The operand image of kXbCentreGroup consecutive tiles is written around ONE shared centre (and carries one shared r^2),
// so the scan re-splits its source points into bf16 pieces once per group instead of once per tile (36 vector
// instructions per source set each time).  The price is a band E that grows with the group's radius: at 1M<->1M groups
// of 1 / 2 / 4 tiles scan in 28.35 / 27.86 / 27.15 ms and leave 563 / 1210 / 2636 of 10^6 points to the overflow pass
// (0.23 -> 0.25 ms of resolve): profiles/r02q_scan_ab_centre_groups.txt.
#ifndef MI355_XB_CENTRE_GROUP
#define MI355_XB_CENTRE_GROUP 4
#endif
constexpr int kXbCentreGroup = MI355_XB_CENTRE_GROUP;

#if defined(__HIPCC__)
// bfloat16 pieces of a float32 (round to nearest even; NaN stays NaN, +-inf stays +-inf).  x ~ hi + lo with
// |x - (hi + lo)| <= 2^-16 |x|; the filters' bounds are written for whatever (hi, lo) these return.
__device__ __forceinline__ uint32_t bf16_bits(float x) {
  const uint32_t u = __float_as_uint(x);
  const uint32_t r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  return x != x ? 0x7FC0u : r;
}
__device__ __forceinline__ float bf16_value(uint32_t bits) { return __uint_as_float(bits << 16); }

// ---- sphere bounds of the pruned scans (nn_pruned_kernel, knn_wave_kernel) ----
// Lower bound of the contract squared distance from any point of the sphere (o, orad) to any point of the sphere c =
// (centre, r^2), with every
// rounding pushed DOWN: relative margins of 2^-18 against float32 errors of a few 2^-24 (v_sqrt_f32 directly: its 1 ulp
// is far inside the margins, and the correctly rounded sqrtf expands to fused multiply-adds, which `make check-isa` bans
// from icp_kernels.hip).  0 when o may be inside the sphere; NaN when nothing can be said (overflow).
__device__ __forceinline__ float sphere_lower_bound(float ox, float oy, float oz, float orad, float4 c) {
  const float dx = ox - c.x, dy = oy - c.y, dz = oz - c.z;
  float a = dx * dx;
  const float b = dy * dy, e = dz * dz;
  a = a + b;
  a = a + e;
  const float D = __builtin_amdgcn_sqrtf(a) * 0.999996185f;                                            // (1 - 2^-18)
  const float gap = D - (orad + __builtin_amdgcn_sqrtf(c.w)) * 1.000003815f - 1.17549435e-38f;         // (1 + 2^-18), FLT_MIN
  float lb = gap * gap;
  lb = lb * 0.999996185f;
  return gap > 0.0f ? lb : (gap <= 0.0f ? 0.0f : gap);
}
// can the sphere hold a point at squared distance <= dk from some point of the sphere (o, orad)?  (NaN bounds answer yes)
__device__ __forceinline__ bool sphere_reachable(float ox, float oy, float oz, float orad, float4 c, float dk) {
  return !(sphere_lower_bound(ox, oy, oz, orad, c) > dk);
}
constexpr int kSuperTiles = 16;          // tiles per super-tile sphere (the first level of the pruned scans' tile search)
constexpr uint32_t kPrunedMaxTiles = 16384;  // bitmap of needed tiles held in LDS: 16M points per cloud

// Marks in `need` (LDS bitmap, tile index relative to tile_base) every tile of [tile_base, tile_base + n_tiles) that some
// lane of this wave can still reach with its current dk: super-tile spheres first, then the 16 tiles of a reachable one.
__device__ __forceinline__ void mark_reachable_tiles(float qx, float qy, float qz, float dk, const float4* __restrict__ super_table,
                                                     const float4* __restrict__ tile_table, uint32_t tile_base, uint32_t n_tiles,
                                                     uint32_t* need) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s0 = tile_base / kSuperTiles, s1 = (tile_base + n_tiles - 1) / kSuperTiles;
  for (uint32_t sidx = s0; sidx <= s1; ++sidx) {
    if (__builtin_amdgcn_ballot_w64(sphere_reachable(qx, qy, qz, 0.0f, super_table[sidx], dk)) == 0) continue;
    const uint32_t t0 = sidx * kSuperTiles > tile_base ? sidx * kSuperTiles : tile_base;
    const uint32_t t1 = (sidx + 1) * kSuperTiles < tile_base + n_tiles ? (sidx + 1) * kSuperTiles : tile_base + n_tiles;
    for (uint32_t t = t0; t < t1; ++t)
      if (__builtin_amdgcn_ballot_w64(sphere_reachable(qx, qy, qz, 0.0f, tile_table[t], dk)) != 0 && lane == 0)
        atomicOr(&need[(t - tile_base) >> 5], 1u << ((t - tile_base) & 31u));
  }
}
#endif

// One alignment job as the kernels see it (read with scalar loads; 112 bytes).
struct JobDesc {
  uint32_t src_off;          // first point of the source cloud in the source planes (multiple of kSrcPerBlock)
  uint32_t src_n;            // real source points
  uint32_t src_blocks;       // ceil(src_n / kSrcPerBlock): resolve workgroups / partial rows
  uint32_t tgt_off;          // first point of the target cloud in the target planes (multiple of kTile)
  uint32_t tgt_n;            // real target points
  uint32_t tgt_tiles;        // ceil(tgt_n / kTile)
  uint32_t n_split;          // the target tiles are cut into n_split balanced, non-empty runs:
  uint32_t split_q;          //   run sp starts at sp*split_q + min(sp, split_r) and holds split_q + (sp < split_r) tiles

  uint32_t out_off;          // first slot of this job in the per-call outputs (keys, index, dist2)
  uint32_t partial_off;      // first per-block partial-sum slot of this job
  uint32_t split_r;
  float max_d2;              // accept a pair iff d2 <= max_d2 (+inf keeps all)
  uint32_t scan_blocks;      // ceil(src_n / (256 * points_per_lane)): scan workgroups along the source
  float pose[12];            // float32 rounding of the row-major 3x4 [R|t] the pass is evaluated under
  uint32_t sub_split;        // all-exact scan of SMALL launches only: > 1 cuts every tile into this many runs of chunks, one
                             // workgroup each (n_split = tgt_tiles * sub_split; 0 or 1 = whole tiles).  A tile is 1024
                             // targets x 512 source slots: tens of microseconds on one CU whatever the cloud holds, which is
                             // what a pass costs on the coarse pyramid levels of reference mode unless it is cut finer.
  uint32_t src_n_total;      // slab-sharded reference mode: points of the WHOLE level cloud over all ranks (0 = src_n)
  uint32_t first_class;      // matrix-core scan launched with kXbLayoutClassed (a ONE-ROUND launch, self-staged kernel; the layout is
                             // the LAUNCH's, decided by the launcher - this field only parametrises it): the grid is
                             // (scan_blocks, jobs, n_split) and the first `first_class` workgroups in grid order - one per CU,
                             // the older wave on every SIMD - get kClassFast / kClassSlow times the tiles of the later ones
};
constexpr uint32_t kClassFast = 13, kClassSlow = 6;  // tools/xb_timeline.py, round 3: 4.6 vs ~10 us per tile while both are resident
static_assert(sizeof(JobDesc) == 112, "JobDesc layout");

// ---- reference mode (point-to-plane / rejection / picky / pyramid): icp_ref_kernels.hip ----
// How ref_accumulate_kernel ends a single-GPU reference-mode pass by itself (done == nullptr: it does not, and
// ref_final_reduce_kernel plus a device-to-host copy follow).
struct PassTail {
  uint32_t* done = nullptr;            // [n_jobs], zero between passes: workgroups of the job that have finished
  double* sums = nullptr;              // the pass result block in pinned host memory, as the device addresses it
  uint32_t* overflow_count = nullptr;  // matrix-core scan: points sent to the overflow list per job (nullptr otherwise)
  uint32_t overflow_cap = 0xFFFFFFFFu;
  uint32_t seq = 0;                    // ready[job] becomes (seq << 1 | error bit) once the job's rows are written
  uint32_t* ready = nullptr;           // [n_jobs] pinned host words
  const uint32_t* error_flag = nullptr;  // the call's "non-finite distance met" word
};

#if defined(__HIPCC__)
// Sum of NV doubles per thread over a 256-thread workgroup, in a FIXED order: every lane parks its NV values in LDS, one thread
// per (wave, value) adds that wave's 64 lanes up - four chains of sixteen lanes each, lanes ascending, combined as
// ((c0 + c1) + (c2 + c3)) - and thread k < NV adds the four wave results in wave order.  Threads 0..NV-1 return the total of
// value [threadIdx.x].
// Rounds 1-2 used a shuffle tree inside each wave: six dependent exchanges per value, each two ds_bpermute for a double -
// 348 LDS-pipe round trips for NV = 29, ~4 us of every accumulation launch whatever the job's size (phase stamps:
// tools/ref_kernel_phases.py; a coarse reference-mode pass is ~45 us in all).  The parked form is ~0.5 us.
// Rows are 65 doubles apart: the summing threads read different rows at the same lane position - 64-bit reads 520 bytes apart
// fall into different banks.
template <int NV>
__device__ __forceinline__ double pass_block_sum(double (&v)[NV], double (*scratch)[NV]) {
  constexpr int kWaves = kBlockThreads / 64;
  constexpr int kRow = 65;
  static_assert(kWaves * NV <= kBlockThreads, "one thread per (wave, value)");
  // static LDS of a workgroup is capped at 64 KiB: the parked values, the caller's scratch[kWaves][NV] and a few words.
  // NV = 30 (reference mode) is 62,400 + 960 bytes - two workgroups per CU; a 31st sum would need the values parked in
  // two halves (same chains, same bits).  The order of these chains is part of the bit contract of the sums since round 3
  // (rounds 1-2: a shuffle tree) - nothing pins the bits of a sum except run-to-run equality; parity is rtol 1e-10.
  static_assert(sizeof(double) * (kWaves * NV * kRow + kWaves * NV) + 64 <= 64 * 1024, "pass_block_sum: parked values exceed the static LDS limit");
  __shared__ double parked[kWaves * NV * kRow];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) parked[(wave * NV + k) * kRow + lane] = v[k];
  __syncthreads();
  if (threadIdx.x < kWaves * NV) {
    const double* __restrict__ row = parked + threadIdx.x * kRow;  // thread w * NV + k owns wave w's value k
    double c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = row[16 * q];
#pragma unroll
      for (int j = 1; j < 16; ++j) a += row[16 * q + j];
      c[q] = a;
    }
    scratch[threadIdx.x / NV][threadIdx.x % NV] = (c[0] + c[1]) + (c[2] + c[3]);
  }
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < NV) {
    r = scratch[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) r += scratch[w][threadIdx.x];
  }
  return r;
}

// How an accumulation launch (grid = (workgroups of the largest job, jobs), 256 threads) ENDS the pass by itself: the last
// workgroup of a job to finish adds the job's partial rows up - in row order, so the sums are those of the separate
// final-reduce kernels bit for bit - and writes them, with whatever `extras` appends behind the n x NV rows, straight into
// the host's (pinned, mapped) result block; then it publishes (tail.seq << 1 | error bit) in tail.ready[job], which is what
// the host polls.  One launch and no copy command instead of two launches and a device-to-host copy, on a chain of
// dependent launches that is most of a small pass.
// This is the classic last-block reduction WITHOUT its device-wide fences (a __threadfence() is a write-back of the whole L2
// on this chip, per workgroup: +15 us on a 14 us launch).  Everything one workgroup hands to another goes through
// agent-scope atomic accesses instead - they are performed at memory, past the per-XCD L2s - and a writer waits for its
// own stores to be acknowledged (s_waitcnt) before it raises the count the reader takes its cue from.  What is left is a
// chain of memory round trips, so every link that can go does: a job of one workgroup has nobody to wait for, every job
// publishes for itself (no count of finished jobs), and the caller reads the error word (`raised`) at kernel start.
//   r: pass_block_sum's result of this workgroup (threads < NV); row / n_rows: this workgroup's partial row of the job
//   extras(block_behind_the_rows, job, n_jobs): thread 0 only; system-scope atomic stores
//
// HARDWARE ASSUMPTIONS (gfx950; nothing in the C++ / HSA memory model orders relaxed accesses, so this is a statement about
// the ISA the file is compiled to, and the #error below keeps it from being compiled to another one):
//   1. relaxed agent-scope atomic stores / loads of 8 bytes lower to global_store_dwordx2 / global_load_dwordx2 with sc1:
//      performed at the memory side of the per-XCD L2s, never served from a CU's L1 (MI355X_MICROARCH.md, "Workgroup
//      dispatch, XCD placement & inter-workgroup visibility": the {8-B agent atomics both sides} form);
//   2. stores, loads and atomics of a wave are counted on ONE counter (vmcnt) in issue order, so s_waitcnt 0 returns only
//      when every row this wave stored has been acknowledged by memory (gfx10+ count stores on vscnt, which the immediate 0
//      does not wait for: there this would be silently wrong);
//   3. the hand-over is the first row of that guide's table of measured hand-offs: every storing wave waits (s_waitcnt),
//      the workgroup meets at a barrier, ONE lane adds to the job's counter, the workgroup whose add returned n_rows - 1
//      is the reader, its other waves read behind a second barrier, and every read of the handed-over rows is an sc1 load.
// The rows are stored by threads < NV; the barrier between the wait and the add makes that correct for any NV <= the
// workgroup size (the waits of ALL waves precede the add), not only while the rows come from wave 0.
// A/B and escape hatch: MI355_ICP_PASS_TAIL=0 takes the separate reduce launch + copy (tools/gpu_check.sh runs the suite
// that way too; tests/test_gpu_reference_mode.py compares the two bit for bit, 150 repetitions).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "finish_pass relies on gfx950 memory-counter behaviour (see HARDWARE ASSUMPTIONS): build with --offload-arch=gfx950 only"
#endif
template <int NV, typename Extras>
__device__ __forceinline__ void finish_pass(double r, double (&acc)[NV], double (*scratch)[NV], uint32_t* last_flag /* LDS word */,
                                            double* job_rows, uint32_t row, uint32_t n_rows, const PassTail& tail, uint32_t raised,
                                            Extras extras, bool reread_error = false) {
  static_assert(NV <= kBlockThreads, "one thread per sum stores a row: NV must not exceed the workgroup size");
  const uint32_t job = blockIdx.y, n_jobs = gridDim.y;
  double total = r;
  if (n_rows > 1) {
    if (threadIdx.x < NV) __hip_atomic_store(job_rows + static_cast<size_t>(row) * NV + threadIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the row is at memory
    __syncthreads();
    if (threadIdx.x == 0)
      *last_flag = __hip_atomic_fetch_add(&tail.done[job], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_rows ? 1u : 0u;
    __syncthreads();
    if (!*last_flag) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    for (uint32_t b = threadIdx.x; b < n_rows; b += kBlockThreads) {  // row order: the final-reduce kernels' sums, bit for bit
      double* src = job_rows + static_cast<size_t>(b) * NV;
#pragma unroll
      for (int k = 0; k < NV; ++k) acc[k] += __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // scratch is reused
    total = pass_block_sum<NV>(acc, scratch);
  }
  if (threadIdx.x < NV) __hip_atomic_store(tail.sums + static_cast<size_t>(job) * NV + threadIdx.x, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) {
    extras(tail.sums + static_cast<size_t>(n_jobs) * NV, job, n_jobs);
    if (n_rows > 1) __hip_atomic_store(&tail.done[job], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next pass
  }
  __builtin_amdgcn_s_waitcnt(0);  // this job's rows have reached the host's memory ...
  __syncthreads();
  // ... and only then its word.  A launch that can raise the error word ITSELF (reread_error) looks at it again now: every
  // workgroup of this job has waited for its own memory operations before it counted in, so whatever the job raised is there
  // (what another job raises travels with that job's word; the host ORs them)
  if (threadIdx.x == 0) {
    if (reread_error) raised |= __hip_atomic_load(tail.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tail.ready + job, (tail.seq << 1) | (raised ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the overflow bookkeeping both result layouts end with: [n excess flags] ... [n counts] at flag_at / count_at
__device__ __forceinline__ uint32_t publish_overflow(const PassTail& tail, uint32_t job, double* flag_at, double* count_at) {
  const uint32_t n = tail.overflow_count ? __hip_atomic_load(&tail.overflow_count[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  __hip_atomic_store(flag_at, n > tail.overflow_cap ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(count_at, static_cast<double>(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (tail.overflow_count) tail.overflow_count[job] = 0u;
  return n;
}
#endif

constexpr int kRefSolveSums = MI355_ICP_NREFSUMS;  // what the 6x6 solve reads: 21 upper-triangle A^T A + 6 A^T b + sum |p-q|^2 + count
constexpr int kRefSums = MI355_ICP_NREFSUMS_EX;    // a pass's row: those 29 + sum |n_p - n_q|^2 (upstream-faithful variant; 0 otherwise)
static_assert(kRefSolveSums == 29 && kRefSums == 30, "reference-mode row layout");
struct RefJobDesc {
  uint32_t owner_off;    // first slot of this job's target in the picky owner table
  uint32_t normal_off;   // first point of the job's target cloud in the LEVEL-0 normal planes
  uint32_t step;         // level stride: level point j is level-0 point j * step
  uint32_t owner_prefix; // this pass's tag in the owner keys (picky_key): smaller in every later pass
  uint32_t src_index_base;  // index, in the WHOLE level cloud, of this rank's first level point (0 unless slab-sharded)
  float centre[3];       // upstream-faithful variant: (float)meanAvg of upstream's pre-normalisation - the plane equations are
                         // written for p - centre (oracle/icp_oracle.c, "UPSTREAM-FAITHFUL variant"); unused otherwise
  float pose_start[12];  // float32 pose the level started from (defines the point the plane equation is written for)
};
static_assert(sizeof(RefJobDesc) == 80, "RefJobDesc layout");

// Up to 8 job descriptors as KERNEL ARGUMENTS (the packet's kernarg segment): what stage_jobs_kernel writes to the device
// tables, and what the all-exact scan of a small pass reads its own descriptor from directly (n = 0: not staged) while its
// first workgroup writes the tables for the launches behind it - one dependent launch less per pass.
struct StageArgs {
  uint32_t n, has_ref, first, pad_;
  JobDesc jobs[8];
  RefJobDesc rjobs[8];
};


// "Picky" owner key of an accepted pair: atomicMin over these makes the smallest distance win, the lowest source index on
// ties.  The 6-bit prefix is the pass's tag, DEcreasing from pass to pass, so a key of this pass beats whatever an earlier
// pass (or another job or level that used the slot) left behind and the table needs no clearing between passes - only at
// the first pass of a call and when the tag wraps after 63 passes (all-ones = "never written" sorts after every tag).  31 bits of distance (finite
// non-negative float32 bits are < 2^31), 27 bits of source index: 134M points per model.
constexpr uint32_t kPickyIndexBits = 27;
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long picky_key(uint32_t prefix, float d, uint32_t source_index) {
  return (static_cast<unsigned long long>(prefix) << 58) | (static_cast<unsigned long long>(__float_as_uint(d)) << kPickyIndexBits) | source_index;
}
#endif

int solve_point_to_plane(const double* sums29, double* x6);
// n systems at once, 8 per SIMD pass; x: n x 6, status: n (MI355_ICP_OK / MI355_ICP_ERR_NUMERIC).  Lane for lane the bits of solve_point_to_plane.
// System k starts at sums + k * stride (stride >= kRefSolveSums doubles; the engine's own rows are kRefSums wide).
void solve_point_to_plane_batch(const double* sums, int n, double* x6, int* status, size_t stride = kRefSolveSums);
void euler_pose(const double* x6, double* pose16);
void mat4_mul(const double* a, const double* b, double* c);

// ---- scene preparation (scene_kernels.hip) ----
// compaction workgroup (scene_compact_kernel): 12 waves x 16 points per thread = 12288-point tiles, two workgroups per CU at
// <= 80 registers.  One look-back per tile is a fixed ~6-8 us of a tile's life, so larger tiles amortise it - until the
// workgroup no longer fits a CU twice: 256 / 512 / 768 / 1024 threads give 4.0 / 4.6 / 4.8 / 5.0 TB/s on a NaN-removal pass over
// 64M points and 3.8 / 4.5 / 4.6 / 4.0 TB/s with three predicates (profiles/r03l_scene_compact_shapes.txt).  -D overrides: A/B only.
#ifndef MI355_SCENE_THREADS
#define MI355_SCENE_THREADS 768
#endif
#ifndef MI355_SCENE_PER_THREAD
#define MI355_SCENE_PER_THREAD 16
#endif
#ifndef MI355_SCENE_MIN_WAVES
#define MI355_SCENE_MIN_WAVES 6      // waves per SIMD the register allocation must leave room for (2 workgroups of 12 waves per CU)
#endif
constexpr int kSceneThreads = MI355_SCENE_THREADS;
constexpr int kSceneTile = MI355_SCENE_THREADS * MI355_SCENE_PER_THREAD;  // points per compaction workgroup, held in registers
using SceneFilter = mi355_scene_filter;
struct SceneFilterSet {
  int32_t n;
  SceneFilter f[MI355_SCENE_MAX_FILTERS];
};
void launch_scene_compact(void* stream, const float* in, uint32_t n, const SceneFilterSet& filters, float* out,
                          unsigned long long* state, uint32_t* counters, uint32_t n_tiles);
// RANSAC plane fit (scene_kernels.hip): hypotheses (float4 a, b, c, d; d = NaN marks a hypothesis with no valid triple), their
// support counts in one pass, and the 10 moments (n, sum p, sum p p^T upper triangle) of one plane's inliers per 1024-point block
constexpr int kPlaneMaxHypotheses = 1024;
constexpr int kPlaneMoments = 10;
// counts: kPlaneMaxHypotheses + 1 support counters; the launch clears the first n_hyp and the last
void launch_plane_hypotheses(void* stream, const float* scene, uint32_t n, unsigned long long seed, uint32_t n_hyp, float* planes4,
                             uint32_t* counts);
void launch_plane_support(void* stream, const float* scene, uint32_t n, const float* planes4, uint32_t n_hyp, float threshold, uint32_t* counts);
// one plane, passed by value (host memory): its inlier count is ADDED to *count
void launch_plane_support_one(void* stream, const float* scene, uint32_t n, const float* plane4, float threshold, uint32_t* count);
void launch_plane_moments(void* stream, const float* scene, uint32_t n, const float* plane4, float threshold, double* partials, uint32_t n_blocks);
// host side (host_math.cpp): PCL's adaptive choice replayed over the counts; the refined plane from the moments
int plane_choose(const uint32_t* counts, const float* planes4, uint32_t n_hyp, uint32_t n_points, double probability, int* iterations);
int plane_refine(const double* moments10, const float* hypothesis4, float* out4);
void launch_rows_to_planes(void* stream, const float* rows, uint32_t fstride, uint32_t n, float* planes, float* normals, size_t stride,
                           uint32_t off, uint32_t n_padded, float pad, uint32_t* flags, uint32_t cloud);
void launch_scene_to_planes(void* stream, const float* in, uint32_t n, float* planes, size_t stride, uint32_t off,
                            uint32_t n_padded, float pad, uint32_t* counters);

// ---- host math (host_math.cpp) ----
void level_slab(uint64_t n_total, uint64_t begin, uint64_t count, uint64_t step, uint64_t* first_local, uint64_t* n_local, uint64_t* index_base,
                uint64_t* n_level_total);
// upstream's sampling step of pyramid level `level` for a model of n_model rows: cvRound(n / cvRound(n / 2^level)), half to
// even; 0 = the level has no samples.  And its pass limit cvRound(iterations / (level + 1)).
uint32_t upstream_step(uint64_t n_model, int level);
int upstream_max_iterations(int iterations, int level);
int solve_rigid(const double* sums, double* pose16);
int host_update(const mi355_icp_params* p, const double* sums, int iteration, double* pose16, double* prev_mse,
                int* out_done, int* out_converged);

// ---- kernel launchers (icp_kernels.hip); all asynchronous on `stream` ----
struct DevicePlanes {
  const float* x;
  const float* y;
  const float* z;
};
// staged != nullptr (<= 8 jobs): the descriptors travel as this launch's kernel arguments and its first workgroup writes them
// to jobs_out / rjobs_out for the launches behind it (no stage_jobs_kernel in front)
void launch_nn_scan(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                    DevicePlanes src, DevicePlanes tgt, unsigned long long* keys, uint32_t key_stride,
                    const StageArgs* staged = nullptr, JobDesc* jobs_out = nullptr, RefJobDesc* rjobs_out = nullptr);
void launch_nn_scan_fused(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                          DevicePlanes src, DevicePlanes tgt, unsigned long long* keys3, uint32_t key_stride,
                          uint32_t rank_stride);
void launch_resolve_fused(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                          DevicePlanes tgt, const unsigned long long* keys3, uint32_t key_stride, uint32_t rank_stride,
                          int32_t* out_index, float* out_dist2, double* partials, uint32_t* overflow_list,
                          uint32_t* overflow_count, uint32_t* error_flag);
void launch_nn_scan_xfilter(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                            DevicePlanes src, const float* tcx, const float* tcy, const float* tcz, const float* tt,
                            const void* tile_table, unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride);
void launch_nn_scan_xmfma(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t blocks_x,
                          uint32_t first_block, DevicePlanes src, const float* tcx, const float* tcy, const float* tcz,
                          const float* tt, const void* tile_table, unsigned long long* keys4, uint32_t key_stride,
                          uint32_t rank_stride);
// how a matrix-core scan launch maps workgroups to (source block, run of tiles): nn_scan_xbf16_kernel's kLayout
constexpr int kXbLayoutStatic = 0, kXbLayoutClassed = 1, kXbLayoutQueue = 2;
void launch_nn_scan_xbf16(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x, DevicePlanes src,
                          const void* bf_image, const void* tile_table, unsigned long long* keys4, uint32_t key_stride,
                          uint32_t rank_stride, int shape, const StageArgs* staged = nullptr, JobDesc* jobs_out = nullptr,
                          RefJobDesc* rjobs_out = nullptr, int layout = kXbLayoutStatic, uint32_t grid3_x = 0, uint32_t grid3_z = 0,
                          uint32_t* queue = nullptr);
void launch_nn_scan_xbf16_dump(void* stream, const JobDesc* jobs_dev, uint32_t blocks_x, DevicePlanes src, const void* bf_image,
                               const void* tile_table, unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride,
                               float* dump, uint32_t dump_stride);
// run_len: targets per remembered granule (kFusedChunk, or kXbRun with two key rows per split: split_mul = 2)
// src_perm != nullptr: `src` is a Morton-ordered copy (position -> original index in src_perm); results and overflow ids
// are written at original indices
// wave_points: 64, 32 or 16 points per wave (matrix-core filter only; small launches spread over more waves)
void launch_resolve_xfilter(void* stream, int run_len, uint32_t split_mul, int wave_points, const JobDesc* jobs_dev, int n_jobs,
                            uint32_t max_src_blocks, DevicePlanes src, const uint32_t* src_perm, DevicePlanes sorted_original, const uint32_t* perm,
                            const unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride, int32_t* out_index,
                            float* out_dist2, uint32_t* overflow_list, uint32_t* overflow_count, uint32_t* error_flag);
void launch_nn_pruned(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes sorted_src,
                      const uint32_t* src_perm, DevicePlanes sorted_tgt, const uint32_t* tgt_perm, const void* tgt_table,
                      const void* tgt_sub, const void* tgt_super, int32_t* out_index, float* out_dist2, uint32_t* error_flag);
constexpr uint32_t kOverflowCap = 1024;  // overflow points per job the device-side exact pass takes without the host
void launch_overflow_exact(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t cap, DevicePlanes src, DevicePlanes tgt,
                           const uint32_t* overflow_list, const uint32_t* overflow_count, int32_t* out_index, float* out_dist2,
                           uint32_t* error_flag);
// settles EVERY listed point of every job (its workgroups stride over the list): the pass behind it never needs the host
void launch_overflow_tiles(void* stream, const JobDesc* jobs_dev, int n_jobs, DevicePlanes src, DevicePlanes sorted_tgt,
                           const uint32_t* tgt_perm, const void* tgt_table, const uint32_t* overflow_list, const uint32_t* overflow_count,
                           int32_t* out_index, float* out_dist2, uint32_t* error_flag);
void launch_accumulate(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                       DevicePlanes tgt, const int32_t* index, const float* dist2, double* partials, const PassTail& tail = PassTail{});
void launch_gather_points(void* stream, DevicePlanes src, uint32_t src_off, const uint32_t* list, uint32_t n, float* out_planes,
                          size_t out_stride, uint32_t n_padded);
void launch_scatter_results(void* stream, const uint32_t* list, uint32_t n, const int32_t* idx_tmp, const float* d2_tmp,
                            int32_t* out_index, float* out_dist2, uint32_t out_off);
void launch_resolve(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                    DevicePlanes tgt, unsigned long long* keys, uint32_t key_stride, int32_t* out_index,
                    float* out_dist2, double* partials, uint32_t* error_flag, const PassTail& tail = PassTail{});
void launch_final_reduce(void* stream, const JobDesc* jobs_dev, int n_jobs, const double* partials, double* sums,
                         uint32_t* overflow_count, uint32_t overflow_cap);

// Pyramid levels of ONE cloud in one launch (a new scene builds seven): level q holds the level-0 rows in_off + i * step.
// normals given: a level with `rows` also gets its rows (the normals share the planes' layout: same stride, same offsets).
struct SubsampleLevel {
  float* out;        // the level's planes, x | y | z, each out_stride floats
  void* rows;        // the level's rows (32 bytes per point: x, y, z, nx | ny, nz, 0, 0) or nullptr
  uint32_t out_stride, out_off, in_off, n, padded, step;
};
constexpr int kSubsampleMaxLevels = 8;
struct SubsampleArgs {
  SubsampleLevel lv[kSubsampleMaxLevels];
  int n_levels;
};
void launch_subsample_levels(void* stream, const float* in, size_t in_stride, const SubsampleArgs& args, float pad, const float* normals);
bool launch_rejection_select(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_n, const float* dist2, float scale,
                             float* thr, uint32_t* hists, uint32_t* states, const RefJobDesc* rjobs_dev = nullptr,
                             const int32_t* index = nullptr, unsigned long long* owner = nullptr);
void launch_select_hist(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_n, const float* dist2, uint32_t* hists,
                        uint32_t* states, int pass);
void launch_select_pick(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t* hists, uint32_t* states, int pass, float scale,
                        float* thr);
void launch_picky_scatter(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs,
                          uint32_t max_src_blocks, const int32_t* index, const float* dist2, const float* thr,
                          unsigned long long* owner);
// slab-sharded passes: this rank's offers as a list (block[0] = count, then {key, slot} records) beside the local atomicMin, and
// the other ranks' gathered lists applied to this rank's table
void launch_picky_offer(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs, uint32_t max_src_blocks, const int32_t* index,
                        const float* dist2, const float* thr, unsigned long long* owner, unsigned long long* block, uint32_t capacity);
void launch_picky_apply(void* stream, const unsigned long long* all, uint32_t block_words, int n_ranks, int my_rank, uint32_t capacity,
                        unsigned long long* owner);
// rows: 32 bytes per point of a pyramid level, (x, y, z, nx | ny, nz, 0, 0); normal of level-0 row normal_off + j * step
void launch_target_rows(void* stream, DevicePlanes level, uint32_t level_off, uint32_t n, DevicePlanes normals, uint32_t normal_off,
                        uint32_t step, void* rows);
// source_rows != nullptr: the upstream-faithful variant (model rows with normals, plane equations about RefJobDesc::centre,
// sum 29 = sum |n_p - n_q|^2); nullptr: the round 1-3 form from the source planes (sum 29 = 0)
void launch_ref_accumulate(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs,
                           uint32_t max_src_blocks, DevicePlanes src, const void* source_rows, const void* target_rows,
                           const int32_t* index, const float* dist2, const float* thr, const unsigned long long* owner,
                           uint8_t* selected, double* partials, const PassTail& tail);

// ---- upstream's pre-normalisation (upstream_kernels.hip) ----
// coordinate sums of a cloud in kCloudSumBlocks fixed chunks: partials[b][3] doubles, added up by the host in block order
constexpr int kCloudSumBlocks = 32;
void launch_cloud_sums(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, double* partials);
// per job: sum over the model of |T0 s - c| and over the scene of |q - c| (float32 expression of upstream's
// computeDistToOrigin, summed in double): partials[job][which][kCloudSumBlocks], which = 0 model, 1 scene
struct NormJob {
  float pose[12];   // float32 candidate pose
  float centre[3];
  uint32_t src_off, src_n, tgt_off, tgt_n;
  uint32_t pad_;
};
struct NormArgs {
  uint32_t n, first, pad_[2];
  NormJob jobs[8];
};
void launch_centre_dists(void* stream, const NormArgs& args, DevicePlanes src, DevicePlanes tgt, double* partials);
void launch_ref_final_reduce(void* stream, const JobDesc* jobs_dev, int n_jobs, const double* partials, double* sums, const float* thr,
                             uint32_t* overflow_count, uint32_t overflow_cap);

// ---- sorted / tile-centred target copy for the expansion filter (xtarget_kernels.hip) ----
int xtarget_sort_temp_bytes(uint32_t n, size_t* bytes);
int xtarget_build_cloud_from_parent(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, const uint32_t* parent_perm,
                                    uint32_t n_parent, uint32_t first, uint32_t step, uint32_t* scratch, float* xt, size_t stride,
                                    uint32_t* perm, float* table, void* bf_image, void* sub_table, float* bf_table, float* tile_box);
int xtarget_build_cloud(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, const uint32_t* box, uint32_t* code,
                        uint32_t* index, uint32_t* code2, uint32_t* order, void* temp, size_t temp_bytes, float* xt,
                        size_t stride, uint32_t* perm, float* table, void* bf_image, void* sub_table, float* bf_table, float* tile_box);
void launch_super_tiles(void* stream, const float* table, uint32_t n_tiles_total, float* super_table);
void launch_bbox(void* stream, DevicePlanes planes, const uint32_t* cloud_off, const uint32_t* cloud_n, int n_clouds, uint32_t max_n,
                 uint32_t* box);
void launch_xt_fill(void* stream, float* xt, size_t stride, uint32_t* perm, void* bf_image, float* table, float* bf_table);  // both tables: 4 floats per tile, cleared

// ---- scene normals (normals_kernels.hip) ----
constexpr int kMaxNormalNeighbors = 32;
void launch_knn_scan(void* stream, int k, DevicePlanes planes, uint32_t n, uint32_t n_tiles, uint32_t n_split,
                     float* list_d, int32_t* list_i, uint32_t q_stride);
void launch_knn_pruned(void* stream, int k, DevicePlanes sorted, const uint32_t* perm, const void* table, const void* sub_table,
                       const void* super_table, uint32_t n,
                       uint32_t n_tiles, float* list_d, int32_t* list_i, uint32_t q_stride);
void launch_normals(void* stream, int k, DevicePlanes planes, uint32_t n, uint32_t k_eff, uint32_t n_split,
                    const float* list_d, const int32_t* list_i, uint32_t q_stride, int flip, const float* viewpoint,
                    float* out_aos, float* out_planes, size_t plane_stride, int32_t* out_knn);

// ---- RCCL, bound at run time (rccl_dyn.cpp) ----
struct RcclApi;
const RcclApi* rccl_api(const char** why_not);
int rccl_unique_id(unsigned char* out128, const char** err);
int rccl_comm_init(void** comm, const unsigned char* id128, int rank, int n_ranks, const char** err);
int rccl_comm_query(void* comm, int* rank, int* n_ranks, const char** err);   // ncclCommUserRank / ncclCommCount
int rccl_allgather_bytes(void* comm, const void* dev_send, void* dev_recv, size_t bytes_per_rank, void* stream, const char** err);
int rccl_allreduce_u32_sum(void* comm, uint32_t* dev_buf, size_t count, void* stream, const char** err);
int rccl_allreduce_u64_min(void* comm, unsigned long long* dev_buf, size_t count, void* stream, const char** err);
void rccl_comm_destroy(void* comm);

// ---- the cross-rank exchange step of a slab-sharded run (comm.hip) ----
// One process per GPU; every rank holds a slab of each source cloud and the whole target.  Two transports:
//   native   - an RCCL communicator (mi355_icp_comm_attach): collectives are enqueued on the engine's stream, no host sync;
//   external - a host callback that all-gathers bytes (mi355_icp_comm_attach_external): the caller's transport
//              (torch.distributed over gloo in the tests); the engine stages through pinned host memory.
// Floating-point sums are NEVER reduced by the transport: the per-rank rows are all-gathered and added in rank order
// (SURVEY.md 8e), so every rank holds identical bits and the result does not depend on RCCL's algorithm choice.
// Integer reductions (histogram counts, picky owner keys) are exact in any order and use ncclAllReduce natively.
struct CommState {
  void* rccl = nullptr;
  mi355_icp_allgather_fn ext = nullptr;
  void* ext_user = nullptr;
  int rank = 0, size = 1;
  unsigned char* gather_dev = nullptr; size_t gather_dev_cap = 0;   // native: size x bytes_per_rank
  unsigned char* stage_send = nullptr; unsigned char* stage_recv = nullptr; size_t stage_send_cap = 0, stage_recv_cap = 0;  // external: pinned
  bool attached() const { return rccl != nullptr || ext != nullptr; }
};
// dev_buf[0..count) <- sum over ranks r = 0, 1, ... (in that order) of rank r's dev_buf.  Returns an mi355_icp_status.
int comm_sum_f64_rank_order(CommState& cs, double* dev_buf, size_t count, void* stream, const char** err);
int comm_sum_f64_rank_order_host(CommState& cs, double* host_buf, size_t count, const char** err);  // external transports: rows already on the host
int comm_sum_u32(CommState& cs, uint32_t* dev_buf, size_t count, void* stream, const char** err);
int comm_min_u64(CommState& cs, unsigned long long* dev_buf, size_t count, void* stream, const char** err);
// dev_recv[r * bytes_per_rank ...] <- rank r's dev_send (device buffers; enqueued on `stream`, external transports stage through the host)
int comm_allgather_device(CommState& cs, const void* dev_send, void* dev_recv, size_t bytes_per_rank, void* stream, const char** err);
// host-side all-gather of a few bytes per rank (cloud sizes at the start of a sharded reference-mode call)
int comm_allgather_host(CommState& cs, const void* send, void* recv, size_t bytes_per_rank, void* stream, const char** err);
void comm_release(CommState& cs);  // frees the staging buffers and destroys a native communicator
// this pass's job descriptors (and reference-mode descriptors, or nullptr) to the device tables, asynchronously on `stream`
int stage_jobs(void* stream, const JobDesc* jobs_host, const RefJobDesc* rjobs_host, int n, JobDesc* jobs_dev, RefJobDesc* rjobs_dev);
// a one-thread launch that stores `value` (system scope) to a word of pinned, device-mapped memory once everything before it on the stream is done
void launch_publish_word(void* stream, uint32_t* mapped_word, uint32_t value);
// the same descriptors as one StageArgs value, for a launch that stages them itself (launch_nn_scan); false: more than 8 jobs
bool fill_stage_args(const JobDesc* jobs_host, const RefJobDesc* rjobs_host, int n, StageArgs* out);

// ---- one handle over several devices of one process (multi.cpp; mi355_icp_create_multi) ----
struct MultiState;
int multi_create(const int* device_ids, int n_devices, MultiState** out, std::string* err);
void multi_destroy(MultiState* ms);
const char* multi_last_error(const MultiState* ms);
int multi_devices(const MultiState* ms);
mi355_icp_ctx* multi_sub(MultiState* ms, int slot);
int multi_broadcast(MultiState* ms, const std::function<int(mi355_icp_ctx*, int)>& f);  // f(sub-context, slot) on every device at once
int multi_set_targets(MultiState* ms, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes);
int multi_set_sources(MultiState* ms, int n_clouds, const float* const* xyz, const size_t* n_points, size_t stride_bytes);
int multi_align(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* init_poses, const mi355_icp_params* params,
                double* out_poses, double* out_residual, int* out_iterations, int* out_status, int* out_converged);
int multi_align_reference(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* init_poses,
                          const mi355_icp_reference_params* rp, double* out_poses, double* out_residual, int* out_passes, int* out_status);
int multi_correspond(MultiState* ms, int n_jobs, const int* source_ids, const int* target_ids, const double* poses, float max_corr,
                     int32_t* out_index, float* out_dist2, double* out_sums);
int multi_single_context(MultiState* ms, mi355_icp_ctx** out);  // device slot 0 holding the whole clouds (diagnostic entry points)

}  // namespace mi355

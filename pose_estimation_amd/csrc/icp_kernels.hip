// icp_kernels.hip — gfx950 (CDNA4) kernels of one ICP correspondence pass.
//
//   nn_scan_kernel            N1  fused source transform + brute-force exact 1-NN scan
//   resolve_accumulate_kernel N1/N2  exact index recovery + centroid / cross-covariance partial sums
//   final_reduce_kernel       N2  fixed-order reduction of the per-workgroup partials
//
// (N1/N2: SURVEY.md 8a.  In the reference this work is the KD-tree query and the
// least-squares accumulation inside cv::ppf_match_3d::ICP::registerModelToScene,
// called at pose_estimation/src/opencv_surface_match.cpp:94.)
//
// Arithmetic contract (bit-exact against oracle/icp_oracle.c): float32, every
// product and sum rounded separately.  This file MUST be compiled with
// -ffp-contract=off; csrc/Makefile greps the ISA for fused ops after the build.
//
// Why the scan is shaped this way.  8 float ops + a min per (source, target)
// pair against 32 bytes of compulsory traffic per source point makes the scan
// FP32-VALU-issue bound by four orders of magnitude, so the design
// goal is "no VALU instruction that is not one of those nine":
//   * a lane keeps 2, 4 or 8 source points in registers (template parameter); the target tile is
//     SoA in LDS and is read with wave-uniform ds_read_b128 (a broadcast: 3 LDS
//     instructions feed 4 targets x 4 sources x 64 lanes = 1024 pairs);
//   * the inner loop tracks only the running minimum (v_min3_f32 folds two
//     candidates per instruction).  WHICH target won is not tracked per pair:
//     once per 32-target chunk the lane notes whether its minimum improved, and
//     the exact index is recovered afterwards by re-scanning that one chunk
//     (resolve_accumulate_kernel) with the same arithmetic, where equality of
//     the float32 bits identifies the winner and "first hit" gives the
//     lowest-index tie rule of the oracle;
//   * the target range is also split across workgroups so small clouds still
//     fill 256 CUs; partial results are 64-bit keys (d2 bits << 32 | chunk id)
//     whose unsigned minimum is the lowest distance, then the lowest chunk.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include <cstdint>
#include <cstdlib>

#include "icp_internal.h"

namespace mi355 {

#ifdef MI355_ICP_DIAG
// Diagnostic build only (make diag): per-workgroup start/end stamps of the scan, 100 MHz realtime counter.
// Never compiled into the shipped library; stamps leave the kernel through this buffer alone.
__device__ unsigned long long* g_diag_stamps = nullptr;
#endif
#ifdef MI355_ICP_DIAG
// ... and phase stamps of ONE workgroup per launch of the small-pass kernels (tools/p2p_kernel_phases.py): records of 16 words,
// [kernel id | n << 8, n_stamps, t0, t1, ...]
__device__ unsigned long long* g_diag_p2p = nullptr;
__device__ unsigned int g_diag_p2p_n = 0;
struct P2pDiag {
  unsigned long long t[14];
  int n = 0;
  bool on = false;
  __device__ void begin(bool who) { on = who && g_diag_p2p != nullptr; n = 0; stamp(); }
  __device__ void stamp() { if (on && n < 14) t[n++] = wall_clock64(); }
  __device__ void end(unsigned long long id) {
    if (!on) return;
    stamp();
    const unsigned int slot = atomicAdd(&g_diag_p2p_n, 1u);
    if (slot >= 4096u) return;
    g_diag_p2p[slot * 16ull] = id;
    g_diag_p2p[slot * 16ull + 1] = static_cast<unsigned long long>(n);
    for (int k = 0; k < n; ++k) g_diag_p2p[slot * 16ull + 2 + k] = t[k];
  }
};
#define P2P_DIAG_BEGIN(who) P2pDiag pd; pd.begin(who)
#define P2P_DIAG_STAMP() pd.stamp()
#define P2P_DIAG_STAMP_WAIT() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pd.stamp(); } while (0)  // loads landed
#define P2P_DIAG_END(id) pd.end(id)
#else
#define P2P_DIAG_BEGIN(who) ((void)0)
#define P2P_DIAG_STAMP() ((void)0)
#define P2P_DIAG_STAMP_WAIT() ((void)0)
#define P2P_DIAG_END(id) ((void)0)
#endif

namespace {

__device__ __forceinline__ void transform_point(const float* __restrict__ m, float x, float y, float z, float& ox,
                                                float& oy, float& oz) {
  float a, b;
  a = m[0] * x; b = m[1] * y; a = a + b; b = m[2] * z; a = a + b; ox = a + m[3];
  a = m[4] * x; b = m[5] * y; a = a + b; b = m[6] * z; a = a + b; oy = a + m[7];
  a = m[8] * x; b = m[9] * y; a = a + b; b = m[10] * z; a = a + b; oz = a + m[11];
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float tx, float ty, float tz) {
  const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  float a = dx * dx;
  const float b = dy * dy, c = dz * dz;
  a = a + b;
  return a + c;
}

__device__ __forceinline__ float min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// N1: scan.  grid = (max_j src_blocks_j * n_split_j, n_jobs), block = 256.
// Workgroup (sb, sp) of job j scans source points [sb*256*P, +256*P) against target tiles
// run sp (JobDesc::split_q/split_r) and writes one key per source point.
template <int kPointsPerLane>
__global__ __launch_bounds__(kBlockThreads) void nn_scan_kernel(const JobDesc* __restrict__ jobs,
                                                                 const float* __restrict__ sx,
                                                                 const float* __restrict__ sy,
                                                                 const float* __restrict__ sz,
                                                                 const float* __restrict__ tx,
                                                                 const float* __restrict__ ty,
                                                                 const float* __restrict__ tz,
                                                                 unsigned long long* __restrict__ keys,
                                                                 uint32_t key_stride, StageArgs staged,
                                                                 JobDesc* __restrict__ jobs_out, RefJobDesc* __restrict__ rjobs_out) {
  // [buffer][plane][tile/4] float4: 2 x 12 KiB
  __shared__ float4 lds[2][3][kTile / 4];

#ifdef MI355_ICP_DIAG
  const unsigned long long diag_t0 = wall_clock64();
  const unsigned long long diag_c0 = clock64();
#endif
  P2P_DIAG_BEGIN(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
  // staged.n != 0: this launch IS the staging (stage_jobs_kernel's job, one dependent launch less per small pass): every
  // workgroup takes its descriptor from the kernel arguments, the first one also writes the device tables for the
  // launches behind this one
  if (staged.n && blockIdx.x == 0 && blockIdx.y == 0) {
    constexpr uint32_t kJ = sizeof(JobDesc) / 4, kR = sizeof(RefJobDesc) / 4;
    const uint32_t* src_w = reinterpret_cast<const uint32_t*>(staged.jobs);
    uint32_t* dst_w = reinterpret_cast<uint32_t*>(jobs_out);
    for (uint32_t w = threadIdx.x; w < staged.n * kJ; w += kBlockThreads) dst_w[w] = src_w[w];
    if (staged.has_ref) {
      const uint32_t* rs = reinterpret_cast<const uint32_t*>(staged.rjobs);
      uint32_t* rd = reinterpret_cast<uint32_t*>(rjobs_out);
      for (uint32_t w = threadIdx.x; w < staged.n * kR; w += kBlockThreads) rd[w] = rs[w];
    }
  }
  const JobDesc J = staged.n ? staged.jobs[blockIdx.y] : jobs[blockIdx.y];
  const uint32_t n_split = J.n_split;
  const uint32_t sb = blockIdx.x / n_split;
  const uint32_t sp = blockIdx.x - sb * n_split;
  if (sb >= J.scan_blocks) return;
  constexpr uint32_t kScanSrcPerBlock = kBlockThreads * kPointsPerLane;
  // balanced runs of tiles (all non-empty because n_split <= tgt_tiles); sp is the fast index of blockIdx.x and
  // n_split is 1, 2, 4 or a multiple of 8, so workgroups b and b+8 (one XCD: MI355X_MICROARCH.md "Workgroup
  // dispatch") walk the same run of target tiles at about the same time and that XCD's L2 serves the re-reads
  uint32_t tile0 = sp * J.split_q + (sp < J.split_r ? sp : J.split_r);
  uint32_t tile1 = tile0 + J.split_q + (sp < J.split_r ? 1u : 0u);
  // small launches (JobDesc::sub_split): split sp is one run of chunks of ONE tile
  uint32_t chunk0 = 0, chunk1 = kChunksPerTile;
  if (J.sub_split > 1) {
    tile0 = sp / J.sub_split;
    tile1 = tile0 + 1;
    const uint32_t per = kChunksPerTile / J.sub_split;  // sub_split divides kChunksPerTile (planner)
    chunk0 = (sp - tile0 * J.sub_split) * per;
    chunk1 = chunk0 + per;
  }

  const uint32_t tid = threadIdx.x;
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];

  float qx[kPointsPerLane], qy[kPointsPerLane], qz[kPointsPerLane];
  float best[kPointsPerLane], prev[kPointsPerLane];
  uint32_t cid[kPointsPerLane];
  const uint32_t src_base = J.src_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const uint32_t i = src_base + p * kBlockThreads;  // planes are padded to kSrcPad points: always in bounds
    transform_point(pose, sx[i], sy[i], sz[i], qx[p], qy[p], qz[p]);
    best[p] = __builtin_inff();
    prev[p] = __builtin_inff();
    cid[p] = tile0 * kChunksPerTile + chunk0;
  }

  const float4* __restrict__ gx = reinterpret_cast<const float4*>(tx + J.tgt_off);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(ty + J.tgt_off);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(tz + J.tgt_off);

  // prologue: first tile -> LDS buffer 0 (target planes are padded to whole tiles)
  {
    const uint32_t g = tile0 * (kTile / 4) + tid;
    lds[0][0][tid] = gx[g];
    lds[0][1][tid] = gy[g];
    lds[0][2][tid] = gz[g];
  }
  __syncthreads();
  P2P_DIAG_STAMP();  // 1: descriptor, transforms, first tile in LDS

  uint32_t cur = 0;
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    const bool more = tile + 1 < tile1;
    float4 rx, ry, rz;
    if (more) {  // prefetch the next tile into registers; its latency hides under this tile's scan
      const uint32_t g = (tile + 1) * (kTile / 4) + tid;
      rx = gx[g];
      ry = gy[g];
      rz = gz[g];
    }
    const float4* __restrict__ LX = lds[cur][0];
    const float4* __restrict__ LY = lds[cur][1];
    const float4* __restrict__ LZ = lds[cur][2];
    const uint32_t chunk_base = tile * kChunksPerTile;
    for (uint32_t c = chunk0; c < chunk1; ++c) {
#pragma unroll
      for (uint32_t g = 0; g < kChunk / 4; ++g) {
        const float4 X = LX[c * (kChunk / 4) + g];  // wave-uniform address: LDS broadcast
        const float4 Y = LY[c * (kChunk / 4) + g];
        const float4 Z = LZ[c * (kChunk / 4) + g];
#pragma unroll
        for (int p = 0; p < kPointsPerLane; ++p) {
          const float d0 = dist2(qx[p], qy[p], qz[p], X.x, Y.x, Z.x);
          const float d1 = dist2(qx[p], qy[p], qz[p], X.y, Y.y, Z.y);
          const float d2 = dist2(qx[p], qy[p], qz[p], X.z, Y.z, Z.z);
          const float d3 = dist2(qx[p], qy[p], qz[p], X.w, Y.w, Z.w);
          best[p] = min3(best[p], d0, d1);
          best[p] = min3(best[p], d2, d3);
        }
      }
      const uint32_t chunk = chunk_base + c;
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) {
        if (best[p] < prev[p]) cid[p] = chunk;  // strict: the FIRST chunk that reaches the final minimum wins
        prev[p] = best[p];
      }
    }
    if (more) {
      lds[cur ^ 1][0][tid] = rx;
      lds[cur ^ 1][1][tid] = ry;
      lds[cur ^ 1][2][tid] = rz;
    }
    __syncthreads();
    cur ^= 1;
  }

  // One key row per split - except in sub_split mode, where the (many, small) workgroups of a point merge into ONE row with
  // an integer atomicMin: the unsigned minimum of (d2 bits, chunk id) is the same whatever the order, and the resolve then
  // reads one key per point instead of one per workgroup (the host clears the row before the launch).
  P2P_DIAG_STAMP();  // 2: tiles scanned
  const uint32_t out = (J.sub_split > 1 ? 0u : sp * key_stride) + J.out_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const unsigned long long key =
        (static_cast<unsigned long long>(__float_as_uint(best[p])) << 32) | static_cast<unsigned long long>(cid[p]);
    if (J.sub_split > 1) atomicMin(&keys[out + p * kBlockThreads], key);
    else keys[out + p * kBlockThreads] = key;
  }
  P2P_DIAG_END(3ull | (static_cast<unsigned long long>(J.src_n) << 8));
#ifdef MI355_ICP_DIAG
  if (g_diag_stamps && threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const size_t slot = (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4;
    g_diag_stamps[slot + 0] = diag_t0;
    g_diag_stamps[slot + 1] = wall_clock64();
    g_diag_stamps[slot + 2] = (static_cast<unsigned long long>(xcc & 0xf) << 32) | hwid;
    g_diag_stamps[slot + 3] = clock64() - diag_c0;  // shader-clock cycles this workgroup lived
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// N1, experimental variant: the target is not staged through LDS; every wave reads it with SCALAR loads
// (uniform address, constant address space -> s_load_dwordx8 into SGPRs) and the VALU ops take the target
// coordinate as their one scalar operand.  No LDS, no barriers.  Same arithmetic, same keys.
// Selected with MI355_ICP_SCAN_MODE=sgpr (A/B only; the LDS kernel above is the default).
typedef const __attribute__((address_space(4))) float* const_f32_ptr;

template <int kPointsPerLane>
__global__ __launch_bounds__(kBlockThreads) void nn_scan_sgpr_kernel(const JobDesc* __restrict__ jobs,
                                                                      const float* __restrict__ sx,
                                                                      const float* __restrict__ sy,
                                                                      const float* __restrict__ sz,
                                                                      const float* __restrict__ tx,
                                                                      const float* __restrict__ ty,
                                                                      const float* __restrict__ tz,
                                                                      unsigned long long* __restrict__ keys,
                                                                      uint32_t key_stride) {
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t n_split = J.n_split;
  const uint32_t sb = blockIdx.x / n_split;
  const uint32_t sp = blockIdx.x - sb * n_split;
  if (sb >= J.scan_blocks) return;
  constexpr uint32_t kScanSrcPerBlock = kBlockThreads * kPointsPerLane;
  const uint32_t tile0 = sp * J.split_q + (sp < J.split_r ? sp : J.split_r);
  const uint32_t tile1 = tile0 + J.split_q + (sp < J.split_r ? 1u : 0u);
  const uint32_t tid = threadIdx.x;
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  float qx[kPointsPerLane], qy[kPointsPerLane], qz[kPointsPerLane];
  float best[kPointsPerLane], prev[kPointsPerLane];
  uint32_t cid[kPointsPerLane];
  const uint32_t src_base = J.src_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const uint32_t i = src_base + p * kBlockThreads;
    transform_point(pose, sx[i], sy[i], sz[i], qx[p], qy[p], qz[p]);
    best[p] = __builtin_inff();
    prev[p] = __builtin_inff();
    cid[p] = tile0 * kChunksPerTile;
  }
  const_f32_ptr cx = (const_f32_ptr)(tx + J.tgt_off);
  const_f32_ptr cy = (const_f32_ptr)(ty + J.tgt_off);
  const_f32_ptr cz = (const_f32_ptr)(tz + J.tgt_off);
  const uint32_t chunk0 = tile0 * kChunksPerTile, chunk1 = tile1 * kChunksPerTile;
  for (uint32_t chunk = chunk0; chunk < chunk1; ++chunk) {
#pragma unroll
    for (uint32_t g = 0; g < kChunk / 8; ++g) {
      const uint32_t base = chunk * kChunk + g * 8;
      float X[8], Y[8], Z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        X[e] = cx[base + e];
        Y[e] = cy[base + e];
        Z[e] = cz[base + e];
      }
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float d0 = dist2(qx[p], qy[p], qz[p], X[e], Y[e], Z[e]);
          const float d1 = dist2(qx[p], qy[p], qz[p], X[e + 1], Y[e + 1], Z[e + 1]);
          best[p] = min3(best[p], d0, d1);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < kPointsPerLane; ++p) {
      if (best[p] < prev[p]) cid[p] = chunk;
      prev[p] = best[p];
    }
  }
  const uint32_t out = sp * key_stride + J.out_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const unsigned long long key =
        (static_cast<unsigned long long>(__float_as_uint(best[p])) << 32) | static_cast<unsigned long long>(cid[p]);
    keys[out + p * kBlockThreads] = key;
  }
}


// ---------------------------------------------------------------------------------------------
// N1 (index recovery) + N2 (accumulation).  grid = (max_j src_blocks_j, n_jobs), block = 256.
// Same (block, lane, p) -> source point mapping as the scan.
// kSums = false (reference mode: its sums are the point-to-plane ones of ref_accumulate_kernel) recovers the indices only,
// on a grid of (4 * kResolvePointsPerLane * max_j src_blocks_j, n_jobs) of ONE-WAVE workgroups with ONE point per lane - the passes of the coarse
// pyramid levels are a chain of latency-bound launches, and four dependent points per lane is four times the latency -
// and, where the scan merged its workgroups into one key row (sub_split), hands the row back cleared for the next pass.
template <bool kSums>
__global__ __launch_bounds__(kBlockThreads) void resolve_accumulate_kernel(
    const JobDesc* __restrict__ jobs, const float* __restrict__ sx, const float* __restrict__ sy,
    const float* __restrict__ sz, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, unsigned long long* __restrict__ keys, uint32_t key_stride,
    int32_t* __restrict__ out_index, float* __restrict__ out_dist2, double* partials,
    uint32_t* __restrict__ error_flag, PassTail tail) {
  __shared__ double scratch[kSums ? kBlockThreads / 64 : 1][MI355_ICP_NSUMS];
  __shared__ uint32_t last_flag;

  P2P_DIAG_BEGIN(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
  const JobDesc& J = jobs[blockIdx.y];
  // kSums = false: a workgroup may be a single wave (blockDim.x = 64: the 256 points of a quarter block spread over four
  // workgroups, i.e. over four CUs' load units - the chunk re-scan is bound by the issue of its scattered loads)
  // (shifts and masks: a division by a run-time block size expands into a float reciprocal, which the ISA check of this file
  // would rightly refuse to tell from fused point arithmetic)
  static_assert(kResolvePointsPerLane == 4 && kBlockThreads == 256, "the shifts below");
  const uint32_t sub_log2 = (!kSums && blockDim.x == 64u) ? 2u : 0u;  // workgroups per 256-point quarter block: 4 or 1
  const uint32_t sb = kSums ? blockIdx.x : blockIdx.x >> (2u + sub_log2);
  if (sb >= J.src_blocks) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t slot = kSums ? tid : (blockIdx.x & ((1u << sub_log2) - 1u)) * blockDim.x + tid;  // position inside the 256-point quarter block
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  const float max_d2 = J.max_d2;
  const uint32_t n_split = J.sub_split > 1 ? 1u : J.n_split;  // sub_split mode: the scan merged its workgroups into row 0

  double acc[MI355_ICP_NSUMS];
#pragma unroll
  for (int k = 0; k < MI355_ICP_NSUMS; ++k) acc[k] = 0.0;

  const int p_first = kSums ? 0 : static_cast<int>((blockIdx.x >> sub_log2) & 3u);
  constexpr int kPts = kSums ? kResolvePointsPerLane : 1;
  // A point is a chain of dependent reads: its key (one per split) -> the chunk the key names.  The keys and the source
  // coordinates of ALL the lane's points are read first, side by side - only the chunk reads are left to follow one another -
  // so the lane waits for 1 + kPts round trips, not 3 kPts: at 1k<->1k that chain IS the launch (phase stamps, round 3).
  unsigned long long key_of[kPts];
  float s0x_of[kPts], s0y_of[kPts], s0z_of[kPts];
#pragma unroll
  for (int u = 0; u < kPts; ++u) {
    const uint32_t local = sb * kSrcPerBlock + (p_first + u) * kBlockThreads + slot;
    const bool there = local < J.src_n;
    unsigned long long key = ~0ull;
    for (uint32_t sp = 0; sp < n_split; ++sp) {
      const unsigned long long k = keys[sp * key_stride + J.out_off + (there ? local : 0u)];
      key = k < key ? k : key;
    }
    key_of[u] = key;
    const uint32_t i = J.src_off + (there ? local : 0u);
    s0x_of[u] = sx[i];
    s0y_of[u] = sy[i];
    s0z_of[u] = sz[i];
  }
  P2P_DIAG_STAMP();  // 1: descriptor read, key and source loads issued
#pragma unroll
  for (int u = 0; u < kPts; ++u) {
    const int p = p_first + u;
    const uint32_t local = sb * kSrcPerBlock + p * kBlockThreads + slot;
    // the scan's padding lanes merge keys as well: the whole span of this workgroup goes back cleared (a job's slots are
    // padded to kSrcPad, a multiple of this kernel's span and of the scan's)
    if (!kSums && J.sub_split > 1 && local >= J.src_n) keys[J.out_off + local] = ~0ull;
    if (local < J.src_n) {
      const unsigned long long key = key_of[u];
      if (!kSums && J.sub_split > 1) keys[J.out_off + local] = ~0ull;
      const uint32_t best_bits = static_cast<uint32_t>(key >> 32);
      const uint32_t chunk = static_cast<uint32_t>(key);
      const float s0x = s0x_of[u], s0y = s0y_of[u], s0z = s0z_of[u];
      float qx, qy, qz;
      transform_point(pose, s0x, s0y, s0z, qx, qy, qz);

      const uint32_t tbase = J.tgt_off + chunk * kChunk;
      const float4* __restrict__ gx = reinterpret_cast<const float4*>(tx + tbase);
      const float4* __restrict__ gy = reinterpret_cast<const float4*>(ty + tbase);
      const float4* __restrict__ gz = reinterpret_cast<const float4*>(tz + tbase);
      int found = -1;
      float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
      for (int g = 0; g < kChunk / 4; ++g) {
        const float4 X = gx[g], Y = gy[g], Z = gz[g];
        const float xs[4] = {X.x, X.y, X.z, X.w};
        const float ys[4] = {Y.x, Y.y, Y.z, Y.w};
        const float zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = dist2(qx, qy, qz, xs[e], ys[e], zs[e]);
          const bool hit = (found < 0) && (__float_as_uint(d) == best_bits);
          if (hit) {
            found = g * 4 + e;
            mx = xs[e];
            my = ys[e];
            mz = zs[e];
          }
        }
      }
      if (found < 0) {
        // Cannot happen for finite inputs (the scan and this loop evaluate identical expressions).
        // With a NaN distance (overflowed transform) report the oracle's answer for "nothing below +inf".
        atomicOr(error_flag, 1u);
        found = 0;
        mx = gx[0].x;
        my = gy[0].x;
        mz = gz[0].x;
      }
      const float best = __uint_as_float(best_bits);
      const uint32_t o = J.out_off + local;
      out_index[o] = static_cast<int32_t>(chunk * kChunk + static_cast<uint32_t>(found));
      out_dist2[o] = best;
      if (kSums && best <= max_d2) {
        const double a0 = s0x, a1 = s0y, a2 = s0z, b0 = mx, b1 = my, b2 = mz;
        acc[0] += 1.0;
        acc[1] += a0; acc[2] += a1; acc[3] += a2;
        acc[4] += b0; acc[5] += b1; acc[6] += b2;
        acc[7] += a0 * b0; acc[8] += a0 * b1; acc[9] += a0 * b2;
        acc[10] += a1 * b0; acc[11] += a1 * b1; acc[12] += a1 * b2;
        acc[13] += a2 * b0; acc[14] += a2 * b1; acc[15] += a2 * b2;
        acc[16] += static_cast<double>(best);
      }
    }
  }

  P2P_DIAG_STAMP();  // 2: chunks re-scanned, products accumulated
  if (!kSums) P2P_DIAG_END(2ull | (static_cast<unsigned long long>(J.src_n) << 8));
  if (kSums) {
    const double r = pass_block_sum<MI355_ICP_NSUMS>(acc, scratch);
    P2P_DIAG_STAMP();  // 3: block sum
    P2P_DIAG_END(1ull | (static_cast<unsigned long long>(J.src_n) << 8));
    double* job_rows = partials + static_cast<size_t>(J.partial_off) * MI355_ICP_NSUMS;
    if (!tail.done) {
      if (tid < MI355_ICP_NSUMS) job_rows[static_cast<size_t>(sb) * MI355_ICP_NSUMS + tid] = r;
      return;
    }
    // single-GPU align loop: this launch ends the pass (finish_pass, icp_internal.h).  This kernel can raise the error
    // word itself, so the job's last workgroup looks at it again right before it publishes (reread_error)
    const uint32_t raised = 0u;
    finish_pass<MI355_ICP_NSUMS>(r, acc, scratch, &last_flag, job_rows, sb, J.src_blocks, tail, raised,
                                 [&](double* t, uint32_t job, uint32_t n_jobs) { publish_overflow(tail, job, t + job, t + n_jobs + job); }, true);
  }
}

// ---------------------------------------------------------------------------------------------
// Exact half of the fused-filter path (icp_fused_kernels.hip explains the filter).  One lane per source point: take
// the three smallest surrogate chunk minima over all splits, keep the chunks inside the error band of the smallest,
// re-scan those (one or two 64-target chunks) with the EXACT unfused formula and pick the (d2, index)-lexicographic
// minimum.  Three or more chunks in the band -> the point goes to the overflow list and is answered by the exact scan.
__global__ __launch_bounds__(kBlockThreads) void resolve_fused_kernel(
    const JobDesc* __restrict__ jobs, const float* __restrict__ sx, const float* __restrict__ sy,
    const float* __restrict__ sz, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, const unsigned long long* __restrict__ keys3, uint32_t key_stride,
    uint32_t rank_stride, int32_t* __restrict__ out_index, float* __restrict__ out_dist2,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ overflow_count, uint32_t* __restrict__ error_flag) {
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t sb = blockIdx.x;
  if (sb >= J.src_blocks) return;
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  const uint32_t n_split = J.n_split;
#pragma unroll 1
  for (int p = 0; p < kResolvePointsPerLane; ++p) {
    const uint32_t local = sb * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    if (local >= J.src_n) continue;
    unsigned long long k1 = ~0ull, k2 = ~0ull, k3 = ~0ull;
    for (uint32_t sp = 0; sp < n_split; ++sp) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned long long k = keys3[static_cast<size_t>(r) * rank_stride + static_cast<size_t>(sp) * key_stride + J.out_off + local];
        const bool l1 = k < k1, l2 = k < k2, l3 = k < k3;
        k3 = l2 ? k2 : (l3 ? k : k3);
        k2 = l1 ? k1 : (l2 ? k : k2);
        k1 = l1 ? k : k1;
      }
    }
    // error band of the fused surrogate around its minimum (relative 2^-19 covers 2 x 4 roundings with margin,
    // absolute 2^-140 covers gradual underflow); evaluated in double so the band itself is exact
    const double v1 = static_cast<double>(__uint_as_float(static_cast<uint32_t>(k1 >> 32)));
    const double band = v1 * (1.0 + 1.9073486328125e-06) + 7.1746481373430634e-43;
    const bool in2 = static_cast<double>(__uint_as_float(static_cast<uint32_t>(k2 >> 32))) <= band;
    const bool in3 = static_cast<double>(__uint_as_float(static_cast<uint32_t>(k3 >> 32))) <= band;
    if (in3) {  // too many candidate chunks (or everything is +inf): let the exact scan answer
      const uint32_t pos = atomicAdd(&overflow_count[blockIdx.y], 1u);
      overflow_list[J.out_off + pos] = local;
      // a valid placeholder: the kernels that follow in the stream must stay in bounds even if this point's answer only
      // arrives after the host has looked at the counts (more than kOverflowCap points: the pass is repeated)
      out_index[J.out_off + local] = 0;
      out_dist2[J.out_off + local] = __builtin_inff();
      continue;
    }
    const uint32_t i = J.src_off + local;
    float qx, qy, qz;
    transform_point(pose, sx[i], sy[i], sz[i], qx, qy, qz);
    float best = __builtin_inff();
    uint32_t best_j = 0xFFFFFFFFu;
    const int n_cand = in2 ? 2 : 1;
    for (int cnd = 0; cnd < n_cand; ++cnd) {
      const uint32_t chunk = static_cast<uint32_t>(cnd == 0 ? k1 : k2);
      const uint32_t tbase = J.tgt_off + chunk * kFusedChunk;
      const float4* __restrict__ gx = reinterpret_cast<const float4*>(tx + tbase);
      const float4* __restrict__ gy = reinterpret_cast<const float4*>(ty + tbase);
      const float4* __restrict__ gz = reinterpret_cast<const float4*>(tz + tbase);
#pragma unroll 4
      for (int g = 0; g < kFusedChunk / 4; ++g) {
        const float4 X = gx[g], Y = gy[g], Z = gz[g];
        const float xs[4] = {X.x, X.y, X.z, X.w};
        const float ys[4] = {Y.x, Y.y, Y.z, Y.w};
        const float zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = dist2(qx, qy, qz, xs[e], ys[e], zs[e]);
          const uint32_t j = chunk * kFusedChunk + static_cast<uint32_t>(g * 4 + e);
          if (d < best || (d == best && j < best_j)) {
            best = d;
            best_j = j;
          }
        }
      }
    }
    if (best_j == 0xFFFFFFFFu) {  // cannot happen for finite data (a finite surrogate minimum implies a finite exact d2 nearby)
      atomicOr(error_flag, 1u);
      best_j = 0;
    }
    out_index[J.out_off + local] = static_cast<int32_t>(best_j);
    out_dist2[J.out_off + local] = best;
  }
}

// Exact half of the expansion-filter paths (nn_scan_xfilter_kernel, nn_scan_xbf16_kernel).  keys4: the three smallest
// lower bounds L (with granule id) per key row.  The re-scan reads the SORTED copy of the original coordinates (sox, soy,
// soz: sorted order, uncentred) and maps positions back through perm; ties are resolved on the ORIGINAL index, which is
// what the oracle's lowest-index rule refers to.
//
// Wave-cooperative: each lane first merges the key rows of its own source point; then the wave takes its (up to) 64
// points one at a time and its 64 lanes read the candidate granule side by side - 256 contiguous bytes per plane per
// load instead of 64 scattered lines - compute the exact contract distance, and reduce (d2, original index)
// lexicographically across the wave.
//
// Granule 1 (smallest lower bound) is always re-scanned; its exact minimum d* then decides: granule 2 is needed iff
// L2 <= d* (<=: an exact tie with a lower original index could sit there), and if afterwards L3 <= d*, a fourth granule
// the scan did not keep could matter too - only then does the all-exact scan answer (overflow list).
// Wave-wide unsigned minimum as a scalar.  Four DPP butterfly steps inside each row of 16 lanes (xor 1, xor 2, then the
// two mirrors, which pair equal-valued groups just as xor 4 and xor 8 would), then the four row results are read into
// SGPRs.  All 64 lanes must be active.  (A __shfl_xor butterfly is six dependent LDS round trips: ~10x the latency.)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t o;
  o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  v = o < v ? o : v;
  o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
  v = o < v ? o : v;
  o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = o < v ? o : v;
  o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x140, 0xF, 0xF, false));  // row_mirror
  v = o < v ? o : v;
  const uint32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const uint32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}

// Wave-wide unsigned minima of EIGHT values per lane at once (one per point of a resolve step).  Reducing them one after
// the other costs eight butterflies; here the first three exchange steps also TRANSPOSE - a lane hands half of its values
// to its partner and keeps the other half, so after strides 1, 2, 4 it holds ONE point's partial minimum (over its group of
// 8 lanes) - and three plain steps (8, 16, 32) finish it: 7 + 3 minima instead of 48.  Lane L ends up with the minimum of
// point k(L) = 4 (L & 1) + 2 (L >> 1 & 1) + (L >> 2 & 1); wave_min8_of() reads point k's back as a scalar.  All 64 lanes active.
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_min8(const uint32_t (&v)[8]) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool b0 = lane & 1u, b1 = lane & 2u, b2 = lane & 4u;
  uint32_t w[4], x[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // stride 1: lanes with bit 0 clear keep points 0-3
    const uint32_t mine = b0 ? v[4 + i] : v[i], give = b0 ? v[i] : v[4 + i];
    w[i] = umin32(mine, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(give), 0xB1, 0xF, 0xF, false)));  // quad_perm [1,0,3,2]
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // stride 2: bit 1 clear keeps the lower two of those
    const uint32_t mine = b1 ? w[2 + i] : w[i], give = b1 ? w[i] : w[2 + i];
    x[i] = umin32(mine, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(give), 0x4E, 0xF, 0xF, false)));  // quad_perm [2,3,0,1]
  }
  uint32_t r;
  {  // stride 4 (ds_swizzle, bit mode: and 0x1F, or 0, xor 4)
    const uint32_t mine = b2 ? x[1] : x[0], give = b2 ? x[0] : x[1];
    r = umin32(mine, static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(give), 0x101F)));
  }
  r = umin32(r, static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(r), 0x201F)));  // xor 8
  r = umin32(r, static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(r), 0x401F)));  // xor 16
  r = umin32(r, static_cast<uint32_t>(__shfl_xor(static_cast<int>(r), 32, 64)));
  return r;
}
__device__ __forceinline__ uint32_t wave_min8_of(uint32_t r, int k) {  // k: compile-time point number
  return __builtin_amdgcn_readlane(r, ((k >> 2) & 1) | (((k >> 1) & 1) << 1) | ((k & 1) << 2));
}

// kRun: targets per remembered granule, kFusedChunk (VALU filter) or kXbRun (matrix-core filter).
// kWavePoints: points per wave (lanes past it idle through the key merge and lend a hand in the granule scans).  A wave
// works through its points kStep at a time, each step one dependent round trip to the caches: 64 points are 8 trips
// (~2.2 us each: tools/p2p_kernel_phases.py).  A launch too small to fill the chip with 64-point waves (a request's
// candidates: 30k-120k points = 0.5-2 waves per SIMD) takes 16 or 32 points per wave on 4 or 2 times the waves instead.
template <int kRun, int kWavePoints = 64>
__global__ __launch_bounds__(kBlockThreads) void resolve_xfilter_kernel(
    const JobDesc* __restrict__ jobs, uint32_t split_mul, const float* __restrict__ sx, const float* __restrict__ sy,
    const float* __restrict__ sz, const uint32_t* __restrict__ src_perm, const float* __restrict__ sox, const float* __restrict__ soy,
    const float* __restrict__ soz, const uint32_t* __restrict__ perm, const unsigned long long* __restrict__ keys4,
    uint32_t key_stride, uint32_t rank_stride, int32_t* __restrict__ out_index, float* __restrict__ out_dist2,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ overflow_count) {
  static_assert(kRun % 64 == 0, "a granule is read 64 targets at a time");
  static_assert(kWavePoints == 64 || kWavePoints == 32 || kWavePoints == 16, "points per wave");
  constexpr uint32_t kBlockPoints = kBlockThreads / 64 * kWavePoints;
  const JobDesc& J = jobs[blockIdx.y];
  if (blockIdx.x * kBlockPoints >= J.src_n) return;  // grid.x covers the largest job in groups of kBlockPoints points
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  const uint32_t n_rows = J.n_split * split_mul;  // key rows: one per split (two for the matrix-core filter's wave halves)
  const uint32_t lane = threadIdx.x & 63u;
  P2P_DIAG_BEGIN(blockIdx.x % 16 == 0 && blockIdx.y == 0 && threadIdx.x == 0);
  {
    const uint32_t local = blockIdx.x * kBlockPoints + (threadIdx.x >> 6) * kWavePoints + lane;
    const bool valid = lane < static_cast<uint32_t>(kWavePoints) && local < J.src_n;
    unsigned long long k1 = ~0ull, k2 = ~0ull, k3 = ~0ull;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
      // four key rows (twelve loads) in flight at a time: one row per trip made this loop a chain of n_rows memory round
      // trips - with the 16-32 rows of a mid-size launch that chain WAS the launch (~28 us whatever the point count)
      for (uint32_t row0 = 0; row0 < n_rows; row0 += 4) {
        unsigned long long kk[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool there = row0 + j < n_rows;
          const size_t base = static_cast<size_t>(there ? row0 + j : row0) * key_stride + J.out_off + local;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const unsigned long long k = keys4[static_cast<size_t>(r) * rank_stride + base];
            kk[j][r] = there ? k : ~0ull;  // a row past the end must not enter twice
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const unsigned long long k = kk[j][r];
            const bool l1 = k < k1, l2 = k < k2, l3 = k < k3;
            k3 = l2 ? k2 : (l3 ? k : k3);
            k2 = l1 ? k1 : (l2 ? k : k2);
            k1 = l1 ? k : k1;
          }
      }
      const uint32_t i = J.src_off + local;
      transform_point(pose, sx[i], sy[i], sz[i], qx, qy, qz);
    }
    const uint32_t g1 = static_cast<uint32_t>(k1), g2 = static_cast<uint32_t>(k2);
    const float L2 = __uint_as_float(static_cast<uint32_t>(k2 >> 32)), L3 = __uint_as_float(static_cast<uint32_t>(k3 >> 32));
    uint32_t res_d = 0x7F800000u, res_j = 0xFFFFFFFFu;  // d2 as bits: exact distances are >= +0, so unsigned order is float order
    // Exact (d2, original index)-minimum of one granule for each of kN points (held by lanes s[k]); results are
    // wave-uniform.  All loads of the step are issued before the first distance is computed, so the kN chains share
    // one memory round trip.
    auto scan_granules = [&](auto n_tag, const int* s, uint32_t granule_of_lane, uint32_t* ud, uint32_t* uj) {
      constexpr int kN = decltype(n_tag)::value;
      constexpr int kIt = kRun / 64;
      float ax[kN], ay[kN], az[kN], X[kN][kIt], Y[kN][kIt], Z[kN][kIt];
      uint32_t Pm[kN][kIt];
#pragma unroll
      for (int k = 0; k < kN; ++k) {
        ax[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qx), s[k]));
        ay[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qy), s[k]));
        az[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qz), s[k]));
        const uint32_t tbase = J.tgt_off + __builtin_amdgcn_readlane(granule_of_lane, s[k]) * kRun + lane;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
          X[k][it] = sox[tbase + it * 64];
          Y[k][it] = soy[tbase + it * 64];
          Z[k][it] = soz[tbase + it * 64];
          Pm[k][it] = perm[tbase + it * 64];
        }
      }
      uint32_t mdb[kN], mjv[kN];
#pragma unroll
      for (int k = 0; k < kN; ++k) {
        float md = __builtin_inff();
        uint32_t mj = 0xFFFFFFFFu;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
          const float d = dist2(ax[k], ay[k], az[k], X[k][it], Y[k][it], Z[k][it]);
          if (d < md || (d == md && Pm[k][it] < mj)) {
            md = d;
            mj = Pm[k][it];
          }
        }
        mdb[k] = __float_as_uint(md);  // md is never NaN (d < md is false for NaN) nor negative: unsigned order is float order
        mjv[k] = mj;
      }
      if constexpr (kN == 8) {  // all eight points' minima in one transposing reduction each (distance, then index among its ties)
        const uint32_t rd = wave_min8(mdb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          ud[k] = wave_min8_of(rd, k);
          mjv[k] = mdb[k] == ud[k] ? mjv[k] : 0xFFFFFFFFu;
        }
        const uint32_t rj = wave_min8(mjv);
#pragma unroll
        for (int k = 0; k < 8; ++k) uj[k] = wave_min8_of(rj, k);
      } else {
#pragma unroll
        for (int k = 0; k < kN; ++k) {
          ud[k] = wave_min_u32(mdb[k]);
          uj[k] = wave_min_u32(mdb[k] == ud[k] ? mjv[k] : 0xFFFFFFFFu);
        }
      }
    };
    // Phase A: granule 1 of every point, kStep points per step (one at a time, a wave's 64 points are 64 dependent
    // memory round trips: 0.15 ms at any size).  A step with fewer than kStep points left repeats its first one.
    // kStep = 8 for the 128-target granule of the matrix-core filter: 8 x 2 x 4 loads in flight per lane, 8 round trips
    // per wave (6 x 20k points against 100k targets: 48 -> ~30 us; the 64-target granule keeps 4 x 1 x 4).
    // (16 points per step did not help in round 2.  Fewer points per wave on more waves - kWavePoints - does: round 2 tried it
    // while the key-row chain above still dominated the kernel and saw nothing; round 3's phase stamps showed 2.2 us per step.)
    constexpr int kStep = kRun >= 128 ? 8 : 4;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    P2P_DIAG_STAMP_WAIT();  // 1: key rows merged, point transformed
    while (todo) {
      int s[kStep];
      bool real[kStep];
#pragma unroll
      for (int k = 0; k < kStep; ++k) {
        real[k] = todo != 0;
        s[k] = real[k] ? __builtin_ctzll(todo) : s[0];
        todo &= todo - (todo != 0);
      }
      uint32_t ud[kStep], uj[kStep];
      scan_granules(std::integral_constant<int, kStep>{}, s, g1, ud, uj);
#pragma unroll
      for (int k = 0; k < kStep; ++k)
        if (real[k] && lane == static_cast<uint32_t>(s[k])) {
          res_d = ud[k];
          res_j = uj[k];
        }
    }
    // Phase B: the points whose second granule could still hold the minimum or a tie (a few per cent)
    todo = __builtin_amdgcn_ballot_w64(valid && L2 <= __uint_as_float(res_d));
    P2P_DIAG_STAMP();  // 2: first granules
#ifdef MI355_ICP_DIAG
    const unsigned long long diag_second = __builtin_popcountll(todo);
#endif
    while (todo) {
      const int s = __builtin_ctzll(todo);
      todo &= todo - 1;
      uint32_t ud, uj;
      scan_granules(std::integral_constant<int, 1>{}, &s, g2, &ud, &uj);
      if (lane == static_cast<uint32_t>(s) && (ud < res_d || (ud == res_d && uj < res_j))) {
        res_d = ud;
        res_j = uj;
      }
    }
    P2P_DIAG_STAMP();  // 3: second granules
    P2P_DIAG_END(4ull | (static_cast<unsigned long long>(J.src_n) << 8) | (diag_second << 40) | (static_cast<unsigned long long>(n_rows) << 48));
    if (!valid) return;
    // `local` is a position in the (possibly Morton-ordered) source planes the scan walked; answers go to the point's own slot
    const uint32_t orig = src_perm ? src_perm[J.src_off + local] : local;
    if (L3 <= __uint_as_float(res_d) || res_j == 0xFFFFFFFFu) {  // a third candidate granule (or more), or nothing finite found
      const uint32_t pos = atomicAdd(&overflow_count[blockIdx.y], 1u);
      overflow_list[J.out_off + pos] = orig;
      // the best pair found so far - an upper bound the overflow pass (overflow_tiles_kernel) prunes with; index -1 = none.
      // Every listed point is rewritten by that pass, or by the host path, before anything reads these slots.
      out_index[J.out_off + orig] = static_cast<int32_t>(res_j);
      out_dist2[J.out_off + orig] = __uint_as_float(res_d);
      return;
    }
    out_index[J.out_off + orig] = static_cast<int32_t>(res_j);
    out_dist2[J.out_off + orig] = __uint_as_float(res_d);
  }
}

// The points the filter paths could not settle (three or more candidate granules): one workgroup per listed point scans
// EVERY target with the contract formula and reduces (d2, index) lexicographically - the all-exact answer, without the
// host having to read the count first.  grid.x is a fixed cap; workgroups past the job's count leave at once, and a
// count above the cap is handled by the host path (run_overflow_fallback) after the pass's single synchronisation.
__global__ __launch_bounds__(kTile) void overflow_exact_kernel(
    const JobDesc* __restrict__ jobs, const float* __restrict__ sx, const float* __restrict__ sy,
    const float* __restrict__ sz, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, const uint32_t* __restrict__ overflow_list,
    const uint32_t* __restrict__ overflow_count, int32_t* __restrict__ out_index, float* __restrict__ out_dist2,
    uint32_t* __restrict__ error_flag) {
  __shared__ uint32_t sd[kTile / 64], sj[kTile / 64];
  const JobDesc& J = jobs[blockIdx.y];
  if (blockIdx.x >= overflow_count[blockIdx.y]) return;
  const uint32_t local = overflow_list[J.out_off + blockIdx.x];
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  const uint32_t i = J.src_off + local;
  float qx, qy, qz;
  transform_point(pose, sx[i], sy[i], sz[i], qx, qy, qz);
  float best = __builtin_inff();
  uint32_t best_j = 0xFFFFFFFFu;
  // 1024 threads, one target of every tile each (padding targets sit at 3e38: their distance overflows to +inf); four
  // tiles in flight - the workgroup is alone on its CU when few points are listed, so memory latency has to be covered
  // from inside the thread
  const uint32_t n_tiles = J.tgt_tiles;
  for (uint32_t tile = 0; tile < n_tiles; tile += 4) {
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t t = J.tgt_off + (tile + k < n_tiles ? tile + k : tile) * kTile + threadIdx.x;
      d[k] = dist2(qx, qy, qz, tx[t], ty[t], tz[t]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t t = (tile + k) * kTile + threadIdx.x;  // increasing: a later equal distance never replaces an earlier one
      if (tile + k < n_tiles && (d[k] < best || (d[k] == best && t < best_j))) {
        best = d[k];
        best_j = t;
      }
    }
  }
  // best is never NaN (d < best and d == best are false for NaN) nor negative: unsigned order of the bits is float order
  const uint32_t wd = wave_min_u32(__float_as_uint(best));
  const uint32_t wj = wave_min_u32(__float_as_uint(best) == wd ? best_j : 0xFFFFFFFFu);
  if ((threadIdx.x & 63u) == 0) {
    sd[threadIdx.x >> 6] = wd;
    sj[threadIdx.x >> 6] = wj;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t d = sd[0], j = sj[0];
    for (int w = 1; w < kTile / 64; ++w)
      if (sd[w] < d || (sd[w] == d && sj[w] < j)) {
        d = sd[w];
        j = sj[w];
      }
    if (j == 0xFFFFFFFFu || j >= J.tgt_n) {  // nothing but NaN / padding: same report as the all-exact resolve
      atomicOr(error_flag, 1u);
      j = 0;
    }
    out_index[J.out_off + local] = static_cast<int32_t>(j);
    out_dist2[J.out_off + local] = __uint_as_float(d);
  }
}

// The same job for the paths that own a Morton-sorted copy of the targets (every expansion filter): one 256-thread
// workgroup per listed point scans, with the contract formula, the TILES THE POINT CAN STILL REACH - those whose bounding
// sphere (the per-tile table of xtarget_kernels.hip) may hold a target at or below d0, the best distance the resolve found
// in the granules it did scan (an upper bound of the answer; +inf if it found none: then every tile is scanned).  A tile
// out of reach holds neither the minimum nor a tie with it (sphere_lower_bound rounds towards "reachable"), so the
// (d2, original index)-lexicographic minimum over the reachable tiles is the all-exact answer.  Typically 2-10 of the
// 1024 tiles of a 10^6-point scene: the pass costs microseconds at any target size, so the overflow points never need
// the host (round 1 went through a host round trip + gathered all-exact scan above 262k targets).
// 1024 threads: one target of a tile per thread.  (With 256 threads a wave was alone on its SIMD and walked four targets per
// tile through ~100 dependent instructions each round: ~1 us per reachable tile whatever was in flight - phase stamps, round 3.)
constexpr int kOverflowThreads = kTile;
__global__ __launch_bounds__(kOverflowThreads) void overflow_tiles_kernel(
    const JobDesc* __restrict__ jobs, const float* __restrict__ sx, const float* __restrict__ sy,
    const float* __restrict__ sz, const float* __restrict__ sox, const float* __restrict__ soy,
    const float* __restrict__ soz, const uint32_t* __restrict__ tperm, const float4* __restrict__ ttable,
    const uint32_t* __restrict__ overflow_list, const uint32_t* __restrict__ overflow_count,
    int32_t* __restrict__ out_index, float* __restrict__ out_dist2, uint32_t* __restrict__ error_flag) {
  __shared__ unsigned long long smask[kOverflowThreads / 64];
  __shared__ uint32_t sd[kOverflowThreads / 64], sj[kOverflowThreads / 64];
  P2P_DIAG_BEGIN(blockIdx.x == 0 && threadIdx.x == 0);
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t n_listed = overflow_count[blockIdx.y];
  if (blockIdx.x >= n_listed) {
    P2P_DIAG_STAMP_WAIT();  // 1: descriptor and count read: nothing listed for this workgroup
    P2P_DIAG_END(5ull);
    return;
  }
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  // grid.x workgroups share the job's list (mostly empty: a launch of one workgroup per POSSIBLE entry spent 11 us finding that out)
  for (uint32_t slot = blockIdx.x; slot < n_listed; slot += gridDim.x) {
  const uint32_t local = overflow_list[J.out_off + slot];
  const uint32_t i = J.src_off + local;
  float qx, qy, qz;
  transform_point(pose, sx[i], sy[i], sz[i], qx, qy, qz);
  const float d0 = out_dist2[J.out_off + local];            // never NaN (resolve_xfilter_kernel)
  float best = d0;
  uint32_t best_j = static_cast<uint32_t>(out_index[J.out_off + local]);  // 0xFFFFFFFF = none yet
  const uint32_t n_tiles = J.tgt_tiles, tile_base = J.tgt_off / kTile, tid = threadIdx.x;
  P2P_DIAG_STAMP_WAIT();  // 1: list entry, point, its upper bound
#ifdef MI355_ICP_DIAG
  unsigned long long diag_tiles = 0;
#endif
  for (uint32_t t0 = 0; t0 < n_tiles; t0 += kOverflowThreads) {
    const uint32_t t = t0 + tid;
    const bool reach = t < n_tiles && sphere_reachable(qx, qy, qz, 0.0f, ttable[tile_base + t], d0);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(reach);
    if ((tid & 63u) == 0) smask[tid >> 6] = m;
    __syncthreads();
    P2P_DIAG_STAMP();  // tile table tested
    for (uint32_t w = 0; w < kOverflowThreads / 64; ++w) {
      unsigned long long todo = smask[w];
#ifdef MI355_ICP_DIAG
      diag_tiles += __builtin_popcountll(todo);
#endif
      // (several tiles' loads in flight at once change nothing here either: 4.5 -> 4.1 us for ten tiles)
      while (todo) {
        const uint32_t tile = t0 + w * 64 + static_cast<uint32_t>(__builtin_ctzll(todo));
        todo &= todo - 1;
        const uint32_t p = J.tgt_off + tile * kTile + tid;
        const float d = dist2(qx, qy, qz, sox[p], soy[p], soz[p]);
        const uint32_t pj = tperm[p];
        const bool real = tile * kTile + tid < J.tgt_n;  // padding slots sit at 3e38 with index 0
        if (real && (d < best || (d == best && pj < best_j))) {
          best = d;
          best_j = pj;
        }
      }
    }
    __syncthreads();  // smask is rewritten by the next round
    P2P_DIAG_STAMP_WAIT();  // reachable tiles scanned
  }
  const uint32_t wd = wave_min_u32(__float_as_uint(best));
  const uint32_t wj = wave_min_u32(__float_as_uint(best) == wd ? best_j : 0xFFFFFFFFu);
  if ((tid & 63u) == 0) {
    sd[tid >> 6] = wd;
    sj[tid >> 6] = wj;
  }
  __syncthreads();
  if (tid < 64) {  // the sixteen waves' results, one per lane: the same two wave minima once more
    const bool has = tid < kOverflowThreads / 64;
    const uint32_t md = has ? sd[tid] : 0xFFFFFFFFu, mj = has ? sj[tid] : 0xFFFFFFFFu;
    uint32_t d = wave_min_u32(md);
    uint32_t j = wave_min_u32(md == d ? mj : 0xFFFFFFFFu);
    if (tid == 0) {
      if (j == 0xFFFFFFFFu || j >= J.tgt_n) {  // nothing but NaN / +inf distances: the oracle's report for "nothing below +inf"
        atomicOr(error_flag, 1u);
        j = 0;
        d = 0x7F800000u;
      }
      out_index[J.out_off + local] = static_cast<int32_t>(j);
      out_dist2[J.out_off + local] = __uint_as_float(d);
    }
  }
  __syncthreads();  // sd / sj are rewritten by the next listed point
  if (slot == blockIdx.x) P2P_DIAG_END(6ull | (static_cast<unsigned long long>(n_listed) << 8) | (diag_tiles << 40) | (static_cast<unsigned long long>(n_tiles) << 48));
  }
}

// =====================================================================================================================
// OPTIONAL exact nearest neighbour WITH pruning (MI355_ICP_SCAN_MODE=pruned; never the default: north_star's metric is
// brute force, and bench.py measures that).  Same results - the (d2, original index)-lexicographic minimum of the
// contract formula over ALL targets - from a fraction of the distance evaluations:
//   * sources and targets both have a Morton-sorted copy (xtarget_kernels.hip) with a bounding sphere per 1024-point
//     tile and per 64-point run; a workgroup takes 256 consecutive SORTED source points - a rigid pose keeps them
//     spatially coherent - transforms them with the contract transform, cuts the run at its largest gap and bounds each
//     of the two parts by a sphere (cq, rq);
//   * every tile holds a point within (|cq - c| + rq + r) of every query: the smallest such bound picks the first tile,
//     which is scanned exhaustively; afterwards each lane's best distance is exact and an upper bound of its minimum;
//   * a tile (or 64-point run) whose nearest possible point is farther than a lane's best can hold neither its minimum
//     nor a tie with it, so: tiles are listed once against the workgroup's largest best (dk), visited nearest first
//     through LDS with dk refreshed after every visit (the walk stops at the first tile out of reach), and each wave
//     scans only the runs one of its lanes can still reach.
// All bound arithmetic rounds towards "reachable" (sphere_lower_bound, icp_internal.h).
__global__ __launch_bounds__(kBlockThreads) void nn_pruned_kernel(
    const JobDesc* __restrict__ jobs, const float* __restrict__ ssx, const float* __restrict__ ssy,
    const float* __restrict__ ssz, const uint32_t* __restrict__ sperm, const float* __restrict__ tsx,
    const float* __restrict__ tsy, const float* __restrict__ tsz, const uint32_t* __restrict__ tperm,
    const float4* __restrict__ ttable, const float4* __restrict__ tsub, int32_t* __restrict__ out_index,
    float* __restrict__ out_dist2, uint32_t* __restrict__ error_flag) {
  constexpr uint32_t kMaxCand = 1024;
  __shared__ float4 lds[4][kTile / 4];  // x, y, z (sorted order) and original indices of one target tile
  __shared__ float4 lds_sphere[kTile / 64 + 1];  // ... the bounding spheres of its 16 runs and its own
  __shared__ unsigned long long cand[kMaxCand], sorted[kMaxCand];  // (lower-bound bits << 32 | tile)
  __shared__ uint32_t n_cand;
  __shared__ uint32_t red_a[2 * (kBlockThreads / 64)];
  __shared__ float red[7][kBlockThreads / 64];
  __shared__ unsigned long long first_key[kBlockThreads / 64], seed_key[2][kBlockThreads / 64];
  const JobDesc& J = jobs[blockIdx.y];
  if (blockIdx.x * kBlockThreads >= J.src_n) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t pos = blockIdx.x * kBlockThreads + tid;
  const uint32_t pc = J.src_off + (pos < J.src_n ? pos : J.src_n - 1);  // out-of-range lanes shadow the last point
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  float qx, qy, qz;
  transform_point(pose, ssx[pc], ssy[pc], ssz[pc], qx, qy, qz);
  const uint32_t n_tiles = J.tgt_tiles, tile_base = J.tgt_off / kTile;
  float best = __builtin_inff();
  uint32_t best_j = 0xFFFFFFFFu;
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(tsx + J.tgt_off);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(tsy + J.tgt_off);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(tsz + J.tgt_off);
  const float4* __restrict__ gp = reinterpret_cast<const float4*>(tperm + J.tgt_off);
  // stage one target tile through LDS; every wave scans the 64-point runs one of its lanes can still reach
  auto scan_tile = [&](uint32_t tile) {
    const uint32_t g0 = tile * (kTile / 4) + tid;
    lds[0][tid] = gx[g0];
    lds[1][tid] = gy[g0];
    lds[2][tid] = gz[g0];
    lds[3][tid] = gp[g0];
    // the spheres travel with the tile: fetched one by one from global memory by every wave, their latency was most of
    // a visit's cost
    if (tid < kTile / 64) lds_sphere[tid] = tsub[(tile_base + tile) * (kTile / 64) + tid];
    if (tid == kTile / 64) lds_sphere[tid] = ttable[tile_base + tile];
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(sphere_reachable(qx, qy, qz, 0.0f, lds_sphere[kTile / 64], best)) != 0) {
      for (uint32_t sub = 0; sub < kTile / 64; ++sub) {
        const bool reach = sphere_reachable(qx, qy, qz, 0.0f, lds_sphere[sub], best);
        if (__builtin_amdgcn_ballot_w64(reach) == 0) continue;
#pragma unroll 4
        for (uint32_t g = sub * 16; g < sub * 16 + 16; ++g) {
          const float4 X = lds[0][g], Y = lds[1][g], Z = lds[2][g], P = lds[3][g];
          const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
          const uint32_t js[4] = {__float_as_uint(P.x), __float_as_uint(P.y), __float_as_uint(P.z), __float_as_uint(P.w)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = dist2(qx, qy, qz, xs[e], ys[e], zs[e]);
            if (d < best || (d == best && js[e] < best_j)) {
              best = d;
              best_j = js[e];
            }
          }
        }
      }
    }
    __syncthreads();
  };
  // The run of 256 sorted points is cut in two at its largest gap between consecutive points: a Morton jump, if the run
  // straddles one, is that gap, so each part stays spatially compact (one sphere over a run with a jump in it reaches
  // every tile of the cloud: 58 of 4096 workgroups at 1M<->1M, and they set the kernel's duration).  Without a jump the
  // cut is merely somewhere, and two smaller spheres bound better than one.
  float* ex = reinterpret_cast<float*>(&lds[0][0]);  // the tile buffer is free until the first scan
  ex[tid] = qx;
  ex[kBlockThreads + tid] = qy;
  ex[2 * kBlockThreads + tid] = qz;
  __syncthreads();
  unsigned long long gk = 0ull;  // (gap bits << 32 | position), the largest wins; position 0 has no predecessor
  if (tid > 0) {
    const float g = dist2(qx, qy, qz, ex[tid - 1], ex[kBlockThreads + tid - 1], ex[2 * kBlockThreads + tid - 1]);
    gk = (static_cast<unsigned long long>(g >= 0.0f ? __float_as_uint(g) : 0u) << 32) | tid;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(gk, off, 64);
    gk = o > gk ? o : gk;
  }
  if (lane == 0) first_key[wave] = gk;
  if (tid == 0) n_cand = 0;
  __syncthreads();
  gk = first_key[0];
#pragma unroll
  for (int w = 1; w < kBlockThreads / 64; ++w) gk = first_key[w] > gk ? first_key[w] : gk;
  const uint32_t cut = static_cast<uint32_t>(gk) ? static_cast<uint32_t>(gk) : kBlockThreads;  // no gap at all: one part
  const int part = tid >= cut ? 1 : 0;
  // bounding sphere of each part (centre of its box, largest distance of a member from it, rounded up)
  float cq[2][3], rq[2];
  bool has[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const bool in = part == g;
    const float ninf = -__builtin_inff();
    float v[6] = {in ? qx : ninf, in ? qy : ninf, in ? qz : ninf, in ? -qx : ninf, in ? -qy : ninf, in ? -qz : ninf};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v[k] = __builtin_fmaxf(v[k], __shfl_xor(v[k], off, 64));
      if (lane == 0) red[k][wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      v[k] = red[k][0];
#pragma unroll
      for (int w = 1; w < kBlockThreads / 64; ++w) v[k] = __builtin_fmaxf(v[k], red[k][w]);
    }
    has[g] = v[0] > ninf;  // part 1 is empty when there is no cut (uniform)
    cq[g][0] = has[g] ? 0.5f * v[0] - 0.5f * v[3] : 0.0f;
    cq[g][1] = has[g] ? 0.5f * v[1] - 0.5f * v[4] : 0.0f;
    cq[g][2] = has[g] ? 0.5f * v[2] - 0.5f * v[5] : 0.0f;
    float r2 = in ? dist2(qx, qy, qz, cq[g][0], cq[g][1], cq[g][2]) : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r2 = __builtin_fmaxf(r2, __shfl_xor(r2, off, 64));
    if (lane == 0) red[6][wave] = r2;
    __syncthreads();
    r2 = red[6][0];
#pragma unroll
    for (int w = 1; w < kBlockThreads / 64; ++w) r2 = __builtin_fmaxf(r2, red[6][w]);
    rq[g] = __builtin_amdgcn_sqrtf(r2) * 1.000003815f;
    __syncthreads();  // red[] is reused by the next part
  }
  // dk of each part: the largest best distance over its lanes (best is never NaN nor negative: unsigned order of the bits is
  // float order)
  uint32_t dk[2];
  auto refresh_dk = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const uint32_t mine = part == g ? __float_as_uint(best) : 0u;
      const uint32_t wk = ~wave_min_u32(~mine);
      if (lane == 0) red_a[g * (kBlockThreads / 64) + wave] = wk;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      uint32_t r = red_a[g * (kBlockThreads / 64)];
#pragma unroll
      for (int w = 1; w < kBlockThreads / 64; ++w) r = red_a[g * (kBlockThreads / 64) + w] > r ? red_a[g * (kBlockThreads / 64) + w] : r;
      dk[g] = r;
    }
    __syncthreads();
  };
  // first tiles: per part, the tile whose farthest possible point is nearest (any tile would do for correctness)
  unsigned long long fk[2] = {~0ull, ~0ull};
  for (uint32_t t = tid; t < n_tiles; t += kBlockThreads) {
    const float4 c = ttable[tile_base + t];
    const float rt = __builtin_amdgcn_sqrtf(c.w);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float u = __builtin_amdgcn_sqrtf(dist2(cq[g][0], cq[g][1], cq[g][2], c.x, c.y, c.z)) + rq[g] + rt;
      const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(u)) << 32) | t;  // u >= 0 or NaN (sorts last)
      fk[g] = key < fk[g] ? key : fk[g];
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(fk[g], off, 64);
      fk[g] = o < fk[g] ? o : fk[g];
    }
    if (lane == 0) seed_key[g][wave] = fk[g];
  }
  __syncthreads();
  uint32_t first[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    unsigned long long k = seed_key[g][0];
#pragma unroll
    for (int w = 1; w < kBlockThreads / 64; ++w) k = seed_key[g][w] < k ? seed_key[g][w] : k;
    first[g] = static_cast<uint32_t>(k);
  }
  if (!has[1]) first[1] = first[0];
  scan_tile(first[0]);
  if (first[1] != first[0]) scan_tile(first[1]);
  refresh_dk();
  // every other tile within reach of a part, nearest first
  for (uint32_t t = tid; t < n_tiles; t += kBlockThreads) {
    if (t == first[0] || t == first[1]) continue;
    const float4 c = ttable[tile_base + t];
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float lb = sphere_lower_bound(cq[g][0], cq[g][1], cq[g][2], rq[g], c);
      const uint32_t lbb = lb >= 0.0f ? __float_as_uint(lb) : 0u;  // NaN: nothing known, treat as 0
      if (has[g] && lbb <= dk[g] && lbb < key) key = lbb;
    }
    if (key != 0xFFFFFFFFu) {
      const uint32_t slot = atomicAdd(&n_cand, 1u);
      if (slot < kMaxCand) cand[slot] = (static_cast<unsigned long long>(key) << 32) | t;
    }
  }
  __syncthreads();
  const uint32_t nc = n_cand;
  if (nc > kMaxCand) {  // more candidates than the list holds: visit every tile
    for (uint32_t t = 0; t < n_tiles; ++t)
      if (t != first[0] && t != first[1]) scan_tile(t);
  } else {
    for (uint32_t i = tid; i < nc; i += kBlockThreads) {  // rank sort (keys are distinct: the tile id is part of them)
      const unsigned long long key = cand[i];
      uint32_t rank = 0;
      for (uint32_t k = 0; k < nc; ++k) rank += cand[k] < key ? 1u : 0u;
      sorted[rank] = key;
    }
    __syncthreads();
    for (uint32_t k = 0; k < nc; ++k) {
      const unsigned long long key = sorted[k];
      const uint32_t reach = dk[0] > dk[1] ? dk[0] : dk[1];
      if (static_cast<uint32_t>(key >> 32) > reach) break;  // this tile and all later ones are out of reach (uniform)
      scan_tile(static_cast<uint32_t>(key));
      refresh_dk();
    }
  }
  if (pos < J.src_n) {
    const uint32_t orig = sperm[J.src_off + pos];
    if (best_j == 0xFFFFFFFFu) {  // nothing but NaN distances: same report as the all-exact resolve
      atomicOr(error_flag, 1u);
      best_j = 0;
    }
    out_index[J.out_off + orig] = static_cast<int32_t>(best_j);
    out_dist2[J.out_off + orig] = best;
  }
}

// N2 for the fused path: accumulation over ALL points after every index is known (fixed order: same bits every run)
__global__ __launch_bounds__(kBlockThreads) void accumulate_kernel(const JobDesc* __restrict__ jobs,
                                                                    const float* __restrict__ sx, const float* __restrict__ sy,
                                                                    const float* __restrict__ sz, const float* __restrict__ tx,
                                                                    const float* __restrict__ ty, const float* __restrict__ tz,
                                                                    const int32_t* __restrict__ index,
                                                                    const float* __restrict__ dist2,
                                                                    double* partials, PassTail tail) {
  __shared__ double scratch[kBlockThreads / 64][MI355_ICP_NSUMS];
  __shared__ uint32_t last_flag;
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t sb = blockIdx.x;
  if (sb >= J.src_blocks) return;
  // raised by earlier launches of the pass only (the resolve and overflow kernels): read now, needed at the very end
  const uint32_t raised = tail.done ? __hip_atomic_load(tail.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const float max_d2 = J.max_d2;
  double acc[MI355_ICP_NSUMS];
#pragma unroll
  for (int k = 0; k < MI355_ICP_NSUMS; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int p = 0; p < kResolvePointsPerLane; ++p) {
    const uint32_t local = sb * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    if (local < J.src_n) {
      const float best = dist2[J.out_off + local];
      if (best <= max_d2) {
        const uint32_t i = J.src_off + local, j = J.tgt_off + static_cast<uint32_t>(index[J.out_off + local]);
        const double a0 = sx[i], a1 = sy[i], a2 = sz[i], b0 = tx[j], b1 = ty[j], b2 = tz[j];
        acc[0] += 1.0;
        acc[1] += a0; acc[2] += a1; acc[3] += a2;
        acc[4] += b0; acc[5] += b1; acc[6] += b2;
        acc[7] += a0 * b0; acc[8] += a0 * b1; acc[9] += a0 * b2;
        acc[10] += a1 * b0; acc[11] += a1 * b1; acc[12] += a1 * b2;
        acc[13] += a2 * b0; acc[14] += a2 * b1; acc[15] += a2 * b2;
        acc[16] += static_cast<double>(best);
      }
    }
  }
  const double r = pass_block_sum<MI355_ICP_NSUMS>(acc, scratch);
  double* job_rows = partials + static_cast<size_t>(J.partial_off) * MI355_ICP_NSUMS;
  if (!tail.done) {
    if (threadIdx.x < MI355_ICP_NSUMS) job_rows[static_cast<size_t>(sb) * MI355_ICP_NSUMS + threadIdx.x] = r;
    return;
  }
  finish_pass<MI355_ICP_NSUMS>(r, acc, scratch, &last_flag, job_rows, sb, J.src_blocks, tail, raised,
                               [&](double* t, uint32_t job, uint32_t n_jobs) { publish_overflow(tail, job, t + job, t + n_jobs + job); });
}

// overflow handling: gather the listed source points into scratch planes / scatter their exact answers back
__global__ __launch_bounds__(kBlockThreads) void gather_points_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                                       const float* __restrict__ sz, uint32_t src_off,
                                                                       const uint32_t* __restrict__ list, uint32_t n,
                                                                       float* __restrict__ out, size_t out_stride,
                                                                       uint32_t n_padded) {
  const uint32_t k = blockIdx.x * kBlockThreads + threadIdx.x;
  if (k >= n_padded) return;
  float x = 0.f, y = 0.f, z = 0.f;
  if (k < n) {
    const uint32_t i = src_off + list[k];
    x = sx[i]; y = sy[i]; z = sz[i];
  }
  out[k] = x;
  out[out_stride + k] = y;
  out[2 * out_stride + k] = z;
}

__global__ __launch_bounds__(kBlockThreads) void scatter_results_kernel(const uint32_t* __restrict__ list, uint32_t n,
                                                                         const int32_t* __restrict__ idx_tmp,
                                                                         const float* __restrict__ d2_tmp,
                                                                         int32_t* __restrict__ out_index,
                                                                         float* __restrict__ out_dist2, uint32_t out_off) {
  const uint32_t k = blockIdx.x * kBlockThreads + threadIdx.x;
  if (k >= n) return;
  out_index[out_off + list[k]] = idx_tmp[k];
  out_dist2[out_off + list[k]] = d2_tmp[k];
}

// ---------------------------------------------------------------------------------------------
// N2 final reduction: one workgroup per job sums that job's per-block partials in a fixed order
// (thread t takes blocks t, t+256, ... ascending; then the pass_block_sum tree), so a given cloud,
// pose and slab always reduces to the same bits.
// Tail of `sums` behind the n = gridDim.x rows: [n flags][n counts] - this pass's overflow count of every job (what the
// host books after the pass's single synchronisation) and whether it exceeded what the device-side overflow pass covers
// (slab-sharded runs add the flags up across ranks with the sums).  The counter is reset here for the next pass: every
// kernel that adds to it or reads it in this pass has run.
__global__ __launch_bounds__(kBlockThreads) void final_reduce_kernel(const JobDesc* __restrict__ jobs,
                                                                      const double* __restrict__ partials,
                                                                      double* __restrict__ sums, uint32_t* __restrict__ overflow_count,
                                                                      uint32_t overflow_cap) {
  __shared__ double scratch[kBlockThreads / 64][MI355_ICP_NSUMS];
  const JobDesc& J = jobs[blockIdx.x];
  double acc[MI355_ICP_NSUMS];
#pragma unroll
  for (int k = 0; k < MI355_ICP_NSUMS; ++k) acc[k] = 0.0;
  for (uint32_t b = threadIdx.x; b < J.src_blocks; b += kBlockThreads) {
    const double* __restrict__ row = partials + static_cast<size_t>(J.partial_off + b) * MI355_ICP_NSUMS;
#pragma unroll
    for (int k = 0; k < MI355_ICP_NSUMS; ++k) acc[k] += row[k];
  }
  const double r = pass_block_sum<MI355_ICP_NSUMS>(acc, scratch);
  if (threadIdx.x < MI355_ICP_NSUMS) sums[static_cast<size_t>(blockIdx.x) * MI355_ICP_NSUMS + threadIdx.x] = r;
  if (threadIdx.x == 0) {
    double* tail = sums + static_cast<size_t>(gridDim.x) * MI355_ICP_NSUMS;
    const uint32_t n = overflow_count ? overflow_count[blockIdx.x] : 0u;
    tail[blockIdx.x] = n > overflow_cap ? 1.0 : 0.0;
    tail[gridDim.x + blockIdx.x] = static_cast<double>(n);
    if (overflow_count) overflow_count[blockIdx.x] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------
void launch_nn_scan(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                    DevicePlanes src, DevicePlanes tgt, unsigned long long* keys, uint32_t key_stride, const StageArgs* staged,
                    JobDesc* jobs_out, RefJobDesc* rjobs_out) {
  dim3 grid(max_blocks_x, static_cast<uint32_t>(n_jobs));
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const StageArgs not_staged{};  // n = 0
  const StageArgs& sa = staged ? *staged : not_staged;
  if (points_per_lane >= 100) {  // experimental scalar-load variant: 100 + P
    if (points_per_lane == 108)
      hipLaunchKernelGGL(nn_scan_sgpr_kernel<8>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride);
    else if (points_per_lane == 104)
      hipLaunchKernelGGL(nn_scan_sgpr_kernel<4>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride);
    else
      hipLaunchKernelGGL(nn_scan_sgpr_kernel<2>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride);
    return;
  }
  switch (points_per_lane) {
    case 2:
      hipLaunchKernelGGL(nn_scan_kernel<2>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride, sa, jobs_out, rjobs_out);
      break;
    case 8:
      hipLaunchKernelGGL(nn_scan_kernel<8>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride, sa, jobs_out, rjobs_out);
      break;
    default:
      hipLaunchKernelGGL(nn_scan_kernel<4>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys, key_stride, sa, jobs_out, rjobs_out);
      break;
  }
}

// partials == nullptr: indices and distances only (reference mode), and merged key rows are handed back cleared
void launch_resolve(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                    DevicePlanes tgt, unsigned long long* keys, uint32_t key_stride, int32_t* out_index,
                    float* out_dist2, double* partials, uint32_t* error_flag, const PassTail& tail) {
  if (partials) {
    dim3 grid(max_src_blocks, static_cast<uint32_t>(n_jobs));
    hipLaunchKernelGGL(resolve_accumulate_kernel<true>, grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream),
                       jobs_dev, src.x, src.y, src.z, tgt.x, tgt.y, tgt.z, keys, key_stride, out_index, out_dist2,
                       partials, error_flag, tail);
  } else {
    // one wave per workgroup: small launches (the coarse pyramid levels) spread over four times the CUs
    constexpr uint32_t kThreads = 64;
    dim3 grid(max_src_blocks * kResolvePointsPerLane * (kBlockThreads / kThreads), static_cast<uint32_t>(n_jobs));
    hipLaunchKernelGGL(resolve_accumulate_kernel<false>, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       jobs_dev, src.x, src.y, src.z, tgt.x, tgt.y, tgt.z, keys, key_stride, out_index, out_dist2,
                       partials, error_flag, PassTail{});
  }
}

#ifdef MI355_ICP_DIAG
extern "C" int mi355_icp_diag_begin(size_t n_slots) {
  unsigned long long* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), n_slots * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemset(buf, 0, n_slots * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_p2p_begin(void) {
  unsigned long long* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), 4096 * 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemset(buf, 0, 4096 * 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  const unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_p2p_n), &zero, sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_p2p), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_p2p_read(unsigned long long* out, unsigned int* n) {
  unsigned long long* buf = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_diag_p2p), sizeof(buf)) != hipSuccess || !buf) return 1;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_diag_p2p_n), sizeof(*n)) != hipSuccess) return 1;
  return hipMemcpy(out, buf, 4096 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_read(unsigned long long* out, size_t n_slots) {
  unsigned long long* buf = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_diag_stamps), sizeof(buf)) != hipSuccess || !buf) return 1;
  return hipMemcpy(out, buf, n_slots * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

void launch_resolve_fused(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                          DevicePlanes tgt, const unsigned long long* keys3, uint32_t key_stride, uint32_t rank_stride,
                          int32_t* out_index, float* out_dist2, double* partials, uint32_t* overflow_list,
                          uint32_t* overflow_count, uint32_t* error_flag) {
  (void)partials;
  dim3 grid(max_src_blocks, static_cast<uint32_t>(n_jobs));
  hipLaunchKernelGGL(resolve_fused_kernel, grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), jobs_dev, src.x,
                     src.y, src.z, tgt.x, tgt.y, tgt.z, keys3, key_stride, rank_stride, out_index, out_dist2, overflow_list,
                     overflow_count, error_flag);
}

void launch_resolve_xfilter(void* stream, int run_len, uint32_t split_mul, int wave_points, const JobDesc* jobs_dev, int n_jobs,
                            uint32_t max_src_blocks, DevicePlanes src, const uint32_t* src_perm, DevicePlanes sorted_original, const uint32_t* perm,
                            const unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride, int32_t* out_index,
                            float* out_dist2, uint32_t* overflow_list, uint32_t* overflow_count, uint32_t* error_flag) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t groups = max_src_blocks * kResolvePointsPerLane;  // 256-point groups of the largest job
  auto go = [&](auto kernel, uint32_t per_group) {
    hipLaunchKernelGGL(kernel, dim3(groups * per_group, static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0, st, jobs_dev, split_mul, src.x,
                       src.y, src.z, src_perm, sorted_original.x, sorted_original.y, sorted_original.z, perm, keys4, key_stride, rank_stride,
                       out_index, out_dist2, overflow_list, overflow_count);
  };
  if (run_len == kXbRun) {
    if (wave_points == 16) go(resolve_xfilter_kernel<kXbRun, 16>, 4);
    else if (wave_points == 32) go(resolve_xfilter_kernel<kXbRun, 32>, 2);
    else go(resolve_xfilter_kernel<kXbRun, 64>, 1);
  } else {
    go(resolve_xfilter_kernel<kFusedChunk, 64>, 1);
  }
  (void)error_flag;  // a point with no finite distance goes to the all-exact scan, which raises the flag
}

void launch_overflow_exact(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t cap, DevicePlanes src, DevicePlanes tgt,
                           const uint32_t* overflow_list, const uint32_t* overflow_count, int32_t* out_index, float* out_dist2,
                           uint32_t* error_flag) {
  hipLaunchKernelGGL(overflow_exact_kernel, dim3(cap, static_cast<uint32_t>(n_jobs)), dim3(kTile), 0,
                     static_cast<hipStream_t>(stream), jobs_dev, src.x, src.y, src.z, tgt.x, tgt.y, tgt.z, overflow_list, overflow_count,
                     out_index, out_dist2, error_flag);
}

void launch_overflow_tiles(void* stream, const JobDesc* jobs_dev, int n_jobs, DevicePlanes src, DevicePlanes sorted_tgt,
                           const uint32_t* tgt_perm, const void* tgt_table, const uint32_t* overflow_list, const uint32_t* overflow_count,
                           int32_t* out_index, float* out_dist2, uint32_t* error_flag) {
  // every listed point of a job is settled here, however many: the job's workgroups stride over its list (two 1024-thread
  // workgroups fit a CU; an empty list costs the launch and one count read per workgroup)
  const uint32_t per_job = std::max<uint32_t>(16u, std::min<uint32_t>(512u, 2048u / static_cast<uint32_t>(std::max(n_jobs, 1))));
  hipLaunchKernelGGL(overflow_tiles_kernel, dim3(per_job, static_cast<uint32_t>(n_jobs)), dim3(kOverflowThreads), 0, static_cast<hipStream_t>(stream),
                     jobs_dev, src.x, src.y, src.z, sorted_tgt.x, sorted_tgt.y, sorted_tgt.z, tgt_perm, static_cast<const float4*>(tgt_table),
                     overflow_list, overflow_count, out_index, out_dist2, error_flag);
}

void launch_nn_pruned(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes sorted_src,
                      const uint32_t* src_perm, DevicePlanes sorted_tgt, const uint32_t* tgt_perm, const void* tgt_table,
                      const void* tgt_sub, const void* tgt_super, int32_t* out_index, float* out_dist2, uint32_t* error_flag) {
  if (!max_src_blocks) return;
  dim3 grid(max_src_blocks * kResolvePointsPerLane, static_cast<uint32_t>(n_jobs));  // one workgroup per 256 sorted points
  hipLaunchKernelGGL(nn_pruned_kernel, grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), jobs_dev, sorted_src.x,
                     sorted_src.y, sorted_src.z, src_perm, sorted_tgt.x, sorted_tgt.y, sorted_tgt.z, tgt_perm,
                     static_cast<const float4*>(tgt_table), static_cast<const float4*>(tgt_sub), out_index, out_dist2, error_flag);
  (void)tgt_super;  // the k-NN scan's per-lane tile search uses it; here it measured slower than the workgroup sphere
}

void launch_accumulate(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_src_blocks, DevicePlanes src,
                       DevicePlanes tgt, const int32_t* index, const float* dist2, double* partials, const PassTail& tail) {
  dim3 grid(max_src_blocks, static_cast<uint32_t>(n_jobs));
  hipLaunchKernelGGL(accumulate_kernel, grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), jobs_dev, src.x, src.y,
                     src.z, tgt.x, tgt.y, tgt.z, index, dist2, partials, tail);
}

void launch_gather_points(void* stream, DevicePlanes src, uint32_t src_off, const uint32_t* list, uint32_t n, float* out_planes,
                          size_t out_stride, uint32_t n_padded) {
  if (!n_padded) return;
  hipLaunchKernelGGL(gather_points_kernel, dim3((n_padded + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), src.x, src.y, src.z, src_off, list, n, out_planes, out_stride, n_padded);
}

void launch_scatter_results(void* stream, const uint32_t* list, uint32_t n, const int32_t* idx_tmp, const float* d2_tmp,
                            int32_t* out_index, float* out_dist2, uint32_t out_off) {
  if (!n) return;
  hipLaunchKernelGGL(scatter_results_kernel, dim3((n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), list, n, idx_tmp, d2_tmp, out_index, out_dist2, out_off);
}

void launch_final_reduce(void* stream, const JobDesc* jobs_dev, int n_jobs, const double* partials, double* sums,
                         uint32_t* overflow_count, uint32_t overflow_cap) {
  hipLaunchKernelGGL(final_reduce_kernel, dim3(static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), jobs_dev, partials, sums, overflow_count, overflow_cap);
}

}  // namespace mi355

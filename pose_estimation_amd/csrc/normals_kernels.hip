// normals_kernels.hip — scene normals on the GPU: the step right before the ICP in the
// reference, cv::ppf_match_3d::computeNormalsPC3d(pc_scene, pc_scene_normals, 20, true, viewpoint)
// (pose_estimation/src/opencv_surface_match.cpp:57-59; third-party, restated in oracle/icp_oracle.c "Scene normals").
//
//   knn_scan_kernel<K>     exact brute-force K nearest neighbours of every scene point within the scene, the NN scan's
//                          machinery (SoA target tiles through LDS, wave-uniform ds_read_b128 broadcasts, unfused float32
//                          d2) with a sorted K-list per lane in registers; the target range is split across workgroups
//   knn_wave_kernel<K>     the default: the same lists with pruning on the Morton-sorted copy of the scene (bounding spheres
//                          per run / tile / 16 tiles), one query at a time per wave, its K-list spread over the lanes
//   normals_kernel<K>      per point: merge the per-split lists by (d2, index), gather the K neighbours, covariance as
//                          upstream's meanCovLocalPCInd (float32 products summed in double), cyclic Jacobi in double,
//                          smallest-eigenvalue eigenvector -> float32, flipped towards the viewpoint
// Arithmetic contract: neighbour sets identical to the oracle ((d2, index) order, so ties resolve the same way); the
// double-precision covariance / Jacobi follow the oracle's operation order (compile with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "icp_internal.h"

namespace mi355 {

namespace {

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float tx, float ty, float tz) {
  const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  float a = dx * dx;
  const float b = dy * dy, c = dz * dz;
  a = a + b;
  return a + c;
}

// Sorted insert into a register-resident list (ascending d; candidates arrive in ascending index, so a strict
// comparison keeps equal distances in index order).  Written as an independent select per slot - slot m takes the old
// slot m-1 if the newcomer sorts before it, the newcomer if it sorts before slot m only - so there is no dependency
// chain and every register index is a compile-time constant (one compare + four selects per slot).
template <int K>
__device__ __forceinline__ void list_insert(float (&d)[K], int32_t (&ix)[K], float nd, int32_t ni) {
  if (nd < d[K - 1]) {
    bool lt[K];
#pragma unroll
    for (int m = 0; m < K; ++m) lt[m] = nd < d[m];
#pragma unroll
    for (int m = K - 1; m > 0; --m) {
      const float keep_d = lt[m] ? nd : d[m];
      const int32_t keep_i = lt[m] ? ni : ix[m];
      d[m] = lt[m - 1] ? d[m - 1] : keep_d;
      ix[m] = lt[m - 1] ? ix[m - 1] : keep_i;
    }
    d[0] = lt[0] ? nd : d[0];
    ix[0] = lt[0] ? ni : ix[0];
  }
}

}  // namespace

// grid = (query_blocks * n_split), block = 256, one query per lane.  planes: SoA x|y|z of the scene, padded to whole
// tiles with kTargetPad.  lists: [split][m][query] for d2 and index.
template <int K>
__global__ __launch_bounds__(kBlockThreads) void knn_scan_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                                  const float* __restrict__ pz, uint32_t n, uint32_t /*n_tiles*/,
                                                                  uint32_t n_split, uint32_t split_q, uint32_t split_r,
                                                                  float* __restrict__ list_d, int32_t* __restrict__ list_i,
                                                                  uint32_t q_stride) {
  __shared__ float4 lds[2][3][kTile / 4];
  const uint32_t qb = blockIdx.x / n_split, sp = blockIdx.x - qb * n_split;
  const uint32_t tile0 = sp * split_q + (sp < split_r ? sp : split_r);
  const uint32_t tile1 = tile0 + split_q + (sp < split_r ? 1u : 0u);
  const uint32_t tid = threadIdx.x;
  const uint32_t q = qb * kBlockThreads + tid;
  const uint32_t qc = q < n ? q : 0;  // out-of-range lanes shadow point 0 and store nothing
  const float qx = px[qc], qy = py[qc], qz = pz[qc];
  float d[K];
  int32_t ix[K];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    d[m] = __builtin_inff();
    ix[m] = -1;
  }
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(px);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(py);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(pz);
  {
    const uint32_t g = tile0 * (kTile / 4) + tid;
    lds[0][0][tid] = gx[g];
    lds[0][1][tid] = gy[g];
    lds[0][2][tid] = gz[g];
  }
  __syncthreads();
  uint32_t cur = 0;
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    const bool more = tile + 1 < tile1;
    float4 rx, ry, rz;
    if (more) {
      const uint32_t g = (tile + 1) * (kTile / 4) + tid;
      rx = gx[g];
      ry = gy[g];
      rz = gz[g];
    }
    const float4* __restrict__ LX = lds[cur][0];
    const float4* __restrict__ LY = lds[cur][1];
    const float4* __restrict__ LZ = lds[cur][2];
    const int32_t base = static_cast<int32_t>(tile * kTile);
#pragma unroll 2
    for (uint32_t g = 0; g < kTile / 4; ++g) {
      const float4 X = LX[g], Y = LY[g], Z = LZ[g];
      const float d0 = dist2(qx, qy, qz, X.x, Y.x, Z.x);
      const float d1 = dist2(qx, qy, qz, X.y, Y.y, Z.y);
      const float d2 = dist2(qx, qy, qz, X.z, Y.z, Z.z);
      const float d3 = dist2(qx, qy, qz, X.w, Y.w, Z.w);
      const float m = __builtin_fminf(__builtin_fminf(d0, d1), __builtin_fminf(d2, d3));
      if (m < d[K - 1]) {  // rare once the list has warmed up; the four inserts keep candidates in index order
        const int32_t j = base + static_cast<int32_t>(g * 4);
        list_insert<K>(d, ix, d0, j);
        list_insert<K>(d, ix, d1, j + 1);
        list_insert<K>(d, ix, d2, j + 2);
        list_insert<K>(d, ix, d3, j + 3);
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
  if (q < n) {
#pragma unroll
    for (int m = 0; m < K; ++m) {
      const size_t o = (static_cast<size_t>(sp) * K + m) * q_stride + q;
      list_d[o] = d[m];
      list_i[o] = ix[m];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact K nearest neighbours WITH pruning (scene normals are not under north_star's brute-force mandate; the ICP scan is),
// on the Morton-sorted copy of the scene (xtarget_kernels.hip), which carries a bounding sphere per 64-point run, per
// 1024-point tile and per 16 tiles.  ONE QUERY AT A TIME PER WAVE, its candidates across the 64 lanes.  (Rounds 1-2 kept a
// K-list per lane in registers, like knn_scan_kernel: a wave then takes the ~100-instruction insert branch whenever ANY of
// its 64 queries improves - on the scan of the queries' own neighbourhood that is nearly every group of candidates - and
// 100k scene points are only ~1.5 waves per SIMD, so the launch lasted as long as one wave's 64 lists took to settle:
// 1.3 ms at 20k-100k points, most of a request's scene preparation; 5.1 ms at 1M.  This form: 0.2 / 0.4 / 3.2 ms.)  Here
//   * the list is DISTRIBUTED: lane m holds entry m as one 64-bit key (d2 bits << 32 | original index), so the oracle's
//     (d2, index) order is the unsigned order of the keys and an insert is a lane shift and two compares (~20 instructions);
//   * a 64-point run of the sorted copy is one step: every lane forms the key of its candidate, the wave ballots
//     "key < K-th key" and only the (few) set lanes are inserted; the query's own run - its Morton neighbourhood - is
//     bitonic-sorted instead, which fills the list in one go;
//   * pruning is per query with the same spheres and the same rounding as before (sphere_reachable): super-tiles -> tiles
//     -> runs, four of each level tested side by side, the survivors queued in LDS and their points loaded four runs at
//     a time so that the memory round trips of a query overlap;
//   * a wave takes `qpw` consecutive sorted positions: 1 (4 from 256k points on) - a query is a chain of dependent memory
//     round trips, and what hides them is other waves (measured: 1 < 2 < 4 < ... < 64 at every size below 256k points).
// A pruned sphere holds no point closer than, or tied with, the query's current K-th neighbour, so the lists are those of
// the exhaustive scan, bit for bit; an unreachable +inf distance never enters (as before).  No workgroup barrier anywhere:
// the four waves of a workgroup run independent loops on private LDS queues.
constexpr uint32_t kKnnRunQueue = 192, kKnnTileQueue = 128;
constexpr unsigned long long kKnnEmpty = 0x7F800000FFFFFFFFull;  // (+inf, index -1): an empty list entry

__device__ __forceinline__ unsigned long long wave_read_u64(unsigned long long v, uint32_t lane) {
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), lane);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), lane);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), mask, 64);
  const uint32_t hi = __shfl_xor(static_cast<uint32_t>(v >> 32), mask, 64);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}
// entry of the lane below (lane 0: key 0, which is below every key); lanes >= 32 are never list entries (K <= 32)
__device__ __forceinline__ unsigned long long list_shift_up(unsigned long long v, uint32_t lane) {
  uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v)), 0x111, 0xF, 0xF, true));      // row_shr:1
  uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v >> 32)), 0x111, 0xF, 0xF, true));
  const uint32_t lo15 = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), 15), hi15 = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), 15);
  lo = lane == 16 ? lo15 : lo;  // a row is 16 lanes
  hi = lane == 16 ? hi15 : hi;
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

template <int K>
__global__ __launch_bounds__(kBlockThreads) void knn_wave_kernel(const float* __restrict__ sox, const float* __restrict__ soy,
                                                                  const float* __restrict__ soz, const uint32_t* __restrict__ perm,
                                                                  const float4* __restrict__ table, const float4* __restrict__ sub_table,
                                                                  const float4* __restrict__ super_table, uint32_t n, uint32_t n_tiles,
                                                                  float* __restrict__ list_d, int32_t* __restrict__ list_i,
                                                                  uint32_t q_stride, uint32_t qpw) {
  static_assert(K <= 32, "the list lives in lanes 0..31");
  __shared__ uint32_t run_queue[kBlockThreads / 64][kKnnRunQueue];
  __shared__ uint32_t tile_queue[kBlockThreads / 64][kKnnTileQueue];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* rq = run_queue[wave];
  uint32_t* tq = tile_queue[wave];
  const uint32_t n_runs = (n + 63u) / 64u, n_super = (n_tiles + kSuperTiles - 1) / kSuperTiles;
  const uint32_t first = (blockIdx.x * (kBlockThreads / 64) + wave) * qpw;
  for (uint32_t qi = 0; qi < qpw; ++qi) {
    const uint32_t pos = first + qi;  // wave-uniform
    if (pos >= n) break;
    const float qx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sox[pos])));
    const float qy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(soy[pos])));
    const float qz = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(soz[pos])));
    const uint32_t own_run = pos >> 6, own_tile = pos / kTile;
    unsigned long long list, tau;
    auto key_of = [&](uint32_t j, float x, float y, float z, uint32_t p) -> unsigned long long {
      const float d = dist2(qx, qy, qz, x, y, z);
      const bool valid = j < n && d < __builtin_inff();  // padding slots, overflowed (or NaN) distances: never a neighbour
      return valid ? (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | p : kKnnEmpty;
    };
    // offers the wave's 64 candidate keys: the lanes below the K-th key are inserted one after the other
    auto offer = [&](unsigned long long key) {
      unsigned long long hits = __builtin_amdgcn_ballot_w64(key < tau);
      while (hits) {
        const uint32_t src = static_cast<uint32_t>(__builtin_ctzll(hits));
        hits &= hits - 1;
        const unsigned long long v = wave_read_u64(key, src);
        if (v < tau) {  // an earlier insert of this step may have lowered the bar
          const unsigned long long below = list_shift_up(list, lane);
          list = list < v ? list : (below < v ? v : below);
          tau = wave_read_u64(list, K - 1);
        }
      }
    };
    {  // the query's own run, sorted: entries 0..K-1 are its K nearest there
      const uint32_t j = own_run * 64u + lane;
      unsigned long long key = key_of(j, sox[j], soy[j], soz[j], perm[j]);
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int s = k >> 1; s > 0; s >>= 1) {
          const unsigned long long other = shfl_xor_u64(key, s);
          const bool take_min = ((lane & static_cast<uint32_t>(k)) == 0) == ((lane & static_cast<uint32_t>(s)) == 0);
          const bool other_less = other < key;
          key = (take_min == other_less) ? other : key;
        }
      list = key;
      tau = wave_read_u64(list, K - 1);
    }
    uint32_t n_rq = 0, n_tq = 0;
    // appends `value` of the lanes set in `mask` to a wave-private LDS queue (in lane order)
    auto append = [&](uint32_t* queue, uint32_t& count, unsigned long long mask, bool mine, uint32_t value) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
      if (mine) queue[count + rank] = value;
      count += static_cast<uint32_t>(__builtin_popcountll(mask));
    };
    auto drain_runs = [&]() {
      for (uint32_t i = 0; i < n_rq; i += 4) {
        float X[4], Y[4], Z[4];
        uint32_t P[4], J[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // four runs in flight; a short tail repeats its last run (offered once)
          const uint32_t r = __builtin_amdgcn_readfirstlane(rq[i + k < n_rq ? i + k : n_rq - 1]);
          J[k] = r * 64u + lane;
          X[k] = sox[J[k]];
          Y[k] = soy[J[k]];
          Z[k] = soz[J[k]];
          P[k] = perm[J[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i + k < n_rq) offer(key_of(J[k], X[k], Y[k], Z[k], P[k]));
      }
      n_rq = 0;
    };
    // the runs of (up to) four tiles side by side: lane = (tile slot, run of the tile)
    auto expand_tiles = [&](const uint32_t (&T)[4], uint32_t count) {
      const uint32_t slot = lane >> 4;
      const uint32_t t = slot == 0 ? T[0] : (slot == 1 ? T[1] : (slot == 2 ? T[2] : T[3]));
      const uint32_t run = t * (kTile / 64) + (lane & 15u);
      const float tau_d = __uint_as_float(static_cast<uint32_t>(tau >> 32));
      const bool reach = slot < count && run < n_runs && run != own_run && sphere_reachable(qx, qy, qz, 0.0f, sub_table[run < n_runs ? run : 0], tau_d);
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(reach);
      if (n_rq + 64u > kKnnRunQueue) drain_runs();
      append(rq, n_rq, mask, reach, run);
    };
    auto drain_tiles = [&]() {
      for (uint32_t i = 0; i < n_tq; i += 4) {
        uint32_t T[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) T[k] = __builtin_amdgcn_readfirstlane(tq[i + k < n_tq ? i + k : n_tq - 1]);
        expand_tiles(T, n_tq - i < 4u ? n_tq - i : 4u);
      }
      n_tq = 0;
      drain_runs();
    };
    {  // the rest of the own tile first: it lowers the bar every other tile has to pass
      const uint32_t T[4] = {own_tile, own_tile, own_tile, own_tile};
      expand_tiles(T, 1);
      drain_runs();
    }
    for (uint32_t s0 = 0; s0 < n_super; s0 += 64) {
      const uint32_t sidx = s0 + lane;
      const float tau_d = __uint_as_float(static_cast<uint32_t>(tau >> 32));
      unsigned long long smask =
          __builtin_amdgcn_ballot_w64(sidx < n_super && sphere_reachable(qx, qy, qz, 0.0f, super_table[sidx < n_super ? sidx : 0], tau_d));
      while (smask) {  // the tiles of (up to) four reachable super-tiles side by side
        uint32_t S[4], count = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          S[k] = smask ? s0 + static_cast<uint32_t>(__builtin_ctzll(smask)) : 0u;
          count += smask ? 1u : 0u;
          smask &= smask - (smask != 0);
        }
        const uint32_t slot = lane >> 4;
        const uint32_t sup = slot == 0 ? S[0] : (slot == 1 ? S[1] : (slot == 2 ? S[2] : S[3]));
        const uint32_t t = sup * kSuperTiles + (lane & 15u);
        const float bar = __uint_as_float(static_cast<uint32_t>(tau >> 32));
        const bool reach = slot < count && t < n_tiles && t != own_tile && sphere_reachable(qx, qy, qz, 0.0f, table[t < n_tiles ? t : 0], bar);
        const unsigned long long tmask = __builtin_amdgcn_ballot_w64(reach);
        if (n_tq + 64u > kKnnTileQueue) drain_tiles();
        append(tq, n_tq, tmask, reach, t);
      }
    }
    drain_tiles();
    if (lane < static_cast<uint32_t>(K)) {
      const uint32_t orig = perm[pos];
      list_d[static_cast<size_t>(lane) * q_stride + orig] = __uint_as_float(static_cast<uint32_t>(list >> 32));
      list_i[static_cast<size_t>(lane) * q_stride + orig] = static_cast<int32_t>(static_cast<uint32_t>(list));
    }
  }
}

// one thread per scene point
template <int K>
__global__ __launch_bounds__(kBlockThreads) void normals_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                                 const float* __restrict__ pz, uint32_t n, uint32_t k_eff,
                                                                 uint32_t n_split, const float* __restrict__ list_d,
                                                                 const int32_t* __restrict__ list_i, uint32_t q_stride,
                                                                 int flip, float vx, float vy, float vz,
                                                                 float* __restrict__ out_aos, float* __restrict__ out_planes,
                                                                 size_t plane_stride, int32_t* __restrict__ out_knn) {
  const uint32_t q = blockIdx.x * kBlockThreads + threadIdx.x;
  if (q >= n) return;
  float d[K];
  int32_t ix[K];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    d[m] = list_d[static_cast<size_t>(m) * q_stride + q];
    ix[m] = list_i[static_cast<size_t>(m) * q_stride + q];
  }
  for (uint32_t sp = 1; sp < n_split; ++sp) {
#pragma unroll 1
    for (int m = 0; m < K; ++m) {
      const size_t o = (static_cast<size_t>(sp) * K + m) * q_stride + q;
      const float nd = list_d[o];
      if (!(nd < d[K - 1])) break;  // each split list is sorted: nothing further can enter
      list_insert<K>(d, ix, nd, list_i[o]);
    }
  }
  // covariance as upstream's meanCovLocalPCInd: float32 products accumulated in double
  double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, mean[3] = {0, 0, 0};
#pragma unroll
  for (int m = 0; m < K; ++m) {
    if (static_cast<uint32_t>(m) < k_eff) {
      const int32_t j = ix[m];
      if (out_knn) out_knn[static_cast<size_t>(q) * k_eff + m] = j;
      const float c[3] = {px[j], py[j], pz[j]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) cov[3 * a + b] += static_cast<double>(c[a] * c[b]);
        mean[a] += static_cast<double>(c[a]);
      }
    }
  }
  const double inv = 1.0 / static_cast<double>(k_eff);
#pragma unroll
  for (int a = 0; a < 3; ++a) mean[a] *= inv;
#pragma unroll
  for (int a = 0; a < 9; ++a) cov[a] *= inv;
  double A[3][3], V[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      A[a][b] = cov[3 * a + b] - mean[a] * mean[b];
      V[a][b] = (a == b) ? 1.0 : 0.0;
    }
  // cyclic Jacobi, the oracle's jacobi_eig3 operation for operation
#pragma unroll 1
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = __builtin_fabs(A[0][1]) + __builtin_fabs(A[0][2]) + __builtin_fabs(A[1][2]);
    const double diag = __builtin_fabs(A[0][0]) + __builtin_fabs(A[1][1]) + __builtin_fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = p + 1; r < 3; ++r) {
        if (A[p][r] != 0.0) {
          const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
          const double tt = (theta >= 0 ? 1.0 : -1.0) / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
          const double c = 1.0 / __builtin_sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double akp = A[k][p], akq = A[k][r];
            A[k][p] = c * akp - s * akq;
            A[k][r] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double apk = A[p][k], aqk = A[r][k];
            A[p][k] = c * apk - s * aqk;
            A[r][k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double vkp = V[k][p], vkq = V[k][r];
            V[k][p] = c * vkp - s * vkq;
            V[k][r] = s * vkp + c * vkq;
          }
        }
      }
  }
  double w0 = A[0][0], w1 = A[1][1], w2 = A[2][2];
  double e0 = V[0][0], e1 = V[1][0], e2 = V[2][0], wm = w0;
  if (w1 < wm) { wm = w1; e0 = V[0][1]; e1 = V[1][1]; e2 = V[2][1]; }
  if (w2 < wm) { wm = w2; e0 = V[0][2]; e1 = V[1][2]; e2 = V[2][2]; }
  const double nrm = __builtin_sqrt(e0 * e0 + e1 * e1 + e2 * e2);
  float n0 = static_cast<float>(e0 / nrm), n1 = static_cast<float>(e1 / nrm), n2 = static_cast<float>(e2 / nrm);
  if (flip) {
    const float ax = vx - px[q], ay = vy - py[q], az = vz - pz[q];
    float c = ax * n0;
    const float t = ay * n1, u = az * n2;
    c = c + t;
    c = c + u;
    if (c < 0) { n0 = -n0; n1 = -n1; n2 = -n2; }
  }
  if (out_aos) {
    out_aos[static_cast<size_t>(q) * 3] = n0;
    out_aos[static_cast<size_t>(q) * 3 + 1] = n1;
    out_aos[static_cast<size_t>(q) * 3 + 2] = n2;
  }
  if (out_planes) {
    out_planes[q] = n0;
    out_planes[plane_stride + q] = n1;
    out_planes[2 * plane_stride + q] = n2;
  }
}

void launch_knn_scan(void* stream, int k, DevicePlanes planes, uint32_t n, uint32_t n_tiles, uint32_t n_split,
                     float* list_d, int32_t* list_i, uint32_t q_stride) {
  const uint32_t qblocks = (n + kBlockThreads - 1) / kBlockThreads;
  if (!qblocks) return;
  const uint32_t split_q = n_tiles / n_split, split_r = n_tiles % n_split;
  dim3 grid(qblocks * n_split);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k <= 8)
    hipLaunchKernelGGL(knn_scan_kernel<8>, grid, dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, n_tiles,
                       n_split, split_q, split_r, list_d, list_i, q_stride);
  else if (k <= 20)
    hipLaunchKernelGGL(knn_scan_kernel<20>, grid, dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, n_tiles,
                       n_split, split_q, split_r, list_d, list_i, q_stride);
  else
    hipLaunchKernelGGL(knn_scan_kernel<32>, grid, dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, n_tiles,
                       n_split, split_q, split_r, list_d, list_i, q_stride);
}

// the pruned scan on the sorted copy (sorted original coordinates, sorted position -> original index, per-tile spheres)
void launch_knn_pruned(void* stream, int k, DevicePlanes sorted, const uint32_t* perm, const void* table, const void* sub_table,
                       const void* super_table, uint32_t n, uint32_t n_tiles, float* list_d, int32_t* list_i, uint32_t q_stride) {
  if (!n) return;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* tb = static_cast<const float4*>(table);
  const float4* sb = static_cast<const float4*>(sub_table);
  const float4* pb = static_cast<const float4*>(super_table);
  static const uint32_t forced = std::getenv("MI355_ICP_KNN_QPW") ? static_cast<uint32_t>(std::atoi(std::getenv("MI355_ICP_KNN_QPW"))) : 0u;
  const uint32_t qpw = forced ? forced : n >= (1u << 18) ? 4u : 1u;  // queries per wave (knn_wave_kernel)
  dim3 wgrid((n + 4u * qpw - 1) / (4u * qpw));
  if (k <= 8)
    hipLaunchKernelGGL(knn_wave_kernel<8>, wgrid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride, qpw);
  else if (k <= 20)
    hipLaunchKernelGGL(knn_wave_kernel<20>, wgrid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride, qpw);
  else
    hipLaunchKernelGGL(knn_wave_kernel<32>, wgrid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride, qpw);
}

void launch_normals(void* stream, int k, DevicePlanes planes, uint32_t n, uint32_t k_eff, uint32_t n_split,
                    const float* list_d, const int32_t* list_i, uint32_t q_stride, int flip, const float* viewpoint,
                    float* out_aos, float* out_planes, size_t plane_stride, int32_t* out_knn) {
  const uint32_t blocks = (n + kBlockThreads - 1) / kBlockThreads;
  if (!blocks) return;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k <= 8)
    hipLaunchKernelGGL(normals_kernel<8>, dim3(blocks), dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, k_eff,
                       n_split, list_d, list_i, q_stride, flip, viewpoint[0], viewpoint[1], viewpoint[2], out_aos, out_planes,
                       plane_stride, out_knn);
  else if (k <= 20)
    hipLaunchKernelGGL(normals_kernel<20>, dim3(blocks), dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, k_eff,
                       n_split, list_d, list_i, q_stride, flip, viewpoint[0], viewpoint[1], viewpoint[2], out_aos, out_planes,
                       plane_stride, out_knn);
  else
    hipLaunchKernelGGL(normals_kernel<32>, dim3(blocks), dim3(kBlockThreads), 0, st, planes.x, planes.y, planes.z, n, k_eff,
                       n_split, list_d, list_i, q_stride, flip, viewpoint[0], viewpoint[1], viewpoint[2], out_aos, out_planes,
                       plane_stride, out_knn);
}

}  // namespace mi355

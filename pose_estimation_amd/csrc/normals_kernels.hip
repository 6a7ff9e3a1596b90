// normals_kernels.hip — scene normals on the GPU: the step right before the ICP in the
// reference, cv::ppf_match_3d::computeNormalsPC3d(pc_scene, pc_scene_normals, 20, true, viewpoint)
// (pose_estimation/src/opencv_surface_match.cpp:57-59; third-party, restated in oracle/icp_oracle.c "Scene normals").
//
//   knn_scan_kernel<K>     exact brute-force K nearest neighbours of every scene point within the scene, the NN scan's
//                          machinery (SoA target tiles through LDS, wave-uniform ds_read_b128 broadcasts, unfused float32
//                          d2) with a sorted K-list per lane in registers; the target range is split across workgroups
//   normals_kernel<K>      per point: merge the per-split lists by (d2, index), gather the K neighbours, covariance as
//                          upstream's meanCovLocalPCInd (float32 products summed in double), cyclic Jacobi in double,
//                          smallest-eigenvalue eigenvector -> float32, flipped towards the viewpoint
// Arithmetic contract: neighbour sets identical to the oracle ((d2, index) order, so ties resolve the same way); the
// double-precision covariance / Jacobi follow the oracle's operation order (compile with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cstdint>

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

// Sorted insert by (d, index), for candidates that arrive in ANY index order (the pruned scan below visits tiles of the
// Morton-sorted copy): the oracle's knn_offer order.
template <int K>
__device__ __forceinline__ void list_insert_lex(float (&d)[K], int32_t (&ix)[K], float nd, int32_t ni) {
  if (nd < d[K - 1] || (nd == d[K - 1] && ni < ix[K - 1])) {
    bool lt[K];
#pragma unroll
    for (int m = 0; m < K; ++m) lt[m] = nd < d[m] || (nd == d[m] && ni < ix[m]);
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

// Exact K nearest neighbours WITH pruning (scene normals are not under north_star's brute-force mandate; the ICP scan
// is).  Works on the Morton-sorted copy of the scene (xtarget_kernels.hip), which carries a bounding sphere per 64-point
// run, per 1024-point tile and per 16 tiles.  A workgroup takes 256 CONSECUTIVE sorted positions - spatially coherent
// queries - and
//   1. scans its own tile exhaustively, which fills every lane's K-list with near neighbours;
//   2. lets every lane mark the tiles whose sphere its current K-th distance still reaches (sphere_lower_bound,
//      icp_internal.h: every rounding pushed towards "reachable"; super-tile spheres first);
//   3. stages the union tile by tile through LDS, each wave scanning only the runs one of its own lanes still reaches.
// A pruned tile or run holds no point that is closer than, or tied with, any lane's K-th neighbour, so the lists are exactly
// those of the exhaustive scan: (d2, original index)-sorted, the oracle's order.  Results are stored at the ORIGINAL index.
template <int K>
__global__ __launch_bounds__(kBlockThreads) void knn_pruned_kernel(const float* __restrict__ sox, const float* __restrict__ soy,
                                                                    const float* __restrict__ soz, const uint32_t* __restrict__ perm,
                                                                    const float4* __restrict__ table,
                                                                    const float4* __restrict__ sub_table,
                                                                    const float4* __restrict__ super_table, uint32_t n,
                                                                    uint32_t n_tiles, float* __restrict__ list_d,
                                                                    int32_t* __restrict__ list_i, uint32_t q_stride) {
  __shared__ float4 lds[4][kTile / 4];           // x, y, z (sorted order, uncentred) and original indices of one tile
  __shared__ float4 lds_sphere[kTile / 64 + 1];  // ... the bounding spheres of its 16 runs and its own
  __shared__ uint32_t need[kPrunedMaxTiles / 32];
  const uint32_t tid = threadIdx.x;
  const uint32_t pos = blockIdx.x * kBlockThreads + tid;
  const uint32_t pc = pos < n ? pos : n - 1;  // out-of-range lanes shadow the last point and store nothing
  const float qx = sox[pc], qy = soy[pc], qz = soz[pc];
  const uint32_t own = blockIdx.x * kBlockThreads / kTile;
  float d[K];
  int32_t ix[K];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    d[m] = __builtin_inff();
    ix[m] = -1;
  }
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(sox);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(soy);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(soz);
  const float4* __restrict__ gp = reinterpret_cast<const float4*>(perm);
  // stage one tile through LDS (its spheres travel with it); every wave scans the 64-point runs one of its lanes can
  // still reach with its current K-th distance
  auto scan_tile = [&](uint32_t tile) {
    const uint32_t g0 = tile * (kTile / 4) + tid;
    lds[0][tid] = gx[g0];
    lds[1][tid] = gy[g0];
    lds[2][tid] = gz[g0];
    lds[3][tid] = gp[g0];
    if (tid < kTile / 64) lds_sphere[tid] = sub_table[tile * (kTile / 64) + tid];
    if (tid == kTile / 64) lds_sphere[tid] = table[tile];
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(sphere_reachable(qx, qy, qz, 0.0f, lds_sphere[kTile / 64], d[K - 1])) != 0) {
      for (uint32_t sub = 0; sub < kTile / 64; ++sub) {
        if (__builtin_amdgcn_ballot_w64(sphere_reachable(qx, qy, qz, 0.0f, lds_sphere[sub], d[K - 1])) == 0) continue;
#pragma unroll 2
        for (uint32_t g = sub * 16; g < sub * 16 + 16; ++g) {
          const float4 X = lds[0][g], Y = lds[1][g], Z = lds[2][g], P = lds[3][g];
          const float d0 = dist2(qx, qy, qz, X.x, Y.x, Z.x);
          const float d1 = dist2(qx, qy, qz, X.y, Y.y, Z.y);
          const float d2 = dist2(qx, qy, qz, X.z, Y.z, Z.z);
          const float d3 = dist2(qx, qy, qz, X.w, Y.w, Z.w);
          const float m = __builtin_fminf(__builtin_fminf(d0, d1), __builtin_fminf(d2, d3));
          if (m <= d[K - 1]) {  // rare once the list has warmed up (<=: an equal distance with a lower index still enters)
            list_insert_lex<K>(d, ix, d0, static_cast<int32_t>(__float_as_uint(P.x)));
            list_insert_lex<K>(d, ix, d1, static_cast<int32_t>(__float_as_uint(P.y)));
            list_insert_lex<K>(d, ix, d2, static_cast<int32_t>(__float_as_uint(P.z)));
            list_insert_lex<K>(d, ix, d3, static_cast<int32_t>(__float_as_uint(P.w)));
          }
        }
      }
    }
    __syncthreads();
  };
  for (uint32_t w = tid; w < kPrunedMaxTiles / 32; w += kBlockThreads) need[w] = 0u;
  scan_tile(own);  // fills every lane's K-list with near neighbours: d[K - 1] bounds what any other point has to beat
  // every other tile some lane can still reach with its own K-th distance: super-tile spheres, then tile spheres.  No
  // bound is taken over a group of queries, so a Morton jump inside the run costs twice the tiles, not all of them.
  mark_reachable_tiles(qx, qy, qz, d[K - 1], super_table, table, 0, n_tiles, need);
  __syncthreads();
  for (uint32_t w = 0; w < (n_tiles + 31) / 32; ++w) {
    uint32_t word = __builtin_amdgcn_readfirstlane(need[w]);
    while (word) {
      const uint32_t t = w * 32 + static_cast<uint32_t>(__builtin_ctz(word));
      word &= word - 1;
      if (t != own) scan_tile(t);
    }
  }
  if (pos < n) {
    const uint32_t orig = perm[pos];
#pragma unroll
    for (int m = 0; m < K; ++m) {
      list_d[static_cast<size_t>(m) * q_stride + orig] = d[m];
      list_i[static_cast<size_t>(m) * q_stride + orig] = ix[m];
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
  const uint32_t qblocks = (n + kBlockThreads - 1) / kBlockThreads;
  if (!qblocks) return;
  dim3 grid(qblocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* tb = static_cast<const float4*>(table);
  const float4* sb = static_cast<const float4*>(sub_table);
  const float4* pb = static_cast<const float4*>(super_table);
  if (k <= 8)
    hipLaunchKernelGGL(knn_pruned_kernel<8>, grid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride);
  else if (k <= 20)
    hipLaunchKernelGGL(knn_pruned_kernel<20>, grid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride);
  else
    hipLaunchKernelGGL(knn_pruned_kernel<32>, grid, dim3(kBlockThreads), 0, st, sorted.x, sorted.y, sorted.z, perm, tb, sb, pb, n, n_tiles, list_d,
                       list_i, q_stride);
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

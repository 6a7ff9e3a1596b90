// icp_fused_kernels.hip — the fast NN scan for large clouds: a FUSED-arithmetic filter in front of the exact test.
//
// The exact contract (oracle/icp_oracle.c) forbids FMA in d2, which costs 9 VALU issue slots per pair.  This kernel
// evaluates every pair with the fused surrogate
//        s = fma(dz, dz, fma(dy, dy, dx * dx))          (same float32 differences dx, dy, dz as the exact formula)
// in 7 slots (3 v_sub, 1 v_mul, 2 v_fma, 1 v_min), and does NOT decide the nearest neighbour.  It only keeps, per
// source point, the three smallest per-chunk minima of s (chunk = 64 consecutive targets).  Because s and the exact
// d2 are both within 4 roundings of the real dx^2 + dy^2 + dz^2, |s - d2| <= 2^-21 * d2 (+ 2^-147 near underflow), so
// the exact nearest neighbour lies in a chunk whose surrogate minimum is <= v1 * (1 + 2^-19) + 2^-140, v1 being the
// smallest surrogate chunk minimum.  resolve_fused_kernel (icp_kernels.hip, exact arithmetic, checked for fused ops)
// re-scans those one or two candidate chunks with the exact formula and picks the (d2, index)-lexicographic minimum;
// if three or more chunks fall inside the band (exact ties spread over the cloud, e.g. lattices) the point is handed to
// the exact scan (nn_scan_kernel) through the engine's overflow list.  Results are bit-identical to the exact path by
// construction; tests run the whole parity suite in both modes.
//
// This file is the ONLY place where a fused float32 multiply-add may appear in the correspondence path, and it is
// therefore excluded from `make check-isa`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_internal.h"

namespace mi355 {

namespace {

__device__ __forceinline__ void transform_point(const float* __restrict__ m, float x, float y, float z, float& ox,
                                                float& oy, float& oz) {
  // the source transform stays exact (unfused): it defines the query point for both the filter and the exact test
  float a, b;
  a = __fmul_rn(m[0], x); b = __fmul_rn(m[1], y); a = __fadd_rn(a, b); b = __fmul_rn(m[2], z); a = __fadd_rn(a, b); ox = __fadd_rn(a, m[3]);
  a = __fmul_rn(m[4], x); b = __fmul_rn(m[5], y); a = __fadd_rn(a, b); b = __fmul_rn(m[6], z); a = __fadd_rn(a, b); oy = __fadd_rn(a, m[7]);
  a = __fmul_rn(m[8], x); b = __fmul_rn(m[9], y); a = __fadd_rn(a, b); b = __fmul_rn(m[10], z); a = __fadd_rn(a, b); oz = __fadd_rn(a, m[11]);
}

__device__ __forceinline__ float surrogate(float qx, float qy, float qz, float tx, float ty, float tz) {
  const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

}  // namespace

// grid / block / workgroup-to-work mapping exactly as nn_scan_kernel.  keys3: [3][n_split][key_stride] u64,
// (surrogate bits << 32 | fused-chunk id), rank 0 = smallest.
template <int kPointsPerLane>
__global__ __launch_bounds__(kBlockThreads) void nn_scan_fused_kernel(const JobDesc* __restrict__ jobs,
                                                                       const float* __restrict__ sx,
                                                                       const float* __restrict__ sy,
                                                                       const float* __restrict__ sz,
                                                                       const float* __restrict__ tx,
                                                                       const float* __restrict__ ty,
                                                                       const float* __restrict__ tz,
                                                                       unsigned long long* __restrict__ keys3,
                                                                       uint32_t key_stride, uint32_t rank_stride) {
  __shared__ float4 lds[2][3][kTile / 4];
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
  float m1[kPointsPerLane], m2[kPointsPerLane], m3[kPointsPerLane];
  uint32_t c1[kPointsPerLane], c2[kPointsPerLane], c3[kPointsPerLane];
  const uint32_t src_base = J.src_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const uint32_t i = src_base + p * kBlockThreads;
    transform_point(pose, sx[i], sy[i], sz[i], qx[p], qy[p], qz[p]);
    m1[p] = m2[p] = m3[p] = __builtin_inff();
    c1[p] = c2[p] = c3[p] = tile0 * kFusedChunksPerTile;
  }
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(tx + J.tgt_off);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(ty + J.tgt_off);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(tz + J.tgt_off);
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
    const uint32_t chunk_base = tile * kFusedChunksPerTile;
    for (uint32_t c = 0; c < kFusedChunksPerTile; ++c) {
      float cmin[kPointsPerLane];
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) cmin[p] = __builtin_inff();
#pragma unroll
      for (uint32_t g = 0; g < kFusedChunk / 4; ++g) {
        const float4 X = LX[c * (kFusedChunk / 4) + g];
        const float4 Y = LY[c * (kFusedChunk / 4) + g];
        const float4 Z = LZ[c * (kFusedChunk / 4) + g];
#pragma unroll
        for (int p = 0; p < kPointsPerLane; ++p) {
          const float s0 = surrogate(qx[p], qy[p], qz[p], X.x, Y.x, Z.x);
          const float s1 = surrogate(qx[p], qy[p], qz[p], X.y, Y.y, Z.y);
          const float s2 = surrogate(qx[p], qy[p], qz[p], X.z, Y.z, Z.z);
          const float s3 = surrogate(qx[p], qy[p], qz[p], X.w, Y.w, Z.w);
          cmin[p] = __builtin_fminf(__builtin_fminf(cmin[p], s0), s1);
          cmin[p] = __builtin_fminf(__builtin_fminf(cmin[p], s2), s3);
        }
      }
      const uint32_t chunk = chunk_base + c;
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) {  // sorted insert of (cmin, chunk) into the three smallest; strict <: earlier chunk first on ties
        // everything through SSA scalars: selects between array ELEMENTS make hipcc select between their stack
        // addresses instead (scratch traffic + branches)
        const float cm = cmin[p], o1 = m1[p], o2 = m2[p], o3 = m3[p];
        const uint32_t i1 = c1[p], i2 = c2[p], i3 = c3[p];
        const bool l1 = cm < o1, l2 = cm < o2, l3 = cm < o3;
        const float keep3 = l3 ? cm : o3, keep2 = l2 ? cm : o2;
        const uint32_t keepi3 = l3 ? chunk : i3, keepi2 = l2 ? chunk : i2;
        m3[p] = l2 ? o2 : keep3;
        c3[p] = l2 ? i2 : keepi3;
        m2[p] = l1 ? o1 : keep2;
        c2[p] = l1 ? i1 : keepi2;
        m1[p] = l1 ? cm : o1;
        c1[p] = l1 ? chunk : i1;
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
  const size_t out = static_cast<size_t>(sp) * key_stride + J.out_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const size_t o = out + p * kBlockThreads;
    keys3[o] = (static_cast<unsigned long long>(__float_as_uint(m1[p])) << 32) | c1[p];
    keys3[rank_stride + o] = (static_cast<unsigned long long>(__float_as_uint(m2[p])) << 32) | c2[p];
    keys3[2 * static_cast<size_t>(rank_stride) + o] = (static_cast<unsigned long long>(__float_as_uint(m3[p])) << 32) | c3[p];
  }
}

void launch_nn_scan_fused(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                          DevicePlanes src, DevicePlanes tgt, unsigned long long* keys3, uint32_t key_stride,
                          uint32_t rank_stride) {
  dim3 grid(max_blocks_x, static_cast<uint32_t>(n_jobs));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (points_per_lane) {
    case 2:
      hipLaunchKernelGGL(nn_scan_fused_kernel<2>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys3, key_stride, rank_stride);
      break;
    case 4:
      hipLaunchKernelGGL(nn_scan_fused_kernel<4>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys3, key_stride, rank_stride);
      break;
    default:
      hipLaunchKernelGGL(nn_scan_fused_kernel<8>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tgt.x,
                         tgt.y, tgt.z, keys3, key_stride, rank_stride);
      break;
  }
}

// =====================================================================================================================
// Second filter, for the largest launches: the EXPANSION surrogate on spatially sorted, tile-centred targets.
//
//   s(q, t) = |qc|^2 + ( tt + q'.tc ),   qc = fl(q - c_tile), q' = -2 qc, tc = fl(t - c_tile), tt = fl(|tc|^2)
//
// costs 3 v_fma + 1 v_min per pair (4 issue slots instead of 7): tt and tc are precomputed per target at upload
// (XTarget, icp_engine.hip: targets Morton-sorted so a 1024-target tile is spatially compact, then centred on the tile),
// q' and |qc|^2 once per (source point, tile).  The price is cancellation: s is only good to an ABSOLUTE error
//        |s - d2_exact| <= 30.4 u (|qc|^2 + r_tile^2),  u = 2^-24          (derivation: DESIGN.md 3.1)
// which is why the tiles must be compact.  The kernel uses E = 2^-18 (|qc|^2 + r_tile^2) + 2^-140 (2x margin), and per
// 64-target chunk forms a lower bound L = max(s_min - E, 0) and an upper bound U = s_min + E of the exact minimum in that
// chunk.  It keeps the three smallest L (with chunk ids) and the smallest U per source point.  The exact nearest
// neighbour (and every exact tie) lies in a chunk with L <= U_min; resolve_xfilter_kernel (exact arithmetic,
// icp_kernels.hip) re-scans those one or two chunks with the contract formula on the ORIGINAL coordinates and picks the
// (d2, original index)-lexicographic minimum; three or more candidate chunks, or a point whose bound is not finite, go
// to the all-exact scan through the overflow list.  Bit-identical results by construction.
template <int kPointsPerLane>
__global__ __launch_bounds__(kBlockThreads) void nn_scan_xfilter_kernel(const JobDesc* __restrict__ jobs,
                                                                         const float* __restrict__ sx,
                                                                         const float* __restrict__ sy,
                                                                         const float* __restrict__ sz,
                                                                         const float* __restrict__ tcx,
                                                                         const float* __restrict__ tcy,
                                                                         const float* __restrict__ tcz,
                                                                         const float* __restrict__ ttp,
                                                                         const float4* __restrict__ tile_table,
                                                                         unsigned long long* __restrict__ keys4,
                                                                         uint32_t key_stride, uint32_t rank_stride) {
  __shared__ float4 lds[2][4][kTile / 4];  // x, y, z (centred) and tt: 2 x 16 KiB
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
  float l1[kPointsPerLane], l2[kPointsPerLane], l3[kPointsPerLane], umin[kPointsPerLane];
  uint32_t c1[kPointsPerLane], c2[kPointsPerLane], c3[kPointsPerLane];
  bool unsafe = false;
  const uint32_t src_base = J.src_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const uint32_t i = src_base + p * kBlockThreads;
    transform_point(pose, sx[i], sy[i], sz[i], qx[p], qy[p], qz[p]);
    l1[p] = l2[p] = l3[p] = umin[p] = __builtin_inff();
    c1[p] = c2[p] = c3[p] = tile0 * kFusedChunksPerTile;
  }
  const uint32_t tile_base = J.tgt_off / kTile;  // tgt_off is a multiple of kTile
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(tcx + J.tgt_off);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(tcy + J.tgt_off);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(tcz + J.tgt_off);
  const float4* __restrict__ gw = reinterpret_cast<const float4*>(ttp + J.tgt_off);
  {
    const uint32_t g = tile0 * (kTile / 4) + tid;
    lds[0][0][tid] = gx[g];
    lds[0][1][tid] = gy[g];
    lds[0][2][tid] = gz[g];
    lds[0][3][tid] = gw[g];
  }
  __syncthreads();
  uint32_t cur = 0;
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    const bool more = tile + 1 < tile1;
    float4 rx, ry, rz, rw;
    if (more) {
      const uint32_t g = (tile + 1) * (kTile / 4) + tid;
      rx = gx[g];
      ry = gy[g];
      rz = gz[g];
      rw = gw[g];
    }
    const float4 tc = tile_table[tile_base + tile];  // centre xyz, r^2 (wave-uniform)
    float ax[kPointsPerLane], ay[kPointsPerLane], az[kPointsPerLane], qq[kPointsPerLane], ee[kPointsPerLane];
#pragma unroll
    for (int p = 0; p < kPointsPerLane; ++p) {
      const float cx = qx[p] - tc.x, cy = qy[p] - tc.y, cz = qz[p] - tc.z;
      ax[p] = -2.0f * cx;
      ay[p] = -2.0f * cy;
      az[p] = -2.0f * cz;
      qq[p] = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx));
      ee[p] = __builtin_fmaf(3.814697265625e-06f, qq[p] + tc.w, 7.1746481e-43f);  // 2^-18 (|qc|^2 + r^2) + 2^-140
      unsafe = unsafe || !(ee[p] < 1.0e37f);                                           // inf / NaN bound: exact scan instead
    }
    const float4* __restrict__ LX = lds[cur][0];
    const float4* __restrict__ LY = lds[cur][1];
    const float4* __restrict__ LZ = lds[cur][2];
    const float4* __restrict__ LW = lds[cur][3];
    const uint32_t chunk_base = tile * kFusedChunksPerTile;
    for (uint32_t c = 0; c < kFusedChunksPerTile; ++c) {
      float cmin[kPointsPerLane];
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) cmin[p] = __builtin_inff();
#pragma unroll
      for (uint32_t g = 0; g < kFusedChunk / 4; ++g) {
        const float4 X = LX[c * (kFusedChunk / 4) + g];
        const float4 Y = LY[c * (kFusedChunk / 4) + g];
        const float4 Z = LZ[c * (kFusedChunk / 4) + g];
        const float4 W = LW[c * (kFusedChunk / 4) + g];
#pragma unroll
        for (int p = 0; p < kPointsPerLane; ++p) {
          const float m0 = __builtin_fmaf(az[p], Z.x, __builtin_fmaf(ay[p], Y.x, __builtin_fmaf(ax[p], X.x, W.x)));
          const float m1 = __builtin_fmaf(az[p], Z.y, __builtin_fmaf(ay[p], Y.y, __builtin_fmaf(ax[p], X.y, W.y)));
          const float m2 = __builtin_fmaf(az[p], Z.z, __builtin_fmaf(ay[p], Y.z, __builtin_fmaf(ax[p], X.z, W.z)));
          const float m3 = __builtin_fmaf(az[p], Z.w, __builtin_fmaf(ay[p], Y.w, __builtin_fmaf(ax[p], X.w, W.w)));
          cmin[p] = __builtin_fminf(__builtin_fminf(cmin[p], m0), m1);
          cmin[p] = __builtin_fminf(__builtin_fminf(cmin[p], m2), m3);
        }
      }
      const uint32_t chunk = chunk_base + c;
#pragma unroll
      for (int p = 0; p < kPointsPerLane; ++p) {
        const float s = cmin[p] + qq[p];
        const float lo = __builtin_fmaxf(s - ee[p], 0.0f), hi = s + ee[p];
        umin[p] = __builtin_fminf(umin[p], hi);
        const float o1 = l1[p], o2 = l2[p], o3 = l3[p];
        const uint32_t i1 = c1[p], i2 = c2[p], i3 = c3[p];
        const bool b1 = lo < o1, b2 = lo < o2, b3 = lo < o3;
        const float keep3 = b3 ? lo : o3, keep2 = b2 ? lo : o2;
        const uint32_t keepi3 = b3 ? chunk : i3, keepi2 = b2 ? chunk : i2;
        l3[p] = b2 ? o2 : keep3;
        c3[p] = b2 ? i2 : keepi3;
        l2[p] = b1 ? o1 : keep2;
        c2[p] = b1 ? i1 : keepi2;
        l1[p] = b1 ? lo : o1;
        c1[p] = b1 ? chunk : i1;
      }
    }
    if (more) {
      lds[cur ^ 1][0][tid] = rx;
      lds[cur ^ 1][1][tid] = ry;
      lds[cur ^ 1][2][tid] = rz;
      lds[cur ^ 1][3][tid] = rw;
    }
    __syncthreads();
    cur ^= 1;
  }
  const size_t out = static_cast<size_t>(sp) * key_stride + J.out_off + sb * kScanSrcPerBlock + tid;
#pragma unroll
  for (int p = 0; p < kPointsPerLane; ++p) {
    const size_t o = out + p * kBlockThreads;
    // an unsafe lane publishes "everything is a candidate": L = 0 three times, U = +inf -> overflow -> exact scan
    const float w1 = unsafe ? 0.0f : l1[p], w2 = unsafe ? 0.0f : l2[p], w3 = unsafe ? 0.0f : l3[p];
    const float wu = unsafe ? __builtin_inff() : umin[p];
    keys4[o] = (static_cast<unsigned long long>(__float_as_uint(w1)) << 32) | c1[p];
    keys4[rank_stride + o] = (static_cast<unsigned long long>(__float_as_uint(w2)) << 32) | c2[p];
    keys4[2 * static_cast<size_t>(rank_stride) + o] = (static_cast<unsigned long long>(__float_as_uint(w3)) << 32) | c3[p];
    keys4[3 * static_cast<size_t>(rank_stride) + o] = static_cast<unsigned long long>(__float_as_uint(wu)) << 32;
  }
}

// =====================================================================================================================
// The expansion filter on the MATRIX cores.  tt + q'.tc is a K = 4 contraction, [tcx tcy tcz tt] . [q'x q'y q'z 1]^T, so
// a pair of v_mfma_f32_32x32x2_f32 produces the surrogate of 32 targets x 32 sources (exact float32 fma chain, one
// rounding per product: cdna_hip_programming.md "FP32-input MFMA") and the VALU is left with the minimum over the 16
// accumulator registers - about one issue slot per pair instead of four.  f32 MFMA runs at the VALU rate, so this
// kernel alone is no faster than nn_scan_xfilter_kernel; the point is that the matrix pipe and the vector ALU work
// concurrently, so a launch that gives part of the source points to these workgroups and the rest to the VALU kernel's
// finishes sooner than either alone (icp_engine.hip, run_fused_index_phase).
//
// Layout.  B operand (columns = sources): lane l holds, for source (l & 31), component k = l >> 5 of the K-pair:
// B0 = (q'x | q'y), B1 = (q'z | 1).  A operand (rows = targets): lane l holds target (l & 31) of the 32-target block,
// A0 = (tcx | tcy), A1 = (tcz | tt), read from the LDS tile with one ds_read_b32 each (lanes 0-31 walk one plane, lanes
// 32-63 the next: conflict-free).  D: lane l holds column (l & 31) and the 16 rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5),
// so the per-source minimum over a block is an in-lane minimum over the 16 registers plus one exchange between lanes l
// and l + 32 per 64-target chunk.  Same bounds, same keys, same resolve as the VALU kernel.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// __launch_bounds__(256, 2): a register budget of <= 256 makes hipcc pick the VGPR-destination form of the MFMA; with
// the default (512 registers allowed) it accumulates in AGPRs and spends one v_accvgpr_read per value, doubling the
// VALU work this kernel exists to avoid.
template <int kSets>  // 32-source column sets per wave: a workgroup covers 128 * kSets source points
__global__ __launch_bounds__(kBlockThreads, 2) void nn_scan_xmfma_kernel(const JobDesc* __restrict__ jobs,
                                                                       const float* __restrict__ sx,
                                                                       const float* __restrict__ sy,
                                                                       const float* __restrict__ sz,
                                                                       const float* __restrict__ tcx,
                                                                       const float* __restrict__ tcy,
                                                                       const float* __restrict__ tcz,
                                                                       const float* __restrict__ ttp,
                                                                       const float4* __restrict__ tile_table,
                                                                       unsigned long long* __restrict__ keys4,
                                                                       uint32_t key_stride, uint32_t rank_stride,
                                                                       uint32_t first_block) {
  __shared__ float lds[2][4][kTile];  // x, y, z (centred) and tt: 2 x 16 KiB
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t n_split = J.n_split;
  const uint32_t bxi = blockIdx.x + first_block * n_split;  // this launch covers scan blocks [first_block, ...)
  const uint32_t sb = bxi / n_split;
  const uint32_t sp = bxi - sb * n_split;
  if (sb >= J.scan_blocks) return;
  constexpr uint32_t kScanSrcPerBlock = 128 * kSets;
  const uint32_t tile0 = sp * J.split_q + (sp < J.split_r ? sp : J.split_r);
  const uint32_t tile1 = tile0 + J.split_q + (sp < J.split_r ? 1u : 0u);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool hi = lane >= 32;
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  float qx[kSets], qy[kSets], qz[kSets];
  float l1[kSets], l2[kSets], l3[kSets], umin[kSets];
  uint32_t c1[kSets], c2[kSets], c3[kSets];
  bool unsafe = false;
  const uint32_t src_local = (wave * kSets) * 32 + (lane & 31u);
#pragma unroll
  for (int m = 0; m < kSets; ++m) {
    const uint32_t i = J.src_off + sb * kScanSrcPerBlock + src_local + m * 32;  // padded planes: always in bounds
    transform_point(pose, sx[i], sy[i], sz[i], qx[m], qy[m], qz[m]);
    l1[m] = l2[m] = l3[m] = umin[m] = __builtin_inff();
    c1[m] = c2[m] = c3[m] = tile0 * kFusedChunksPerTile;
  }
  const uint32_t tile_base = J.tgt_off / kTile;
  const float4* __restrict__ gx = reinterpret_cast<const float4*>(tcx + J.tgt_off);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(tcy + J.tgt_off);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(tcz + J.tgt_off);
  const float4* __restrict__ gw = reinterpret_cast<const float4*>(ttp + J.tgt_off);
  {
    const uint32_t g = tile0 * (kTile / 4) + tid;
    reinterpret_cast<float4*>(lds[0][0])[tid] = gx[g];
    reinterpret_cast<float4*>(lds[0][1])[tid] = gy[g];
    reinterpret_cast<float4*>(lds[0][2])[tid] = gz[g];
    reinterpret_cast<float4*>(lds[0][3])[tid] = gw[g];
  }
  __syncthreads();
  uint32_t cur = 0;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    const bool more = tile + 1 < tile1;
    float4 rx, ry, rz, rw;
    if (more) {
      const uint32_t g = (tile + 1) * (kTile / 4) + tid;
      rx = gx[g];
      ry = gy[g];
      rz = gz[g];
      rw = gw[g];
    }
    const float4 tc = tile_table[tile_base + tile];
    float b0[kSets], b1[kSets], qq[kSets], ee[kSets];
#pragma unroll
    for (int m = 0; m < kSets; ++m) {
      const float cx = qx[m] - tc.x, cy = qy[m] - tc.y, cz = qz[m] - tc.z;
      b0[m] = hi ? -2.0f * cy : -2.0f * cx;
      b1[m] = hi ? 1.0f : -2.0f * cz;
      qq[m] = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx));
      ee[m] = __builtin_fmaf(3.814697265625e-06f, qq[m] + tc.w, 7.1746481e-43f);
      unsafe = unsafe || !(ee[m] < 1.0e37f);
    }
    const float* __restrict__ A0p = hi ? lds[cur][1] : lds[cur][0];  // (tcx | tcy)
    const float* __restrict__ A1p = hi ? lds[cur][3] : lds[cur][2];  // (tcz | tt)
    const uint32_t chunk_base = tile * kFusedChunksPerTile;
    for (uint32_t c = 0; c < kFusedChunksPerTile; ++c) {
      float cmin[kSets];
#pragma unroll
      for (int m = 0; m < kSets; ++m) cmin[m] = __builtin_inff();
#pragma unroll
      for (uint32_t blk = 0; blk < kFusedChunk / 32; ++blk) {
        const uint32_t t = c * kFusedChunk + blk * 32 + (lane & 31u);
        const float a0 = A0p[t], a1 = A1p[t];
#pragma unroll
        for (int m = 0; m < kSets; ++m) {
          f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[m], zero, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[m], acc, 0, 0, 0);
          float v = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
          v = __builtin_fminf(__builtin_fminf(v, acc[3]), acc[4]);
          v = __builtin_fminf(__builtin_fminf(v, acc[5]), acc[6]);
          v = __builtin_fminf(__builtin_fminf(v, acc[7]), acc[8]);
          v = __builtin_fminf(__builtin_fminf(v, acc[9]), acc[10]);
          v = __builtin_fminf(__builtin_fminf(v, acc[11]), acc[12]);
          v = __builtin_fminf(__builtin_fminf(v, acc[13]), acc[14]);
          cmin[m] = __builtin_fminf(__builtin_fminf(cmin[m], v), acc[15]);
        }
      }
      const uint32_t chunk = chunk_base + c;
#pragma unroll
      for (int m = 0; m < kSets; ++m) {
        const float both = __builtin_fminf(cmin[m], __shfl_xor(cmin[m], 32, 64));  // rows of the other lane half
        const float s = both + qq[m];
        const float lo = __builtin_fmaxf(s - ee[m], 0.0f), up = s + ee[m];
        umin[m] = __builtin_fminf(umin[m], up);
        const float o1 = l1[m], o2 = l2[m], o3 = l3[m];
        const uint32_t i1 = c1[m], i2 = c2[m], i3 = c3[m];
        const bool d1 = lo < o1, d2 = lo < o2, d3 = lo < o3;
        const float keep3 = d3 ? lo : o3, keep2 = d2 ? lo : o2;
        const uint32_t keepi3 = d3 ? chunk : i3, keepi2 = d2 ? chunk : i2;
        l3[m] = d2 ? o2 : keep3;
        c3[m] = d2 ? i2 : keepi3;
        l2[m] = d1 ? o1 : keep2;
        c2[m] = d1 ? i1 : keepi2;
        l1[m] = d1 ? lo : o1;
        c1[m] = d1 ? chunk : i1;
      }
    }
    if (more) {
      reinterpret_cast<float4*>(lds[cur ^ 1][0])[tid] = rx;
      reinterpret_cast<float4*>(lds[cur ^ 1][1])[tid] = ry;
      reinterpret_cast<float4*>(lds[cur ^ 1][2])[tid] = rz;
      reinterpret_cast<float4*>(lds[cur ^ 1][3])[tid] = rw;
    }
    __syncthreads();
    cur ^= 1;
  }
  unsafe = unsafe || __shfl_xor(static_cast<int>(unsafe), 32, 64) != 0;
  if (!hi) {
    const size_t out = static_cast<size_t>(sp) * key_stride + J.out_off + sb * kScanSrcPerBlock + src_local;
#pragma unroll
    for (int m = 0; m < kSets; ++m) {
      const size_t o = out + m * 32;
      const float w1 = unsafe ? 0.0f : l1[m], w2 = unsafe ? 0.0f : l2[m], w3 = unsafe ? 0.0f : l3[m];
      const float wu = unsafe ? __builtin_inff() : umin[m];
      keys4[o] = (static_cast<unsigned long long>(__float_as_uint(w1)) << 32) | c1[m];
      keys4[rank_stride + o] = (static_cast<unsigned long long>(__float_as_uint(w2)) << 32) | c2[m];
      keys4[2 * static_cast<size_t>(rank_stride) + o] = (static_cast<unsigned long long>(__float_as_uint(w3)) << 32) | c3[m];
      keys4[3 * static_cast<size_t>(rank_stride) + o] = static_cast<unsigned long long>(__float_as_uint(wu)) << 32;
    }
  }
}

// =====================================================================================================================
// The expansion filter on the matrix cores at the bf16 rate: SPLIT-bf16 operands.
//
// f32-input MFMA runs on the vector ALUs (measured: no overlap with VALU work, tools/ubench_mfma_valu.hip), so the
// kernel above buys nothing.  v_mfma_f32_32x32x16_bf16 is a different unit: 1024 products of a K = 16 contraction in 32
// cycles.  Sixteen K slots are enough for the expansion surrogate of points QUANTISED to two bfloat16 pieces per
// coordinate (16 significand bits).  Notation: u = 2^-24; q the transformed source point, t a target, c the tile centre
// (all float32) - since round 2 the centre of a GROUP of kXbCentreGroup consecutive tiles, so that the split of q below
// is redone once per group, not once per tile; qc = fl(q - c), tc = fl(t - c); r^2 >= |tc|^2 and >= |tc'|^2 for every
// target of the group (rounded up at upload); N := |qc|^2 + r^2.
//
//   q'  = -2 qc (exact), split into bf16 pieces q'h + q'l, |q' - (q'h + q'l)| <= 2^-16 |q'| per coordinate
//         (two round-to-nearest bf16 roundings: 2^-8 of 2^-8); qc' := -(q'h + q'l) / 2
//   tc' = th + tl likewise (xtarget_kernels.hip), tt' = fl(|tc'|^2) from double, in three bf16 pieces t1 + t2 + t3
//   slots 0-11: (th, th, tl, 0) x (q'h, q'l, q'h, 0) per coordinate      = q'.tc' - sum tl q'l (every product is exact in float32)
//   slots 12-14: (t1, t2, t3) x (1, 1, 1)                                = tt'     slot 15: 1 x b, a bias (see min16)
//   s = fl(MFMA + fl(qq - b)),  qq = fl(|qc'|^2)      ~   S := |qc' - tc'|^2, the squared distance of the QUANTISED points,
//   short of the three LOW x LOW products.  Since round 4 their slots hold zero on both sides: this kernel runs at the clock the
//   chip can hold under its load (1.95-2.05 GHz of 2.4), and what raises that clock is less energy per MFMA - zero operands in 3
//   of the 16 slots bought 3.5-3.9 % of the launch (profiles/r04u_zero_slots_ab.txt; cdna_hip_programming.md rule 28), against
//   term (h) below.  Zeroing more (the third piece of tt', every low piece of the source) gained nothing further.
//
// What the code needs is ONE inequality per granule:  exact contract minimum d^2 >= L := max((s_min - E) * 0.99902, 0),
// which follows from  s <= (1 + 2^-10) d^2 + E  for every pair  (0.99902 < 1 / (1 + 2^-10)).  The terms of that
// inequality, each as a multiple of u N:
//   (a) quantisation.  qc' - tc' differs from qc - tc by at most D = 2^-16 (|qc| + r), so with e = 2^-10:
//       S <= (1 + e) |qc - tc|^2 + (1 + 1/e) D^2,  and (1 + 2^10) 2^-32 (|qc| + r)^2 <= 2050 * 2^-32 * N        =   8.1 u N
//   (b) centring.  qc - tc differs from q - t by at most u (|qc| + r) (two float32 subtractions), so
//       |qc - tc|^2 <= |q - t|^2 + 2 |q - t| u (|qc| + r) + ... <= |q - t|^2 + 4.2 u N                           =   4.2 u N
//   (c) the contract value itself: d^2 is within 5 u |q - t|^2 <= 10 u N of |q - t|^2 (3 sub, 3 mul, 2 add)     =  10   u N
//   (d) qq: three roundings of partial sums <= |q'|^2, times 1/4                                                  =   3   u N
//   (e) tt': fl() of the double sum and the dropped fourth piece, 2 u r^2                                         =   2   u N
//   (f) the two float32 additions that form s from the MFMA result (|s| <= 2 N)                                   =   4   u N
//   (g) the matrix unit's accumulation.  Sum of |terms| T: cross terms <= (1 + 2^-7)^2 2 |qc| |tc| <= 1.016 N,
//       tt' pieces <= 1.008 r^2, bias <= 1.008 |qc|^2 + 2^-14 N: T <= 2.03 N.  WORST-CASE MODEL: the 16 products are
//       exact (8-bit x 8-bit significands) and are added in any order, in float32 or wider, each addition - or each
//       operand aligned to the largest exponent - rounded OR TRUNCATED: at most 16 errors of one float32 ulp of a
//       quantity <= T, i.e. 16 * 2u * T                                                                           =  65   u N
//       (tools/probe_mfma_bf16.hip measures <= 4.2 u T for the whole sum on 2*10^6 cancelling contractions:
//       profiles/r01n_probe_mfma_bf16.txt, re-run by tests/test_gpu_mfma_probe.py - 15x inside the model.)
//   (h) the dropped low x low products: |tl| <= 2^-8 |tc| and |q'l| <= 2^-8 |q'| per coordinate (round to nearest), so
//       |sum tl q'l| <= 2^-16 |tc| |q'| = 2^-16 * 2 |tc| |qc'| <= 2^-16 (1 + 2^-15) N                              = 256.1 u N
//   total (1 + e) * ((b) + (c)) + the rest                                                                       < 353.1 u N
// E = 1.75 * 2^-16 N + 2^-120 = 448 u N: 1.27x the worst case, of which the matrix unit's 65 u N are a model the probe sits
// 15x inside and (h) is attained only by coordinates that all sit just above a power of two on both sides.  (Rounds 1-3
// computed the low x low products and shipped 2^-17, then 2^-16 = 2.6x the worst case of that form; a band twice as wide costs
// ~3x the overflow points - a few hundred of 10^6, 0.1 ms at 1M<->1M for 2^-15 - and no scan time.)
// The absolute floor 2^-120 is for the matrix unit flushing subnormal operand pieces, products or sums (17 x 2^-126 at
// most, plus 6 x 2^-126 (|q'| + r) for flushed low pieces, which the relative part covers once |qc| + r >= 2^-106):
// squared distances that small make every granule a candidate and the all-exact scan answers.  A point whose E is not
// finite (overflowing coordinates) publishes L = 0 three times: everything is a candidate for it.
// tests/test_gpu_filter_bounds.py reads the per-granule L back (mi355_icp_debug_filter_bounds, the same kernel
// instantiated with kDump) and asserts L <= the oracle's exact granule minimum for EVERY (point, granule) on the
// adversarial clouds of the parity suite - the invariant itself, not its consequence.
//
// The scan keeps, per source point and wave half, the three granules with the smallest L; the exact resolve
// (resolve_xfilter_kernel) re-scans granule 1 on the original coordinates, granule 2 if L2 <= the exact minimum d*
// found so far, and sends the point to the all-exact scan if L3 <= d* (a granule the scan did not keep could matter).
//
// Geometry.  A wave computes 32 targets x 32 sources per MFMA.  D: lane l holds source column (l & 31) and the 16 rows
// 8j + 4(l >> 5) + k, so each HALF of the wave sees half of the rows; the operand image (xtarget_kernels.hip) assigns
// those rows to one contiguous 128-target run of the sorted copy per 256-target chunk, and each half keeps its own
// keys - no cross-lane traffic at all; the resolve merges the two key rows like two more target splits.
// VALU work per pair: the minimum over the 16 accumulator registers (8 v_min3_u32 per MFMA: 44.8 cycles per MFMA + minima
// at two waves per SIMD, tools/ubench_bf16_min.hip) + ~0.14 slots of bookkeeping.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#if defined(MI355_PIN_OFF) || !defined(__HIP_DEVICE_COMPILE__)  // (the host pass must not see a "v" constraint: there it names an x86 register class)
#define MI355_PIN(frag, acc, cm) ((void)0)
#else
#define MI355_PIN(frag, acc, cm) asm volatile("" : "+v"(frag), "+v"(acc), "+v"(cm))
#endif
__device__ __forceinline__ u32x4 load_frag(const uint4* p) { return *reinterpret_cast<const u32x4*>(p); }

__device__ __forceinline__ void split_bf16(float v, uint32_t& packed, float& value) {
  const __bf16 hb = static_cast<__bf16>(v);
  const float hf = static_cast<float>(hb);
  const __bf16 lb = static_cast<__bf16>(v - hf);
  const float lf = static_cast<float>(lb);
  packed = static_cast<uint32_t>(__builtin_bit_cast(unsigned short, hb)) |
           (static_cast<uint32_t>(__builtin_bit_cast(unsigned short, lb)) << 16);
  value = hf + lf;
}

// Minimum of the 16 accumulators as UNSIGNED integers.  The sixteenth K slot adds a per-source bias b >= |qc'|^2 + 2E,
// so every finite result is a non-negative float (>= E by the bound above) and unsigned order is float order; +inf
// (padding rows) sorts after every finite value and NaNs of either sign after that.  A float minimum would cost two
// issue slots per value here: hipcc has to canonicalise MFMA results before v_min_f32 (IEEE mode), the integer one is
// a single integer op.
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t t = a < b ? a : b;
  return t < c ? t : c;  // v_min3_u32
}
// min(seed, the 16 results): 8 v_min3_u32.  Integer VALU ops take 4 cycles per wave on gfx950 whatever their operand
// count (tools/ubench_bf16_min.hip), so the three-operand form halves the cost of the minimum: 32 cycles per MFMA.
__device__ __forceinline__ uint32_t min16(const f32x16& acc, uint32_t seed) {
  uint32_t b[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) b[k] = __float_as_uint(acc[k]);
  const uint32_t t0 = umin3(b[0], b[1], b[2]), t1 = umin3(b[3], b[4], b[5]), t2 = umin3(b[6], b[7], b[8]);
  const uint32_t t3 = umin3(b[9], b[10], b[11]), t4 = umin3(b[12], b[13], b[14]);
  const uint32_t t5 = umin3(b[15], t0, t1), t6 = umin3(t2, t3, t4);
  return umin3(t5, t6, seed);
}

// kSets: 32-source column sets per wave; kThreads: workgroup size.  A workgroup covers kThreads / 2 * kSets source points:
// <4, 256> (4 waves, 2 workgroups per CU = 2 waves per SIMD at <= 256 registers) or <2, 512> (4 waves per SIMD at <= 128).
// The next tile travels global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers, no
// ds_write).  Round 1 staged it through a register array, which hipcc - out of registers at 224 - promoted to LDS
// ("promote alloca"): every chunk paid global_load -> s_waitcnt vmcnt -> ds_write to a scratch slot -> ds_read ->
// ds_write to the tile, 8 LDS instructions next to the 8 fragment reads that do the work, at a 32-byte lane stride that
// conflicts two ways (SQ_LDS_BANK_CONFLICT = 73 % of SQ_ACTIVE_INST_LDS in profiles/r01o_pmc_summary.txt; 33.6 -> 32.0 ms
// at 1M<->1M without it: profiles/r02b_scan_ab_lds_dma.txt).
// kDump (mi355_icp_debug_filter_bounds only, never the product path): additionally writes the lower bound L of EVERY
// (point, 128-target run) to dump[run * dump_stride + point position] - what the parity tests check the bound on.
// kStaged: the launch is also the staging of a small pass (<= 8 jobs; see nn_scan_kernel): every workgroup reads its
// descriptor from the kernel arguments and the first one writes the device tables for the launches behind it.
#ifdef MI355_ICP_DIAG
// Diagnostic build only (make diag): per-workgroup stamps of the matrix-core scan (100 MHz realtime counter):
// [start, first tile in LDS, tiles walked, end, tiles << 48 | job << 40 | xcc << 32 | hw id, the walk's three quarter points]
// per workgroup (tools/xb_timeline.py)
__device__ unsigned long long* g_diag_xb = nullptr;
#endif

// kLayout: how the launch's workgroups map to (source block, run of tiles) - chosen ONCE, by the launcher, for the whole launch:
//   0  grid (scan_blocks * n_split, jobs): workgroup b walks the static run b % n_split of source block b / n_split
//   1  grid (scan_blocks, jobs, n_split), kStaged only: static runs sized by the class of the workgroup (JobDesc::first_class)
//   2  grid (scan_blocks, jobs, n_split): the n_split workgroups of a source block form a TEAM that draws groups of kXbCentreGroup
//      tiles from the block's ticket (queue[2 * team]) until the target is walked.  Launches of one to four rounds of resident
//      workgroups are decided by their last round and, inside a CU, by the SIMD arbiter serving the older wave first (a tile
//      takes 4.4 us for it and ~12 us for the younger one: tools/xb_timeline.py); with the split as the slow grid index every
//      team has a member in the launch's first round and whoever is served faster simply draws more.  Each member still
//      publishes its own key rows (the top three granules of the tiles IT walked); the resolve merges the rows of a point like
//      so many splits, and since it settles every point with the exact formula, WHICH member walked which tiles - the one thing
//      that now differs from run to run - cannot reach an output (it moves points between "settled by the filter" and "settled
//      by the overflow pass", both exact).  A ticket is drawn one group ahead (the next tile's global->LDS transfer starts a
//      tile early), the last member out clears the team's two words for the next launch.
template <int kSets, int kThreads, bool kDump = false, bool kStaged = false, int kLayout = 0>
__global__ __launch_bounds__(kThreads, 2) void nn_scan_xbf16_kernel(const JobDesc* __restrict__ jobs,
                                                                       const float* __restrict__ sx,
                                                                       const float* __restrict__ sy,
                                                                       const float* __restrict__ sz,
                                                                       const uint4* __restrict__ bf_image,
                                                                       const float4* __restrict__ tile_table,
                                                                       unsigned long long* __restrict__ keys4,
                                                                       uint32_t key_stride, uint32_t rank_stride,
                                                                       float* __restrict__ dump = nullptr, uint32_t dump_stride = 0,
                                                                       StageArgs staged = StageArgs{}, JobDesc* __restrict__ jobs_out = nullptr,
                                                                       RefJobDesc* __restrict__ rjobs_out = nullptr, uint32_t* __restrict__ queue = nullptr) {
  static_assert(kLayout == 0 || kLayout == 2 || kStaged, "the class-weighted layout reads its descriptor from the kernel arguments");
  constexpr bool kQueue = kLayout == 2;
  __shared__ uint32_t q_next;  // kQueue: the group of tiles drawn ahead (written by thread 0, read by all behind a barrier)
  // [buffer][k half][row slot]: 2 x 32 KiB, + slots the fragment prefetch of a tile's last-but-one step runs into (it reads
  // blocks 31 and 32; block 32 does not exist and what is read there is never used: the tile's final step issues no MFMA for it)
  constexpr uint32_t kLdsPad = 96;
  __shared__ uint4 lds[2 * kXbTileVecs + kLdsPad];
#ifdef MI355_ICP_DIAG
  const unsigned long long diag_t0 = wall_clock64();
#endif
  if constexpr (kStaged) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      constexpr uint32_t kJ = sizeof(JobDesc) / 4, kR = sizeof(RefJobDesc) / 4;
      const uint32_t* src_w = reinterpret_cast<const uint32_t*>(staged.jobs);
      uint32_t* dst_w = reinterpret_cast<uint32_t*>(jobs_out);
      for (uint32_t w = threadIdx.x; w < staged.n * kJ; w += kThreads) dst_w[w] = src_w[w];
      if (staged.has_ref) {
        const uint32_t* rs = reinterpret_cast<const uint32_t*>(staged.rjobs);
        uint32_t* rd = reinterpret_cast<uint32_t*>(rjobs_out);
        for (uint32_t w = threadIdx.x; w < staged.n * kR; w += kThreads) rd[w] = rs[w];
      }
    }
  }
  const JobDesc& J = kStaged ? staged.jobs[blockIdx.y] : jobs[blockIdx.y];
  const uint32_t n_split = J.n_split;
  // One-round launches (JobDesc::first_class, self-staged kernel only): the split is the SLOW grid index, so that the
  // workgroups of the launch's first round - one per CU, the older wave on every SIMD, which the SIMD's arbiter serves first -
  // are whole planes of the grid, and a source block's runs of tiles are sized by the class of the workgroup that walks them:
  // both workgroups of a CU then finish together instead of the younger one walking its last third alone (tools/xb_timeline.py)
  constexpr bool classed = kLayout == 1;
  const uint32_t sb = kLayout != 0 ? blockIdx.x : blockIdx.x / n_split;
  const uint32_t sp = kLayout != 0 ? blockIdx.z : blockIdx.x - sb * n_split;
  if (sb >= J.scan_blocks || sp >= n_split) return;
  constexpr uint32_t kScanSrcPerBlock = (kThreads / 64) * 32 * kSets;
  uint32_t tile0 = sp * J.split_q + (sp < J.split_r ? sp : J.split_r);
  uint32_t tile1 = tile0 + J.split_q + (sp < J.split_r ? 1u : 0u);
  // kQueue: the team's two words (ticket, members done) and the number of tile groups there are to draw
  uint32_t* const team = kQueue ? queue + 2 * (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) : nullptr;
  const uint32_t n_groups = (J.tgt_tiles + kXbCentreGroup - 1) / kXbCentreGroup;
  if constexpr (kQueue) {
    if (threadIdx.x == 0) q_next = atomicAdd(team, 1u);
    __syncthreads();
    const uint32_t g0 = __builtin_amdgcn_readfirstlane(q_next);
    __syncthreads();  // everybody has read the first ticket before thread 0 draws the next one
    tile0 = g0 < n_groups ? g0 * kXbCentreGroup : 0u;
    tile1 = g0 < n_groups ? (tile0 + kXbCentreGroup < J.tgt_tiles ? tile0 + kXbCentreGroup : J.tgt_tiles) : 0u;  // empty: nothing left for this member
    if (g0 < n_groups && threadIdx.x == 0) q_next = atomicAdd(team, 1u);  // one group ahead; first read behind the first tile's barrier
  }
  if constexpr (classed) {
    const uint32_t plane = gridDim.x * gridDim.y, at = blockIdx.y * gridDim.x + blockIdx.x;
    uint32_t before = 0, mine = 0, total = 0;
    for (uint32_t k = 0; k < n_split; ++k) {
      const uint32_t w = k * plane + at < J.first_class ? kClassFast : kClassSlow;
      before += k < sp ? w : 0u;
      mine = k == sp ? w : mine;
      total += w;
    }
    tile0 = J.tgt_tiles * before / total;  // the planner keeps tgt_tiles >= 4 n_split: every run holds a tile
    tile1 = J.tgt_tiles * (before + mine) / total;
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool hi = lane >= 32;
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = J.pose[k];
  float q2x[kSets], q2y[kSets], q2z[kSets];  // -2 q (exact scaling): -2 fl(q - c) = fl(-2q + 2c) bit for bit
  float l1[kSets], l2[kSets], l3[kSets];
  uint32_t c1[kSets], c2[kSets], c3[kSets];
  bool unsafe[kSets];  // per point: its bound is not finite somewhere along the walk -> everything is a candidate for it
  const uint32_t src_local = (wave * kSets) * 32 + (lane & 31u);
#pragma unroll
  for (int m = 0; m < kSets; ++m) {
    const uint32_t i = J.src_off + sb * kScanSrcPerBlock + src_local + m * 32;  // padded planes: always in bounds
    float qx, qy, qz;
    transform_point(pose, sx[i], sy[i], sz[i], qx, qy, qz);
    q2x[m] = -2.0f * qx;
    q2y[m] = -2.0f * qy;
    q2z[m] = -2.0f * qz;
    l1[m] = l2[m] = l3[m] = __builtin_inff();
    c1[m] = c2[m] = c3[m] = tile0 * (2 * kXbChunksPerTile);
    unsafe[m] = false;
  }
  const uint32_t tile_base = J.tgt_off / kTile;  // tgt_off is a multiple of kTile
  const uint4* __restrict__ image = bf_image + static_cast<size_t>(tile_base) * kXbTileVecs;
  if (!kQueue || tile0 < tile1) {
#pragma unroll
    for (int k = 0; k < kXbTileVecs / kThreads; ++k)
      lds[k * kThreads + tid] = image[static_cast<size_t>(tile0) * kXbTileVecs + k * kThreads + tid];
  }
  __syncthreads();
#ifdef MI355_ICP_DIAG
  const unsigned long long diag_t1 = wall_clock64();
#endif
  uint32_t cur = 0;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint4 bq[kSets];
  float adj[kSets], ee[kSets];
  // Per-chunk skip test in the RAW domain.  lo() below is monotone in the accumulator value, so there is a smallest value
  // from which the chunk's lower bound reaches l3 and the chunk cannot enter the point's three best; rawthr[m] holds the bits
  // of an UPPER bound R of that value, and "minimum >= R" (one unsigned compare: positive floats order like their bits,
  // +inf and NaN patterns after them) replaces the four float operations of lo() in almost every chunk.
  //   lo = (raw + adj - ee) * 0.99902 >= l3   <=   raw >= l3 / 0.99902 + ee - adj;   R = (l3 * 1.001 + (ee - adj)) * (1 + 2^-20):
  // the 2^-20 (16 u) covers the three roundings of lo() on operands of raw's size and R's own two; l3 = +inf gives R = +inf
  // (only +inf / NaN minima skip, as lo < +inf demands), and ee - adj >= 2 ee > 0 keeps R positive.
  uint32_t rawthr[kSets];
  auto raw_threshold = [](float l3v, float adjv, float eev) -> uint32_t {
    const float x = l3v * 1.001f + (eev - adjv);
    return __float_as_uint(x + x * 9.5367431640625e-07f);
  };
#ifdef MI355_ICP_DIAG
  unsigned long long diag_q[3] = {0, 0, 0};  // the walk's quarter points
#endif
  [[maybe_unused]] uint32_t tiles_walked = 0;  // (diagnostics; kQueue: what this member drew)
  for (uint32_t tile = tile0; tile < tile1;) {
#ifdef MI355_ICP_DIAG
    if constexpr (!kQueue) {
      const uint32_t done = tile - tile0, all = tile1 - tile0;
      if (done == all / 4) diag_q[0] = wall_clock64();
      if (done == all / 2) diag_q[1] = wall_clock64();
      if (done == all - all / 4) diag_q[2] = wall_clock64();
    }
#endif
    ++tiles_walked;
    uint32_t next_tile = tile + 1, next_end = tile1;
    bool more = next_tile < tile1;
    if constexpr (kQueue) {
      if (!more) {  // the group's last tile: on to the group drawn ahead, if the target is not walked yet
        const uint32_t gn = __builtin_amdgcn_readfirstlane(q_next);
        if (gn < n_groups) {
          more = true;
          next_tile = gn * kXbCentreGroup;
          next_end = next_tile + kXbCentreGroup < J.tgt_tiles ? next_tile + kXbCentreGroup : J.tgt_tiles;
        }
      }
    }
    // The operand image of kXbCentreGroup consecutive tiles shares one centre and one r^2 (xtarget_kernels.hip): the
    // source points are re-split into bf16 pieces at the first tile of the walk and whenever a new group begins
    if (tile == tile0 || tile % kXbCentreGroup == 0) {
    const float4 tc = tile_table[tile_base + tile];  // the group's centre xyz and r^2 (wave-uniform)
    const float c2x = 2.0f * tc.x, c2y = 2.0f * tc.y, c2z = 2.0f * tc.z;
#pragma unroll
    for (int m = 0; m < kSets; ++m) {
      uint32_t px, py, pz;
      float vx, vy, vz;
      split_bf16(q2x[m] + c2x, px, vx);  // q' = -2 fl(q - c) in two bf16 pieces; v = their sum
      split_bf16(q2y[m] + c2y, py, vy);
      split_bf16(q2z[m] + c2z, pz, vz);
      const float qq = 0.25f * __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));  // |qc'|^2 of the quantised point (x 1/4: exact)
      ee[m] = __builtin_fmaf(2.6702880859375e-05f, qq + tc.w, 7.5231638e-37f);  // E = 1.75 * 2^-16 (|qc'|^2 + r^2) + 2^-120
      unsafe[m] = unsafe[m] || !(ee[m] < 1.0e37f);                              // inf / NaN bound: exact scan instead
      const uint32_t bias = (__float_as_uint(__builtin_fmaf(2.0f, ee[m], qq)) + 0xFFFFu) >> 16;  // bf16, rounded up
      adj[m] = qq - bf16_value(bias);  // s = MFMA + adj
      const uint32_t w0 = hi ? pz : px;
      // the slot that would meet the target's low piece with the source's low piece holds zero on both sides (see "low x low" above)
      bq[m] = make_uint4(w0, w0 & 0xFFFFu, hi ? 0x3F803F80u : py, hi ? (0x00003F80u | (bias << 16)) : (py & 0xFFFFu));
      rawthr[m] = raw_threshold(l3[m], adj[m], ee[m]);
    }
    }
    const uint4* __restrict__ A = lds + cur * kXbTileVecs + (hi ? kTile : 0) + (lane & 31u);
    // Two accumulator sets in flight over the tile's 32 MFMA blocks, one MFMA then the 8 integer minima of an EARLIER
    // result, alternating in program order: a wave's next MFMA waits at the vector issue port until the matrix pipe
    // is free (32 cycles), and anything behind it - including the other resident wave's VALU work - waits with it, so
    // MFMAs issued back to back serialise the two pipes.  The loop is kept rolled at one 256-target chunk per
    // iteration (unrolled further, hipcc hoists every MFMA of a tile ahead of the minima and spills their results)
    // and the order is pinned as data flow (MI355_PIN).
    f32x16 acc_a[kSets], acc_b[kSets];
    {
      const uint4 a = A[0];
#pragma unroll
      for (int m = 0; m < kSets; ++m)
        acc_a[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq[m]), zero, 0, 0, 0);
    }
    uint32_t cmin[kSets];
#pragma unroll
    for (int m = 0; m < kSets; ++m) cmin[m] = 0x7F800000u;
    // the whole next tile, asynchronously: the barrier that ended the previous tile says every wave is done reading the
    // other buffer; the __syncthreads() at the end of this tile waits for these (vmcnt(0)) before anyone reads it
    if (more) {
      const uint4* __restrict__ next = image + static_cast<size_t>(next_tile) * kXbTileVecs;
#pragma unroll
      for (int k = 0; k < kXbTileVecs / kThreads; ++k)
        __builtin_amdgcn_global_load_lds(next + k * kThreads + tid, &lds[(cur ^ 1) * kXbTileVecs + k * kThreads + wave * 64], 16, 0, 0);
    }
    // operands of the NEXT step's MFMAs are read one step ahead, into two register sets used alternately (no copies);
    // Ap walks the tile two blocks per step, so the four reads of an iteration sit at immediate offsets from it
    u32x4 f0_b = load_frag(A + 32), f0_a = load_frag(A + 64), f1_b, f1_a;
    const uint4* __restrict__ Ap = A;
    // MI355_PIN: an empty asm that "rewrites" the next MFMA's operand, the accumulator the next minima read and the
    // previous minimum - so step m (one MFMA + 8 minima) can start only when step m - 1's minima are done, which is
    // the alternation the comment above asks for, expressed as data flow the scheduler cannot undo.
    // kLast: the tile's final step.  Its second half would start block 32, which does not exist (the next tile begins with a
    // prologue of its own behind the barrier): only the minima of block 31 are left to do, so 4 of a tile's 132 MFMAs and the
    // fragment reads for blocks 33 / 34 are never issued (27.69 -> 27.0 ms at 1M<->1M together with the placement of the
    // fragment reads below: profiles/r03h_scan_ab_peel.txt).
    // The fragment reads of a step sit in its MIDDLE: at its head hipcc waits for them with lgkmcnt(0) - the step's own reads
    // included, an exposed LDS round trip twice per chunk - while from the middle they are four MFMA groups old when the
    // next step waits.
    auto step = [&](auto last_tag, u32x4& use_b, u32x4& use_a, u32x4& next_b, u32x4& next_a) {
      constexpr bool kLast = decltype(last_tag)::value;
#pragma unroll
      for (int m = 0; m < kSets; ++m) {
        MI355_PIN(use_b, acc_a[m], cmin[(m + kSets - 1) % kSets]);
        acc_b[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, use_b), __builtin_bit_cast(bf16x8, bq[m]), zero, 0, 0, 0);
        cmin[m] = min16(acc_a[m], cmin[m]);
      }
      if constexpr (!kLast) {
        next_b = load_frag(Ap + 96);   // blocks 2p + 3 and 2p + 4 of step p (past the tile's end on step 14: see kLdsPad)
        next_a = load_frag(Ap + 128);
        Ap += 64;
      }
#pragma unroll
      for (int m = 0; m < kSets; ++m) {
        MI355_PIN(use_a, acc_b[m], cmin[(m + kSets - 1) % kSets]);
        if constexpr (!kLast)
          acc_a[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, use_a), __builtin_bit_cast(bf16x8, bq[m]), zero, 0, 0, 0);
        cmin[m] = min16(acc_b[m], cmin[m]);
      }
    };
    auto finish_chunk = [&](uint32_t c) {
      const uint32_t run = ((tile * kXbChunksPerTile + c) << 1) | (hi ? 1u : 0u);
#pragma unroll
      for (int m = 0; m < kSets; ++m) {
        const uint32_t raw = cmin[m];
        cmin[m] = 0x7F800000u;
        // once the scan has passed a point's neighbourhood its three best granules rarely change: unless some lane of the
        // wave holds a minimum below its raw threshold, the chunk is done with one compare
        if (!kDump && __builtin_amdgcn_ballot_w64(raw < rawthr[m]) == 0) continue;
        const float s = __uint_as_float(raw) + adj[m];
        const float lo = __builtin_fmaxf((s - ee[m]) * 0.99902f, 0.0f);
        if constexpr (kDump)  // NaN (a non-finite bound) is reported as 0: "no statement", like the keys of an unsafe point
          dump[static_cast<size_t>(run) * dump_stride + J.out_off + sb * kScanSrcPerBlock + src_local + m * 32] = (unsafe[m] || !(lo == lo)) ? 0.0f : lo;
        // the sorted insert (12 selects) runs only if some lane needs it (rarely, with the sources in Morton order too)
        if (__builtin_amdgcn_ballot_w64(lo < l3[m]) != 0) {
          const float o1 = l1[m], o2 = l2[m], o3 = l3[m];
          const uint32_t i1 = c1[m], i2 = c2[m], i3 = c3[m];
          const bool d1 = lo < o1, d2 = lo < o2, d3 = lo < o3;
          const float keep3 = d3 ? lo : o3, keep2 = d2 ? lo : o2;
          const uint32_t keepi3 = d3 ? run : i3, keepi2 = d2 ? run : i2;
          l3[m] = d2 ? o2 : keep3;
          c3[m] = d2 ? i2 : keepi3;
          l2[m] = d1 ? o1 : keep2;
          c2[m] = d1 ? i1 : keepi2;
          l1[m] = d1 ? lo : o1;
          c1[m] = d1 ? run : i1;
          rawthr[m] = raw_threshold(l3[m], adj[m], ee[m]);
        }
      }
    };
    constexpr std::false_type kMid{};
    constexpr std::true_type kEnd{};
#pragma unroll 1
    for (uint32_t c = 0; c + 1 < kXbChunksPerTile; ++c) {  // a 256-target chunk = 8 blocks = 4 steps
      step(kMid, f0_b, f0_a, f1_b, f1_a);
      step(kMid, f1_b, f1_a, f0_b, f0_a);
      step(kMid, f0_b, f0_a, f1_b, f1_a);
      step(kMid, f1_b, f1_a, f0_b, f0_a);
      finish_chunk(c);
    }
    {  // the tile's last chunk, peeled for its final step
      step(kMid, f0_b, f0_a, f1_b, f1_a);
      step(kMid, f1_b, f1_a, f0_b, f0_a);
      step(kMid, f0_b, f0_a, f1_b, f1_a);
      step(kEnd, f1_b, f1_a, f0_b, f0_a);
      finish_chunk(kXbChunksPerTile - 1);
    }
    __syncthreads();
    cur ^= 1;
    if constexpr (kQueue) {
      if (!more) break;
      if (next_tile != tile + 1) {  // a new group begins: draw the one behind it (everybody read q_next before the barrier above)
        if (threadIdx.x == 0) q_next = atomicAdd(team, 1u);
        if (next_end - next_tile == 1) __syncthreads();  // a one-tile group reads the ticket at once: no tile barrier in between
      }
      tile1 = next_end;
    }
    tile = next_tile;
  }
  if constexpr (kQueue) {
    // last member out clears the team's words for the next launch (every member has made its last draw by then)
    if (threadIdx.x == 0 && atomicAdd(team + 1, 1u) + 1u == n_split) {
      __hip_atomic_store(team, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(team + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef MI355_ICP_DIAG
  const unsigned long long diag_t2 = wall_clock64();
#endif
  // each wave half publishes its own key row: rows 2 sp (lanes 0-31) and 2 sp + 1 (lanes 32-63)
  const size_t out = static_cast<size_t>(2 * sp + (hi ? 1u : 0u)) * key_stride + J.out_off + sb * kScanSrcPerBlock + src_local;
#pragma unroll
  for (int m = 0; m < kSets; ++m) {
    const size_t o = out + m * 32;
    const float w1 = unsafe[m] ? 0.0f : l1[m], w2 = unsafe[m] ? 0.0f : l2[m], w3 = unsafe[m] ? 0.0f : l3[m];
    keys4[o] = (static_cast<unsigned long long>(__float_as_uint(w1)) << 32) | c1[m];
    keys4[rank_stride + o] = (static_cast<unsigned long long>(__float_as_uint(w2)) << 32) | c2[m];
    keys4[2 * static_cast<size_t>(rank_stride) + o] = (static_cast<unsigned long long>(__float_as_uint(w3)) << 32) | c3[m];
  }
#ifdef MI355_ICP_DIAG
  if (g_diag_xb && threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const size_t slot = ((static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;  // grid order
    g_diag_xb[slot + 0] = diag_t0;
    g_diag_xb[slot + 1] = diag_t1;
    g_diag_xb[slot + 2] = diag_t2;
    g_diag_xb[slot + 3] = wall_clock64();
    g_diag_xb[slot + 5] = diag_q[0];
    g_diag_xb[slot + 6] = diag_q[1];
    g_diag_xb[slot + 7] = diag_q[2];
    g_diag_xb[slot + 4] = (static_cast<unsigned long long>(tiles_walked) << 48) | (static_cast<unsigned long long>(blockIdx.y & 0xffu) << 40) |
                           (static_cast<unsigned long long>(xcc & 0xf) << 32) | hwid;
  }
#endif
}

#ifdef MI355_ICP_DIAG
extern "C" int mi355_icp_diag_xb_begin(size_t n_slots) {
  unsigned long long* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), n_slots * 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemset(buf, 0, n_slots * 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_xb), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_xb_read(unsigned long long* out, size_t n_slots) {
  unsigned long long* buf = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_diag_xb), sizeof(buf)) != hipSuccess || !buf) return 1;
  if (hipMemcpy(out, buf, n_slots * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  unsigned long long* none = nullptr;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_xb), &none, sizeof(none));  // later launches stamp nothing
  return hipFree(buf) == hipSuccess ? 0 : 1;
}
#endif

// a workgroup spans 512 source points either way (the planner's 2 points per lane); shape 0 = <4 sets, 256 threads> (default),
// 1 = <2 sets, 512 threads> (1.4 % slower: profiles/r02b_scan_ab_lds_dma.txt)
// layout: XbLayout of the WHOLE launch (the kernel's kLayout) - the one place it is decided is the caller (run_fused_index_phase);
// layouts 1 and 2 launch the 3-D grid (grid3_x source blocks, jobs, grid3_z splits), layout 2 needs `queue` (2 zeroed words per
// (job, source block) of that grid; the launch leaves them zeroed)
void launch_nn_scan_xbf16(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x, DevicePlanes src,
                          const void* bf_image, const void* tile_table, unsigned long long* keys4, uint32_t key_stride,
                          uint32_t rank_stride, int shape, const StageArgs* staged, JobDesc* jobs_out, RefJobDesc* rjobs_out, int layout,
                          uint32_t grid3_x, uint32_t grid3_z, uint32_t* queue) {
  if (!max_blocks_x) return;
  dim3 grid(max_blocks_x, static_cast<uint32_t>(n_jobs));
  if (layout != kXbLayoutStatic) grid = dim3(grid3_x, static_cast<uint32_t>(n_jobs), grid3_z);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint4* image = static_cast<const uint4*>(bf_image);
  const float4* table = static_cast<const float4*>(tile_table);
  if (staged && shape != 1) {  // small passes: this launch stages the descriptors itself
    auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, image, table, keys4, key_stride, rank_stride,
                         nullptr, 0u, *staged, jobs_out, rjobs_out, queue);
    };
    if (layout == kXbLayoutQueue) go(nn_scan_xbf16_kernel<4, 256, false, true, kXbLayoutQueue>);
    else if (layout == kXbLayoutClassed) go(nn_scan_xbf16_kernel<4, 256, false, true, kXbLayoutClassed>);
    else go(nn_scan_xbf16_kernel<4, 256, false, true, kXbLayoutStatic>);
    return;
  }
  if (layout == kXbLayoutQueue && shape != 1) {
    hipLaunchKernelGGL((nn_scan_xbf16_kernel<4, 256, false, false, kXbLayoutQueue>), grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z,
                       image, table, keys4, key_stride, rank_stride, nullptr, 0u, StageArgs{}, nullptr, nullptr, queue);
    return;
  }
  if (shape == 1)
    hipLaunchKernelGGL((nn_scan_xbf16_kernel<2, 512>), grid, dim3(512), 0, static_cast<hipStream_t>(stream), jobs_dev, src.x, src.y,
                       src.z, static_cast<const uint4*>(bf_image), static_cast<const float4*>(tile_table), keys4, key_stride,
                       rank_stride);
  else
    hipLaunchKernelGGL((nn_scan_xbf16_kernel<4, 256>), grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), jobs_dev, src.x,
                       src.y, src.z, static_cast<const uint4*>(bf_image), static_cast<const float4*>(tile_table), keys4, key_stride,
                       rank_stride);
}

// the production kernel instantiated with kDump: one job, unsplit, L of every (point, run) to `dump` (diagnostic entry point)
void launch_nn_scan_xbf16_dump(void* stream, const JobDesc* jobs_dev, uint32_t blocks_x, DevicePlanes src, const void* bf_image,
                               const void* tile_table, unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride,
                               float* dump, uint32_t dump_stride) {
  if (!blocks_x) return;
  hipLaunchKernelGGL((nn_scan_xbf16_kernel<4, 256, true>), dim3(blocks_x, 1), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), jobs_dev,
                     src.x, src.y, src.z, static_cast<const uint4*>(bf_image), static_cast<const float4*>(tile_table), keys4, key_stride,
                     rank_stride, dump, dump_stride);
}

// points_per_lane selects the workgroup's source span exactly as for the VALU kernel (256 * ppl = 128 * kSets points)
void launch_nn_scan_xmfma(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t blocks_x,
                          uint32_t first_block, DevicePlanes src, const float* tcx, const float* tcy, const float* tcz,
                          const float* tt, const void* tile_table, unsigned long long* keys4, uint32_t key_stride,
                          uint32_t rank_stride) {
  if (!blocks_x) return;
  dim3 grid(blocks_x, static_cast<uint32_t>(n_jobs));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* table = static_cast<const float4*>(tile_table);
  if (points_per_lane == 2)
    hipLaunchKernelGGL(nn_scan_xmfma_kernel<4>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tcx, tcy, tcz,
                       tt, table, keys4, key_stride, rank_stride, first_block);
  else
    hipLaunchKernelGGL(nn_scan_xmfma_kernel<8>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tcx, tcy, tcz,
                       tt, table, keys4, key_stride, rank_stride, first_block);
}

void launch_nn_scan_xfilter(void* stream, int points_per_lane, const JobDesc* jobs_dev, int n_jobs, uint32_t max_blocks_x,
                            DevicePlanes src, const float* tcx, const float* tcy, const float* tcz, const float* tt,
                            const void* tile_table, unsigned long long* keys4, uint32_t key_stride, uint32_t rank_stride) {
  dim3 grid(max_blocks_x, static_cast<uint32_t>(n_jobs));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* table = static_cast<const float4*>(tile_table);
  switch (points_per_lane) {
    case 2:
      hipLaunchKernelGGL(nn_scan_xfilter_kernel<2>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tcx, tcy,
                         tcz, tt, table, keys4, key_stride, rank_stride);
      break;
    case 8:
      hipLaunchKernelGGL(nn_scan_xfilter_kernel<8>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tcx, tcy,
                         tcz, tt, table, keys4, key_stride, rank_stride);
      break;
    default:
      hipLaunchKernelGGL(nn_scan_xfilter_kernel<4>, grid, dim3(kBlockThreads), 0, st, jobs_dev, src.x, src.y, src.z, tcx, tcy,
                         tcz, tt, table, keys4, key_stride, rank_stride);
      break;
  }
}

}  // namespace mi355

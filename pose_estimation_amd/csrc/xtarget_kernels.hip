// xtarget_kernels.hip — builds, on the device, the copy of the target clouds the expansion filter scans
// (nn_scan_xfilter_kernel, icp_fused_kernels.hip): points Morton-sorted so that a 1024-point tile is spatially
// compact, then centred on their tile.  Nothing here affects results - the filter's bounds hold for ANY order and
// ANY centre (only the number of candidate chunks, i.e. speed, depends on compactness) - so the sort may use a library
// primitive: rocprim::radix_sort_pairs (stable, hence deterministic).
//
//   bbox_kernel      bounding box of every cloud (the Morton grid): strided workgroups, merged with integer atomics
//   morton_kernel    30-bit code per point (10 bits per axis inside the box)
//   <rocPRIM>        stable radix sort of (code, index)
//   tile_gather_kernel  one workgroup per 1024-point tile: the tile's points in sorted order (the only gather of the build),
//                    the sorted-position -> original-index table, the tile's bounding box
//   centre_kernel    one workgroup per 1024-point tile: tile box -> centre c; tc = fl(t - c) (the float32 subtraction the
//                    bound is written for), tt = fl(|tc|^2) from double, r^2 = max |tc|^2 rounded up; also the sorted
//                    copy of the original coordinates, the sorted-position -> original-index table, and the bf16
//                    operand image of the matrix-core filter (nn_scan_xbf16_kernel): tc quantised to two bfloat16
//                    pieces per coordinate, |tc'|^2 of the QUANTISED point in three, laid out as MFMA A fragments
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "icp_internal.h"

namespace mi355 {

namespace {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

template <int NT>
__device__ __forceinline__ void block_minmax(float (&lo)[3], float (&hi)[3], float (*slo)[3], float (*shi)[3]) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], off, 64));
      hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
    }
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      slo[wave][k] = lo[k];
      shi[wave][k] = hi[k];
    }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = slo[0][k];
    hi[k] = shi[0][k];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
      lo[k] = __builtin_fminf(lo[k], slo[w][k]);
      hi[k] = __builtin_fmaxf(hi[k], shi[w][k]);
    }
  }
}

}  // namespace

// Order-preserving float <-> uint32 keys, so the bounding box can be reduced with integer atomics across workgroups.
__device__ __forceinline__ uint32_t ordered_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// box[cloud] = {lo x, y, z, hi x, y, z} as ordered keys
__global__ void bbox_init_kernel(uint32_t* __restrict__ box, uint32_t n_entries) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_entries) box[i] = (i % 6 < 3) ? 0xFFFFFFFFu : 0u;
}

// grid = (chunks, n_clouds), block = 1024: each workgroup reduces a strided share of its cloud and merges it with atomics
__global__ __launch_bounds__(1024) void bbox_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                     const float* __restrict__ pz, const uint32_t* __restrict__ cloud_off,
                                                     const uint32_t* __restrict__ cloud_n, uint32_t* __restrict__ box) {
  __shared__ float slo[16][3], shi[16][3];
  const uint32_t off = cloud_off[blockIdx.y], n = cloud_n[blockIdx.y];
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
    const float p[3] = {px[off + i], py[off + i], pz[off + i]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = __builtin_fminf(lo[k], p[k]);
      hi[k] = __builtin_fmaxf(hi[k], p[k]);
    }
  }
  block_minmax<1024>(lo, hi, slo, shi);
  if (threadIdx.x < 3 && lo[threadIdx.x] <= hi[threadIdx.x]) {  // this workgroup saw at least one point
    atomicMin(&box[6 * blockIdx.y + threadIdx.x], ordered_key(lo[threadIdx.x]));
    atomicMax(&box[6 * blockIdx.y + 3 + threadIdx.x], ordered_key(hi[threadIdx.x]));
  }
}

// one thread per point of one cloud
__global__ __launch_bounds__(kBlockThreads) void morton_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                                const float* __restrict__ pz, uint32_t off, uint32_t n,
                                                                const uint32_t* __restrict__ box, uint32_t* __restrict__ code,
                                                                uint32_t* __restrict__ index) {
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {px[off + i], py[off + i], pz[off + i]};
  uint32_t q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo = ordered_value(box[k]), ext = ordered_value(box[3 + k]) - lo;
    const float f = ext > 0.f ? (p[k] - lo) / ext * 1023.0f : 0.f;
    q[k] = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf(f, 0.f), 1023.f));
  }
  code[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  index[i] = i;
}

// grid = tiles of one cloud, block = 256 (4 points per thread).  order[]: sorted position -> index inside the cloud.
// Writes the sorted copy of the original coordinates (planes 4-6 of xt), perm, the tile's box (tile_box: 8 floats per tile,
// lo in 0-2, hi in 4-6) and clears the slot the group's r^2 is gathered in (bf_table row, word 3: centre_kernel).
__global__ __launch_bounds__(kBlockThreads) void tile_gather_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                                     const float* __restrict__ pz, uint32_t off, uint32_t n,
                                                                     const uint32_t* __restrict__ order, float* __restrict__ xt,
                                                                     size_t stride, uint32_t* __restrict__ perm,
                                                                     float* __restrict__ tile_box, float* __restrict__ bf_table) {
  __shared__ float slo[4][3], shi[4][3];
  const uint32_t t0 = blockIdx.x * kTile;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  float* tcx = xt + off;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t s = t0 + k * kBlockThreads + threadIdx.x;
    if (s < n) {
      const uint32_t src = order[s];
      const float x = px[off + src], y = py[off + src], z = pz[off + src];
      tcx[4 * stride + s] = x;
      tcx[5 * stride + s] = y;
      tcx[6 * stride + s] = z;
      perm[off + s] = src;
      lo[0] = __builtin_fminf(lo[0], x); hi[0] = __builtin_fmaxf(hi[0], x);
      lo[1] = __builtin_fminf(lo[1], y); hi[1] = __builtin_fmaxf(hi[1], y);
      lo[2] = __builtin_fminf(lo[2], z); hi[2] = __builtin_fmaxf(hi[2], z);
    }
  }
  block_minmax<kBlockThreads>(lo, hi, slo, shi);
  if (threadIdx.x == 0) {
    const size_t tile = (off + t0) / kTile;
    float* b = tile_box + 8 * tile;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2];
    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2];
    bf_table[4 * tile + 3] = 0.0f;
  }
}

// grid = tiles of one cloud, block = 256 (4 points per thread), after tile_gather_kernel on the same stream.
__global__ __launch_bounds__(kBlockThreads) void centre_kernel(uint32_t off, uint32_t n, const float* __restrict__ tile_box,
                                                                float* __restrict__ xt, size_t stride, float* __restrict__ table,
                                                                uint4* __restrict__ bf_image, float4* __restrict__ sub_table,
                                                                float* __restrict__ bf_table) {
  __shared__ double sr2[4];
  const uint32_t t0 = blockIdx.x * kTile;
  float* tcx = xt + off;
  float x[4], y[4], z[4];
  bool here[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t s = t0 + k * kBlockThreads + threadIdx.x;
    here[k] = s < n;
    x[k] = y[k] = z[k] = 0.f;
    if (here[k]) {
      x[k] = tcx[4 * stride + s];
      y[k] = tcx[5 * stride + s];
      z[k] = tcx[6 * stride + s];
    }
  }
  // bounding spheres of the 64-point runs (a wave's lanes at one k hold 64 consecutive sorted positions): the pruned
  // k-NN scan skips runs no query of a wave can reach.  An empty run (padding only) sits at 3e38 with radius 0.
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool real = here[k];
    float blo[3] = {real ? x[k] : __builtin_inff(), real ? y[k] : __builtin_inff(), real ? z[k] : __builtin_inff()};
    float bhi[3] = {real ? x[k] : -__builtin_inff(), real ? y[k] : -__builtin_inff(), real ? z[k] : -__builtin_inff()};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int off2 = 32; off2 >= 1; off2 >>= 1) {
        blo[a] = __builtin_fminf(blo[a], __shfl_xor(blo[a], off2, 64));
        bhi[a] = __builtin_fmaxf(bhi[a], __shfl_xor(bhi[a], off2, 64));
      }
    const bool any = blo[0] <= bhi[0];
    const float sx_ = any ? 0.5f * blo[0] + 0.5f * bhi[0] : kTargetPad, sy_ = any ? 0.5f * blo[1] + 0.5f * bhi[1] : kTargetPad,
                sz_ = any ? 0.5f * blo[2] + 0.5f * bhi[2] : kTargetPad;
    double rr = 0.0;
    if (real) {
      const double ex = static_cast<double>(x[k]) - sx_, ey = static_cast<double>(y[k]) - sy_, ez = static_cast<double>(z[k]) - sz_;
      rr = ex * ex + ey * ey + ez * ez;
    }
#pragma unroll
    for (int off2 = 32; off2 >= 1; off2 >>= 1) {
      const double other = __shfl_xor(rr, off2, 64);
      rr = other > rr ? other : rr;
    }
    if ((threadIdx.x & 63u) == 0) {
      const float rf = __uint_as_float(__float_as_uint(static_cast<float>(rr)) + 1u);  // next float up (rr >= 0, finite)
      sub_table[static_cast<size_t>((off + t0) / 64) + k * 4 + (threadIdx.x >> 6)] = make_float4(sx_, sy_, sz_, any ? rf : 0.0f);
    }
  }
  // this tile's box, and the box of its GROUP of kXbCentreGroup tiles (icp_internal.h) - the matrix-core operand image is
  // centred per group: the union of the group's tile boxes (tile_gather_kernel), the same words in every workgroup of the group
  const size_t own_tile = (off + t0) / kTile;
  const uint32_t n_tiles = (n + kTile - 1) / kTile;
  const float* ob = tile_box + 8 * own_tile;
  const float lo[3] = {ob[0], ob[1], ob[2]}, hi[3] = {ob[4], ob[5], ob[6]};
  const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
  const uint32_t group_first = (blockIdx.x / kXbCentreGroup) * kXbCentreGroup;   // first tile of the group, inside the cloud
  float glo[3] = {lo[0], lo[1], lo[2]}, ghi[3] = {hi[0], hi[1], hi[2]};
#pragma unroll
  for (int g = 0; g < kXbCentreGroup; ++g) {
    const uint32_t tg = group_first + g;
    if (tg < n_tiles && tg != blockIdx.x) {
      const float* gb = tile_box + 8 * (static_cast<size_t>(off / kTile) + tg);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        glo[a] = __builtin_fminf(glo[a], gb[a]);
        ghi[a] = __builtin_fmaxf(ghi[a], gb[4 + a]);
      }
    }
  }
  const float bcx = 0.5f * glo[0] + 0.5f * ghi[0], bcy = 0.5f * glo[1] + 0.5f * ghi[1], bcz = 0.5f * glo[2] + 0.5f * ghi[2];
  // |tc'|^2 of a point quantised around the group centre, and its bf16 pieces
  auto quantise = [&](float xx, float yy, float zz, uint32_t (&h)[3], uint32_t (&l)[3], float& bdx, float& bdy, float& bdz) -> double {
    bdx = xx - bcx; bdy = yy - bcy; bdz = zz - bcz;   // the float32 subtractions the bound is written for
    h[0] = bf16_bits(bdx); h[1] = bf16_bits(bdy); h[2] = bf16_bits(bdz);
    l[0] = bf16_bits(bdx - bf16_value(h[0])); l[1] = bf16_bits(bdy - bf16_value(h[1])); l[2] = bf16_bits(bdz - bf16_value(h[2]));
    const double qx = static_cast<double>(bf16_value(h[0])) + bf16_value(l[0]), qy = static_cast<double>(bf16_value(h[1])) + bf16_value(l[1]),
                 qz = static_cast<double>(bf16_value(h[2])) + bf16_value(l[2]);
    return qx * qx + qy * qy + qz * qz;
  };
  double br2 = 0.0;   // max over THIS tile of |tc|^2 and |tc'|^2 around the group centre; the group's is gathered below
  double r2 = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t s = t0 + k * kBlockThreads + threadIdx.x;
    if (here[k]) {
      const float dx = x[k] - cx, dy = y[k] - cy, dz = z[k] - cz;  // the float32 subtractions the bound is written for
      const double n2 = static_cast<double>(dx) * dx + static_cast<double>(dy) * dy + static_cast<double>(dz) * dz;
      tcx[s] = dx;
      tcx[stride + s] = dy;
      tcx[2 * stride + s] = dz;
      tcx[3 * stride + s] = static_cast<float>(n2);
      r2 = n2 > r2 ? n2 : r2;
      // matrix-core operand image.  Row slot of sorted position w inside its tile: chunk c = w / 256; the chunk's two
      // 128-target runs belong to the two halves of the wave (hh); within a run, 16 consecutive targets are the rows
      // 8j + 4hh + k (j, k < 4) of MFMA block blk - exactly the rows whose results land in lanes of half hh.
      const uint32_t w = k * kBlockThreads + threadIdx.x;
      const uint32_t cc = w >> 8, hh = (w >> 7) & 1u, u = w & 127u;
      const uint32_t slot = (cc * 8 + (u >> 4)) * 32 + 8 * ((u >> 2) & 3u) + 4 * hh + (u & 3u);
      uint32_t hq[3], lq[3];
      float bdx, bdy, bdz;
      const double n2q = quantise(x[k], y[k], z[k], hq, lq, bdx, bdy, bdz);   // around the GROUP centre
      const uint32_t hx = hq[0], hy = hq[1], hz = hq[2], lx = lq[0], ly = lq[1], lz = lq[2];
      const double n2b = static_cast<double>(bdx) * bdx + static_cast<double>(bdy) * bdy + static_cast<double>(bdz) * bdz;
      br2 = n2b > br2 ? n2b : br2;
      br2 = n2q > br2 ? n2q : br2;
      const float ttq = static_cast<float>(n2q);
      const uint32_t t1 = bf16_bits(ttq);
      const float rem1 = ttq - bf16_value(t1);
      const uint32_t t2 = bf16_bits(rem1);
      const uint32_t t3 = bf16_bits(rem1 - bf16_value(t2));
      uint4* img = bf_image + static_cast<size_t>((off + t0) / kTile) * kXbTileVecs;
      // slots 3, 7 and 11 (low piece x low piece) are zero: see nn_scan_xbf16_kernel
      img[slot] = make_uint4(hx | (hx << 16), lx, hy | (hy << 16), ly);
      img[kTile + slot] = make_uint4(hz | (hz << 16), lz, t1 | (t2 << 16), t3 | 0x3F800000u);  // slot 15: 1.0 (bias row)
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(r2, o, 64);
    r2 = other > r2 ? other : r2;
  }
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) sr2[threadIdx.x >> 6] = r2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = sr2[0];
    for (int w = 1; w < 4; ++w) m = sr2[w] > m ? sr2[w] : m;
    const float r = __uint_as_float(__float_as_uint(static_cast<float>(m)) + 1u);  // next float up: r^2 >= every |tc|^2 (m >= 0, finite)
    float* row = table + 4 * static_cast<size_t>((off + t0) / kTile);
    row[0] = cx; row[1] = cy; row[2] = cz; row[3] = r;
  }
  // the group's r^2 for the matrix-core image (every workgroup of the group computes the same value)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(br2, o, 64);
    br2 = other > br2 ? other : br2;
  }
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) sr2[threadIdx.x >> 6] = br2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = sr2[0];
    for (int w = 1; w < 4; ++w) m = sr2[w] > m ? sr2[w] : m;
    const float r = __uint_as_float(__float_as_uint(static_cast<float>(m)) + 1u);
    float* row = bf_table + 4 * own_tile;
    row[0] = bcx; row[1] = bcy; row[2] = bcz;
    // the group's r^2 = the largest of its tiles': every workgroup of the group raises the slot of EVERY tile of the group
    // (cleared by tile_gather_kernel; non-negative floats order like their bits; rounding to float and the step up are
    // monotone, so the largest rounded value is the rounded largest value - what one workgroup walking the whole group
    // wrote in rounds 1-3)
#pragma unroll
    for (int g = 0; g < kXbCentreGroup; ++g) {
      const uint32_t tg = group_first + g;
      if (tg < n_tiles) atomicMax(reinterpret_cast<uint32_t*>(bf_table + 4 * (static_cast<size_t>(off / kTile) + tg) + 3), __float_as_uint(r));
    }
  }
}

// fills the padding of all seven planes and the perm table before the per-cloud kernels run
// ... and clears the two per-tile tables (four words per tile each: tiles the builder never writes stay r^2 = 0 at the origin)
__global__ __launch_bounds__(kBlockThreads) void xt_fill_kernel(float* __restrict__ xt, size_t stride, uint32_t* __restrict__ perm,
                                                                 uint4* __restrict__ bf_image, float* __restrict__ table,
                                                                 float* __restrict__ bf_table) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kBlockThreads + threadIdx.x;
  if (i >= stride) return;
  if (i < 4 * (stride / kTile)) {
    table[i] = 0.f;
    bf_table[i] = 0.f;
  }
  // padding row of the matrix-core image: |tc'|^2 = +inf (bf16 0x7F80), everything else zero
  const size_t tile = i / kTile, slot = i % kTile;
  bf_image[tile * kXbTileVecs + slot] = make_uint4(0u, 0u, 0u, 0u);
  bf_image[tile * kXbTileVecs + kTile + slot] = make_uint4(0u, 0u, 0x7F80u, 0x3F800000u);
  xt[i] = 0.f;
  xt[stride + i] = 0.f;
  xt[2 * stride + i] = 0.f;
  xt[3 * stride + i] = __builtin_inff();
  xt[4 * stride + i] = kTargetPad;
  xt[5 * stride + i] = kTargetPad;
  xt[6 * stride + i] = kTargetPad;
  perm[i] = 0u;
}

// one thread per super-tile: a sphere around 16 consecutive tile spheres (centre = mean of their centres, radius = the
// farthest reach of any of them, rounded up).  Tiles the builder never wrote (r^2 = 0 at the origin) only widen it.
__global__ __launch_bounds__(kBlockThreads) void super_tile_kernel(const float4* __restrict__ table, uint32_t n_tiles,
                                                                    float4* __restrict__ super_table) {
  const uint32_t s = blockIdx.x * kBlockThreads + threadIdx.x;
  const uint32_t t0 = s * kSuperTiles;
  if (t0 >= n_tiles) return;
  const uint32_t t1 = t0 + kSuperTiles < n_tiles ? t0 + kSuperTiles : n_tiles;
  double cx = 0, cy = 0, cz = 0;
  for (uint32_t t = t0; t < t1; ++t) {
    cx += table[t].x;
    cy += table[t].y;
    cz += table[t].z;
  }
  const double inv = 1.0 / static_cast<double>(t1 - t0);
  const float fx = static_cast<float>(cx * inv), fy = static_cast<float>(cy * inv), fz = static_cast<float>(cz * inv);
  double reach = 0.0;
  for (uint32_t t = t0; t < t1; ++t) {
    const double ex = static_cast<double>(table[t].x) - fx, ey = static_cast<double>(table[t].y) - fy, ez = static_cast<double>(table[t].z) - fz;
    const double r = __builtin_sqrt(ex * ex + ey * ey + ez * ez) + __builtin_sqrt(static_cast<double>(table[t].w));
    reach = r > reach ? r : reach;
  }
  const double r2 = reach * reach * (1.0 + 1e-6);
  super_table[s] = make_float4(fx, fy, fz, __uint_as_float(__float_as_uint(static_cast<float>(r2)) + 1u));
}

void launch_super_tiles(void* stream, const float* table, uint32_t n_tiles_total, float* super_table) {
  const uint32_t n_super = (n_tiles_total + kSuperTiles - 1) / kSuperTiles;
  if (!n_super) return;
  hipLaunchKernelGGL(super_tile_kernel, dim3((n_super + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(table), n_tiles_total,
                     reinterpret_cast<float4*>(super_table));
}

int xtarget_sort_temp_bytes(uint32_t n, size_t* bytes) {
  uint32_t* k = nullptr;
  size_t tmp = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp, k, k, k, k, n, 0, 30, static_cast<hipStream_t>(nullptr), false) != hipSuccess) return 1;
  *bytes = tmp;
  return 0;
}

// Enqueues the whole build for one cloud on `stream`.  code/index/code2/order: n entries each; temp: sort scratch.
// ---- sorted copy of a PYRAMID LEVEL without a sort of its own ------------------------------------------------------------
// Level l of a cloud is rows first, first + step, first + 2 step, ... of level 0, so its points in the Morton order of the
// level-0 sorted copy are that copy's order with everything else left out: an order-preserving compaction of one uint32
// array (level-0 sorted position -> index) into (level sorted position -> level index).  Any spatially coherent order
// serves the filter and the pruned scans equally - results never depend on it, the tile spheres are computed from the points
// actually in the tile - so the level needs neither bounding box nor codes nor the eight launches of a radix sort; a new
// scene pays for those once per request instead of once per matrix-core level.
constexpr uint32_t kOrderItems = 16;  // consecutive parent positions per thread
__device__ __forceinline__ bool level_member(uint32_t parent_index, uint32_t first, uint32_t step, uint32_t n_level, uint32_t* level_index) {
  if (parent_index < first) return false;
  const uint32_t rel = parent_index - first;
  *level_index = rel / step;
  return rel % step == 0 && *level_index < n_level;
}
// grid = ceil(n_parent / (256 * kOrderItems))
__global__ __launch_bounds__(kBlockThreads) void level_order_count_kernel(const uint32_t* __restrict__ parent_perm, uint32_t n_parent,
                                                                           uint32_t first, uint32_t step, uint32_t n_level,
                                                                           uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wave_tot[kBlockThreads / 64];
  const uint32_t base = (blockIdx.x * kBlockThreads + threadIdx.x) * kOrderItems;
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < kOrderItems; ++k) {
    uint32_t li;
    if (base + k < n_parent && level_member(parent_perm[base + k], first, step, n_level, &li)) ++mine;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if ((threadIdx.x & 63u) == 0) wave_tot[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kBlockThreads / 64; ++w) t += wave_tot[w];
    block_counts[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kBlockThreads) void level_order_write_kernel(const uint32_t* __restrict__ parent_perm, uint32_t n_parent,
                                                                           uint32_t first, uint32_t step, uint32_t n_level,
                                                                           const uint32_t* __restrict__ block_counts,
                                                                           uint32_t* __restrict__ order) {
  __shared__ uint32_t wave_tot[kBlockThreads / 64];
  __shared__ uint32_t block_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // members in the workgroups before this one
  uint32_t before = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kBlockThreads) before += block_counts[b];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) wave_tot[wave] = before;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kBlockThreads / 64; ++w) t += wave_tot[w];
    block_base = t;
  }
  __syncthreads();
  const uint32_t base = (blockIdx.x * kBlockThreads + threadIdx.x) * kOrderItems;
  uint32_t li[kOrderItems];
  uint32_t flags = 0, mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < kOrderItems; ++k) {
    li[k] = 0;
    if (base + k < n_parent && level_member(parent_perm[base + k], first, step, n_level, &li[k])) {
      flags |= 1u << k;
      ++mine;
    }
  }
  uint32_t incl = mine;  // exclusive scan of `mine` over the workgroup, in thread order
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += up;
  }
  __syncthreads();  // wave_tot is reused
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t at = block_base + incl - mine;
  for (uint32_t w = 0; w < wave; ++w) at += wave_tot[w];
#pragma unroll
  for (uint32_t k = 0; k < kOrderItems; ++k)
    if (flags & (1u << k)) order[at++] = li[k];
}

// the two launches that turn an order into the sorted copy: tile_box = 8 floats per tile of the whole set (scratch)
static void launch_centre(hipStream_t st, DevicePlanes planes, uint32_t off, uint32_t n, const uint32_t* order, float* xt, size_t stride,
                          uint32_t* perm, float* table, void* bf_image, void* sub_table, float* bf_table, float* tile_box) {
  const dim3 grid((n + kTile - 1) / kTile), block(kBlockThreads);
  hipLaunchKernelGGL(tile_gather_kernel, grid, block, 0, st, planes.x, planes.y, planes.z, off, n, order, xt, stride, perm, tile_box, bf_table);
  hipLaunchKernelGGL(centre_kernel, grid, block, 0, st, off, n, tile_box, xt, stride, table, static_cast<uint4*>(bf_image),
                     static_cast<float4*>(sub_table), bf_table);
}

// The level's copy from its parent's order: `parent_perm` = the level-0 sorted copy's table for this cloud (n_parent
// entries), `first`/`step` = which level-0 rows the level holds.  scratch: n + ceil(n_parent / 4096) words.
int xtarget_build_cloud_from_parent(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, const uint32_t* parent_perm,
                                    uint32_t n_parent, uint32_t first, uint32_t step, uint32_t* scratch, float* xt, size_t stride,
                                    uint32_t* perm, float* table, void* bf_image, void* sub_table, float* bf_table, float* tile_box) {
  if (!n) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t blocks = (n_parent + kBlockThreads * kOrderItems - 1) / (kBlockThreads * kOrderItems);
  uint32_t* order = scratch;
  uint32_t* counts = scratch + n;
  hipLaunchKernelGGL(level_order_count_kernel, dim3(blocks), dim3(kBlockThreads), 0, st, parent_perm, n_parent, first, step, n, counts);
  hipLaunchKernelGGL(level_order_write_kernel, dim3(blocks), dim3(kBlockThreads), 0, st, parent_perm, n_parent, first, step, n, counts, order);
  launch_centre(st, planes, off, n, order, xt, stride, perm, table, bf_image, sub_table, bf_table, tile_box);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int xtarget_build_cloud(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, const uint32_t* box, uint32_t* code,
                        uint32_t* index, uint32_t* code2, uint32_t* order, void* temp, size_t temp_bytes, float* xt,
                        size_t stride, uint32_t* perm, float* table, void* bf_image, void* sub_table, float* bf_table, float* tile_box) {
  if (!n) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(morton_kernel, dim3((n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, st, planes.x, planes.y,
                     planes.z, off, n, box, code, index);
  size_t tb = temp_bytes;
  if (rocprim::radix_sort_pairs(temp, tb, code, code2, index, order, n, 0, 30, st, false) != hipSuccess) return 1;
  launch_centre(st, planes, off, n, order, xt, stride, perm, table, bf_image, sub_table, bf_table, tile_box);
  return 0;
}

void launch_bbox(void* stream, DevicePlanes planes, const uint32_t* cloud_off, const uint32_t* cloud_n, int n_clouds, uint32_t max_n,
                 uint32_t* box) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t entries = 6u * static_cast<uint32_t>(n_clouds);
  hipLaunchKernelGGL(bbox_init_kernel, dim3((entries + 255) / 256), dim3(256), 0, st, box, entries);
  const uint32_t chunks = std::min<uint32_t>(128u, std::max<uint32_t>(1u, (max_n + 8191u) / 8192u));  // >= 8 points per thread
  hipLaunchKernelGGL(bbox_kernel, dim3(chunks, static_cast<uint32_t>(n_clouds)), dim3(1024), 0, st, planes.x, planes.y, planes.z,
                     cloud_off, cloud_n, box);
}

void launch_xt_fill(void* stream, float* xt, size_t stride, uint32_t* perm, void* bf_image, float* table, float* bf_table) {
  hipLaunchKernelGGL(xt_fill_kernel, dim3(static_cast<uint32_t>((stride + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), xt, stride, perm, static_cast<uint4*>(bf_image), table, bf_table);
}

}  // namespace mi355

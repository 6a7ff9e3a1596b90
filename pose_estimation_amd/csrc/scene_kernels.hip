// scene_kernels.hip — scene preparation on the GPU: the point filters the reference runs
// between the camera frame and the ICP, as order-preserving stream compaction.
//
// Reference (its OWN code this time, pose_estimation/src/pose_estimation.cpp):
//   :248      pcl::removeNaNFromPointCloud        keep points whose x, y, z are all finite
//   :347-372  filter_points                        d = sqrt(dx*dx + dy*dy + dz*dz) <= radius, float32;
//                                                  kept inside or removed inside (ExtractIndices negative)
//   :310-333  remove_planes, distance band         d = (x*a + y*b + z*c + dd) / sqrt(x*x + y*y + z*z), float32
//                                                  (it divides by |p|, not |(a,b,c)| - reproduced);
//                                                  points with |d| <= 0.005 (double compare) are removed
// Those loops run serialised under `omp critical` and emit indices in thread-arrival order; this kernel emits the
// kept points in their original order (the order the reference produces when run on one thread).
// Plane coefficients come from the caller (the RANSAC fit, pose_estimation.cpp:285-297, is PCL's and stays on the host).
//
// One pass over HBM: 12 B read per point, 12 B written per kept point.  Each 1024-point tile evaluates the
// predicates, scans its keep flags, and gets its output offset by DECOUPLED LOOK-BACK over 8-byte
// {status, count} granules published with agent-scope atomics (one self-contained 64-bit word per tile, the only
// cross-workgroup data; MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility", granule form R2).
// Tile numbers are drawn from an atomic counter, so a tile only ever waits for tiles that already run: no
// dependence on dispatch order or XCD placement.  Every spin is bounded; a timeout raises the error flag.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_internal.h"

namespace mi355 {

namespace {

constexpr uint32_t kTilePts = kSceneTile;  // 1024 = 256 threads x 4 points
constexpr unsigned long long kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kValueMask = (1ull << 62) - 1;
constexpr int kSpinLimit = 1 << 22;

__device__ __forceinline__ bool keep_point(float x, float y, float z, const SceneFilterSet& F) {
  bool keep = true;
#pragma unroll 1
  for (int k = 0; k < F.n; ++k) {
    const SceneFilter& f = F.f[k];
    if (f.kind == MI355_SCENE_FILTER_FINITE) {
      keep = keep && __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
    } else if (f.kind == MI355_SCENE_FILTER_SPHERE_KEEP_INSIDE || f.kind == MI355_SCENE_FILTER_SPHERE_REMOVE_INSIDE) {
      const float dx = f.p[0] - x, dy = f.p[1] - y, dz = f.p[2] - z;
      float s = dx * dx;
      const float t = dy * dy, u = dz * dz;
      s = s + t;
      s = s + u;
      const float d = __builtin_sqrtf(s);
      const bool inside = d <= f.p[3];
      keep = keep && (f.kind == MI355_SCENE_FILTER_SPHERE_KEEP_INSIDE ? inside : !inside);
    } else {  // MI355_SCENE_FILTER_PLANE_BAND_REMOVE
      float num = x * f.p[0];
      const float t = y * f.p[1], u = z * f.p[2];
      num = num + t;
      num = num + u;
      num = num + f.p[3];
      float den = x * x;
      const float v = y * y, w = z * z;
      den = den + v;
      den = den + w;
      const float d = num / __builtin_sqrtf(den);
      const bool near = static_cast<double>(__builtin_fabsf(d)) <= static_cast<double>(f.p[4]);
      keep = keep && !near;
    }
  }
  return keep;
}

}  // namespace

// in: packed AoS xyz (12 B / point).  out: packed AoS xyz, kept points in original order.
// state: one u64 per tile, zeroed before the launch.  counters[0]: tile ticket, counters[1]: kept total (written
// by the last tile), counters[2]: error flag.
__global__ __launch_bounds__(kBlockThreads) void scene_compact_kernel(const float* __restrict__ in, uint32_t n,
                                                                       SceneFilterSet F, float* __restrict__ out,
                                                                       unsigned long long* __restrict__ state,
                                                                       uint32_t* __restrict__ counters,
                                                                       uint32_t n_tiles) {
  __shared__ uint32_t s_tile, s_excl;
  __shared__ uint32_t s_wave[kBlockThreads / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(&counters[0], 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  if (tile >= n_tiles) return;

  // 4 consecutive points per thread = 48 contiguous bytes = 3 x 16-byte loads
  const uint32_t first = tile * kTilePts + tid * 4;
  float v[12];
  if (first + 4 <= n) {
    const float4* p = reinterpret_cast<const float4*>(in + static_cast<size_t>(first) * 3);
    const float4 a = p[0], b = p[1], c = p[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const size_t idx = static_cast<size_t>(first) * 3 + k;
      v[k] = idx < static_cast<size_t>(n) * 3 ? in[idx] : 0.0f;
    }
  }
  bool keep[4];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    keep[k] = (first + k < n) && keep_point(v[3 * k], v[3 * k + 1], v[3 * k + 2], F);
    mine += keep[k] ? 1u : 0u;
  }
  // exclusive scan of `mine` over the 256 threads
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kBlockThreads / 64; ++w) {
    if (w < wave) base += s_wave[w];
    total += s_wave[w];
  }
  const uint32_t offset_in_tile = base + incl - mine;

  // decoupled look-back by wave 0: lane l inspects tile (tile - 1 - l) of the current 64-tile window
  if (wave == 0) {
    uint32_t exclusive = 0;
    bool failed = false;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&state[0], kFlagPrefix | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&state[tile], kFlagAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t window = static_cast<int64_t>(tile) - 1;
      bool done = false;
      while (!done) {
        const int64_t j = window - lane;
        unsigned long long s = kFlagPrefix;  // tiles before 0 count as an empty prefix
        if (j >= 0) {
          int spins = 0;
          do {
            s = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s >> 62) break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < kSpinLimit);
          if (!(s >> 62)) failed = true;
        }
        const unsigned long long has_prefix = __ballot((s >> 62) == 2ull);
        const int stop = has_prefix ? __ffsll(static_cast<long long>(has_prefix)) - 1 : 63;
        uint32_t contrib = (static_cast<int>(lane) <= stop && j >= 0) ? static_cast<uint32_t>(s & kValueMask) : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) contrib += __shfl_xor(contrib, off, 64);
        exclusive += contrib;
        done = has_prefix != 0ull || __any(failed);
        window -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&state[tile], kFlagPrefix | (static_cast<unsigned long long>(exclusive) + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__any(failed) && lane == 0) atomicOr(&counters[2], 1u);
    if (lane == 0) {
      s_excl = exclusive;
      if (tile == n_tiles - 1) counters[1] = exclusive + total;
    }
  }
  __syncthreads();
  uint32_t o = s_excl + offset_in_tile;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (keep[k]) {
      float* dst = out + static_cast<size_t>(o) * 3;
      dst[0] = v[3 * k];
      dst[1] = v[3 * k + 1];
      dst[2] = v[3 * k + 2];
      ++o;
    }
  }
}

// Packed AoS xyz (n points) -> SoA planes at plane offset `off`, slots [n, n_padded) filled with `pad`.
// Raises counters[2] bit 1 if a non-finite coordinate is met (a target cloud must be finite).
__global__ __launch_bounds__(kBlockThreads) void scene_to_planes_kernel(const float* __restrict__ in, uint32_t n,
                                                                         float* __restrict__ planes, size_t stride,
                                                                         uint32_t off, uint32_t n_padded, float pad,
                                                                         uint32_t* __restrict__ counters) {
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n_padded) return;
  float x = pad, y = pad, z = pad;
  if (i < n) {
    x = in[static_cast<size_t>(i) * 3];
    y = in[static_cast<size_t>(i) * 3 + 1];
    z = in[static_cast<size_t>(i) * 3 + 2];
    if (!(__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z))) atomicOr(&counters[2], 2u);
  }
  planes[off + i] = x;
  planes[stride + off + i] = y;
  planes[2 * stride + off + i] = z;
}

void launch_scene_compact(void* stream, const float* in, uint32_t n, const SceneFilterSet& filters, float* out,
                          unsigned long long* state, uint32_t* counters, uint32_t n_tiles) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(scene_compact_kernel, dim3(n_tiles), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), in, n,
                     filters, out, state, counters, n_tiles);
}

void launch_scene_to_planes(void* stream, const float* in, uint32_t n, float* planes, size_t stride, uint32_t off,
                            uint32_t n_padded, float pad, uint32_t* counters) {
  if (!n_padded) return;
  hipLaunchKernelGGL(scene_to_planes_kernel, dim3((n_padded + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), in, n, planes, stride, off, n_padded, pad, counters);
}

}  // namespace mi355

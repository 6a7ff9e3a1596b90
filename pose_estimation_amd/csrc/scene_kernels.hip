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
// Plane coefficients come from the caller - or from the RANSAC fit further down (mi355_icp_scene_fit_plane), the
// replacement of the pcl::SACSegmentation call at pose_estimation.cpp:285-297.
//
// One pass over HBM: 12 B read per point, 12 B written per kept point.  Each tile (kSceneTile points) evaluates the
// predicates, scans its keep flags, and gets its output offset by DECOUPLED LOOK-BACK over 8-byte
// {status, count} granules published with agent-scope atomics (one self-contained 64-bit word per tile, the only
// cross-workgroup data; MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility", granule form R2).
// Tile numbers are drawn from an atomic counter, so a tile only ever waits for tiles that already run: no
// dependence on dispatch order or XCD placement.  Every spin is bounded; a timeout raises the error flag.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_internal.h"

namespace mi355 {

#ifdef MI355_ICP_DIAG
// Diagnostic build only (make diag): eight 100 MHz time stamps per compaction tile - entry, ticket, tile in LDS, flags ranked,
// prefix ready, look-back done, stores issued, stores complete (tools/scene_timeline.py).  Never in the shipped library.
__device__ unsigned long long* g_diag_scene = nullptr;
#define MI355_SCENE_STAMP(k) do { if (threadIdx.x == 0 && g_diag_scene) diag_t[k] = wall_clock64(); } while (0)
#else
#define MI355_SCENE_STAMP(k) ((void)0)
#endif

namespace {

constexpr unsigned long long kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kValueMask = (1ull << 62) - 1;
constexpr int kSpinLimit = 1 << 22;

// One predicate on one point; `keep` comes in as the AND of the predicates before it.  The filter's parameters are passed as
// VALUES: the caller reads them once per filter - they live in the kernel-argument segment, and fetching them per point (a
// scalar load with a dynamic index, issued and waited for sixteen times per thread) was 5 us of a tile's 18 us of life.
__device__ __forceinline__ bool apply_filter(bool keep, float x, float y, float z, int32_t kind, float p0, float p1, float p2, float p3, float p4) {
  if (kind == MI355_SCENE_FILTER_FINITE) {
    keep = keep && __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
  } else if (kind == MI355_SCENE_FILTER_SPHERE_KEEP_INSIDE || kind == MI355_SCENE_FILTER_SPHERE_REMOVE_INSIDE) {
    const float dx = p0 - x, dy = p1 - y, dz = p2 - z;
    float s = dx * dx;
    const float t = dy * dy, u = dz * dz;
    s = s + t;
    s = s + u;
    const float d = __builtin_sqrtf(s);
    const bool inside = d <= p3;
    keep = keep && (kind == MI355_SCENE_FILTER_SPHERE_KEEP_INSIDE ? inside : !inside);
  } else {  // MI355_SCENE_FILTER_PLANE_BAND_REMOVE
    float num = x * p0;
    const float t = y * p1, u = z * p2;
    num = num + t;
    num = num + u;
    num = num + p3;
    float den = x * x;
    const float v = y * y, w = z * z;
    den = den + v;
    den = den + w;
    const float d = num / __builtin_sqrtf(den);
    const bool near = static_cast<double>(__builtin_fabsf(d)) <= static_cast<double>(p4);
    keep = keep && !near;
  }
  return keep;
}

}  // namespace

// in: packed AoS xyz (12 B / point).  out: packed AoS xyz, kept points in original order.
// state: one u64 per tile, zeroed before the launch.  counters[0]: tile ticket, counters[1]: kept total (written
// by the last tile), counters[2]: error flag.
//
// A tile is kSceneTile points; thread t owns points t, t + kSceneThreads, ... and holds them in REGISTERS from load to store: three
// dword loads per point, 12 bytes apart across the lanes (a wave's x, y and z loads of one k together read every byte of the
// 768 they span; the second and third hit the lines the first brought in), all 48 in flight before the first is used.
// Rounds 1-2 staged the tile through 48 KiB of LDS (lane-contiguous 16-byte loads, then a transposing read): three
// workgroups per CU - and a tile lives ~17 us (stamps: tools/scene_timeline.py) of which 3 are its loads, 5-6 its look-back
// and the rest dependent phases in between, so the bytes in flight, not any unit's throughput, set the rate.  Without the
// LDS tile the workgroups per CU are bounded by registers only, and the tile can grow (icp_internal.h: 12288 points).
// Keep flags are ranked with wave ballots; ONE look-back per tile.
__global__ __launch_bounds__(kSceneThreads, MI355_SCENE_MIN_WAVES) void scene_compact_kernel(const float* __restrict__ in, uint32_t n,
                                                                       SceneFilterSet F, float* __restrict__ out,
                                                                       unsigned long long* __restrict__ state,
                                                                       uint32_t* __restrict__ counters,
                                                                       uint32_t n_tiles) {
  constexpr int kPer = kSceneTile / kSceneThreads;  // points per thread
  constexpr int kWaves = kSceneThreads / 64;
  __shared__ uint32_t s_tile, s_excl;
  __shared__ uint32_t s_cnt[kPer * kWaves];         // kept points per (k, wave), in point order
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef MI355_ICP_DIAG
  unsigned long long diag_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  MI355_SCENE_STAMP(0);
  if (tid == 0) s_tile = atomicAdd(&counters[0], 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  if (tile >= n_tiles) return;
  MI355_SCENE_STAMP(1);

  const size_t tile_first = static_cast<size_t>(tile) * kSceneTile;           // first point of the tile
  float px[kPer], py[kPer], pz[kPer];
  uint32_t keep_mask = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const size_t q = tile_first + static_cast<size_t>(k) * kSceneThreads + tid;
    const bool there = q < n;
    const float* __restrict__ src = in + (there ? q : 0) * 3;
    px[k] = src[0];
    py[k] = src[1];
    pz[k] = src[2];
    keep_mask |= (there ? 1u : 0u) << k;
  }
  MI355_SCENE_STAMP(2);
  // filters outside, the thread's sixteen points inside: a filter's parameters are read once (wave-uniform)
#pragma unroll 1
  for (int fi = 0; fi < F.n; ++fi) {
    const int32_t kind = F.f[fi].kind;
    const float p0 = F.f[fi].p[0], p1 = F.f[fi].p[1], p2 = F.f[fi].p[2], p3 = F.f[fi].p[3], p4 = F.f[fi].p[4];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const bool keep = apply_filter((keep_mask >> k) & 1u, px[k], py[k], pz[k], kind, p0, p1, p2, p3, p4);
      keep_mask = (keep_mask & ~(1u << k)) | ((keep ? 1u : 0u) << k);
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const unsigned long long m = __ballot((keep_mask >> k) & 1u);
    if (lane == 0) s_cnt[k * kWaves + wave] = static_cast<uint32_t>(__popcll(m));
  }
  __syncthreads();
  MI355_SCENE_STAMP(3);
  // exclusive prefix of the kPer * kWaves (k, wave) counts, kEpl per lane: every wave computes the total, wave 0 stores
  uint32_t total;
  {
    constexpr int kEpl = kPer * kWaves / 64;
    static_assert(kEpl * 64 == kPer * kWaves && kEpl >= 1, "whole counts per lane");
    uint32_t c[kEpl], mine = 0;
#pragma unroll
    for (int e = 0; e < kEpl; ++e) {
      c[e] = s_cnt[kEpl * lane + e];
      mine += c[e];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off, 64);
      if (lane >= static_cast<uint32_t>(off)) incl += up;
    }
    total = __shfl(incl, 63, 64);
    __syncthreads();  // everyone has read s_cnt
    if (wave == 0) {
      uint32_t run = incl - mine;
#pragma unroll
      for (int e = 0; e < kEpl; ++e) {
        s_cnt[kEpl * lane + e] = run;  // now the exclusive prefix
        run += c[e];
      }
    }
  }

  MI355_SCENE_STAMP(4);
  // decoupled look-back by wave 0: lane l inspects tile (window - l)
  if (wave == 0) {
    uint32_t exclusive = 0;
    bool failed = false;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&state[0], kFlagPrefix | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&state[tile], kFlagAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // 64 predecessors per round trip.  (Reading 128 or 256 per round - two or four loads per lane in flight - is SLOWER,
      // on the LDS-staged kernel of rounds 1-2 and on this one: profiles/r03i_scene_compact_experiments.txt.)
      int64_t window = static_cast<int64_t>(tile) - 1;
      bool done = false;
      while (!done) {
        const int64_t j = window - lane;
        unsigned long long st = kFlagPrefix;  // tiles before 0 count as an empty prefix
        if (j >= 0) {
          int spins = 0;
          do {
            st = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st >> 62) break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < kSpinLimit);
          if (!(st >> 62)) failed = true;
        }
        const unsigned long long has_prefix = __ballot((st >> 62) == 2ull);
        const int stop = has_prefix ? __ffsll(static_cast<long long>(has_prefix)) - 1 : 63;
        uint32_t contrib = (static_cast<int>(lane) <= stop && j >= 0) ? static_cast<uint32_t>(st & kValueMask) : 0u;
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
  MI355_SCENE_STAMP(5);
  const uint32_t excl = s_excl;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const bool keep = (keep_mask >> k) & 1u;
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const uint32_t o = excl + s_cnt[k * kWaves + wave] + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
      float* dst = out + static_cast<size_t>(o) * 3;
      dst[0] = px[k];
      dst[1] = py[k];
      dst[2] = pz[k];
    }
  }
#ifdef MI355_ICP_DIAG
  MI355_SCENE_STAMP(6);
  __builtin_amdgcn_s_waitcnt(0);
  MI355_SCENE_STAMP(7);
  if (threadIdx.x == 0 && g_diag_scene)
    for (int k = 0; k < 8; ++k) g_diag_scene[static_cast<size_t>(tile) * 8 + k] = diag_t[k];
#endif
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

// Upload path: caller rows (fstride floats each: xyz, optionally + normal + anything) -> SoA planes at plane offset `off`,
// slots [n, n_padded) filled with `pad` (normals: 0).  Sets flags[cloud] if a coordinate or normal is not finite.
__global__ __launch_bounds__(kBlockThreads) void rows_to_planes_kernel(const float* __restrict__ rows, uint32_t fstride, uint32_t n,
                                                                        float* __restrict__ planes, float* __restrict__ normals,
                                                                        size_t stride, uint32_t off, uint32_t n_padded, float pad,
                                                                        uint32_t* __restrict__ flags, uint32_t cloud) {
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n_padded) return;
  float x = pad, y = pad, z = pad, a = 0.f, b = 0.f, c = 0.f;
  if (i < n) {
    const float* r = rows + static_cast<size_t>(i) * fstride;
    x = r[0];
    y = r[1];
    z = r[2];
    bool ok = __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
    if (normals) {
      a = r[3];
      b = r[4];
      c = r[5];
      ok = ok && __builtin_isfinite(a) && __builtin_isfinite(b) && __builtin_isfinite(c);
    }
    if (!ok) atomicOr(&flags[cloud], 1u);
  }
  planes[off + i] = x;
  planes[stride + off + i] = y;
  planes[2 * stride + off + i] = z;
  if (normals) {
    normals[off + i] = a;
    normals[stride + off + i] = b;
    normals[2 * stride + off + i] = c;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// RANSAC plane fit (reference pose_estimation.cpp:285-297 calls pcl::SACSegmentation; definition: include/mi355_icp.h,
// checked against oracle/icp_oracle.c).  Counter-based generator of csrc/synth.c: point index = floor(u01 * n).
namespace {

__device__ __forceinline__ unsigned long long plane_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t plane_draw(unsigned long long seed, unsigned long long ctr, uint32_t n) {
  const unsigned long long x = plane_mix64(seed + ctr * 0x9E3779B97F4A7C15ull);
  const double u = static_cast<double>(x >> 11) * (1.0 / 9007199254740992.0);
  const uint32_t i = static_cast<uint32_t>(u * static_cast<double>(n));
  return i < n ? i : n - 1;
}

}  // namespace

// one thread per hypothesis
// Also clears the support counters of the fit: counts[h] for its hypothesis and the spare one behind the last possible
// hypothesis (the refitted plane's), so that the fit needs no fill launch in front.
__global__ void plane_hypotheses_kernel(const float* __restrict__ scene, uint32_t n, unsigned long long seed, uint32_t n_hyp,
                                        float4* __restrict__ planes, uint32_t* __restrict__ counts) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h == 0) counts[kPlaneMaxHypotheses] = 0;
  if (h >= n_hyp) return;
  counts[h] = 0;
  float4 out = make_float4(0.f, 0.f, 0.f, __builtin_nanf(""));
  for (uint32_t attempt = 0; attempt < 16; ++attempt) {
    const unsigned long long base = (static_cast<unsigned long long>(h) * 16ull + attempt) * 3ull;
    const uint32_t i0 = plane_draw(seed, base, n), i1 = plane_draw(seed, base + 1, n), i2 = plane_draw(seed, base + 2, n);
    if (i0 == i1 || i0 == i2 || i1 == i2) continue;
    const float p0x = scene[3 * static_cast<size_t>(i0)], p0y = scene[3 * static_cast<size_t>(i0) + 1], p0z = scene[3 * static_cast<size_t>(i0) + 2];
    const float ax = scene[3 * static_cast<size_t>(i1)] - p0x, ay = scene[3 * static_cast<size_t>(i1) + 1] - p0y, az = scene[3 * static_cast<size_t>(i1) + 2] - p0z;
    const float bx = scene[3 * static_cast<size_t>(i2)] - p0x, by = scene[3 * static_cast<size_t>(i2) + 1] - p0y, bz = scene[3 * static_cast<size_t>(i2) + 2] - p0z;
    float t, u;
    t = ay * bz; u = az * by; const float nx = t - u;
    t = az * bx; u = ax * bz; const float ny = t - u;
    t = ax * by; u = ay * bx; const float nz = t - u;
    float l2 = nx * nx;
    t = ny * ny; l2 = l2 + t;
    t = nz * nz; l2 = l2 + t;
    const float len = __builtin_sqrtf(l2);
    if (!(len > 0.0f) || !__builtin_isfinite(len)) continue;  // collinear, coincident or non-finite points
    const float a = nx / len, b = ny / len, c = nz / len;
    float d = a * p0x;
    t = b * p0y; d = d + t;
    t = c * p0z; d = d + t;
    out = make_float4(a, b, c, -d);
    break;
  }
  planes[h] = out;
}

// Support of every hypothesis in one pass: a thread takes 4 points; for each plane the wave's inlier ballots are counted
// and one lane adds them to the workgroup's LDS counters, which go to the global counters at the end.  Integer counts.
// planes == nullptr: ONE plane, `one`, handed over as a kernel argument (the refitted plane: no upload in front of the launch).
__global__ __launch_bounds__(kBlockThreads) void plane_support_kernel(const float* __restrict__ scene, uint32_t n,
                                                                       const float4* __restrict__ planes, uint32_t n_hyp, float threshold,
                                                                       uint32_t* __restrict__ counts, float4 one) {
  __shared__ float4 sp[kPlaneMaxHypotheses];
  __shared__ uint32_t sc[kPlaneMaxHypotheses];
  for (uint32_t h = threadIdx.x; h < n_hyp; h += kBlockThreads) {
    sp[h] = planes ? planes[h] : one;
    sc[h] = 0;
  }
  __syncthreads();
  float x[4], y[4], z[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = (blockIdx.x * 4 + k) * kBlockThreads + threadIdx.x;
    const bool in = i < n;
    x[k] = in ? scene[3 * static_cast<size_t>(i)] : __builtin_nanf("");
    y[k] = in ? scene[3 * static_cast<size_t>(i) + 1] : 0.f;
    z[k] = in ? scene[3 * static_cast<size_t>(i) + 2] : 0.f;
  }
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t h = 0; h < n_hyp; ++h) {
    const float4 P = sp[h];  // wave-uniform: LDS broadcast
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float dist = P.x * x[k];
      const float t = P.y * y[k], u = P.z * z[k];
      dist = dist + t;
      dist = dist + u;
      dist = dist + P.w;
      mine += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(dist) <= threshold)));  // NaN: false
    }
    if (lane == 0 && mine) atomicAdd(&sc[h], mine);
  }
  __syncthreads();
  for (uint32_t h = threadIdx.x; h < n_hyp; h += kBlockThreads)
    if (sc[h]) atomicAdd(&counts[h], sc[h]);
}

// n, sum x, y, z, sum xx, xy, xz, yy, yz, zz (double) of one plane's inliers: one partial row per 1024-point block
__global__ __launch_bounds__(kBlockThreads) void plane_moments_kernel(const float* __restrict__ scene, uint32_t n, float4 P, float threshold,
                                                                       double* __restrict__ partials) {
  __shared__ double scratch[kBlockThreads / 64][kPlaneMoments];
  double acc[kPlaneMoments];
#pragma unroll
  for (int k = 0; k < kPlaneMoments; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = (blockIdx.x * 4 + k) * kBlockThreads + threadIdx.x;
    if (i < n) {
      const float fx = scene[3 * static_cast<size_t>(i)], fy = scene[3 * static_cast<size_t>(i) + 1], fz = scene[3 * static_cast<size_t>(i) + 2];
      float dist = P.x * fx;
      const float t = P.y * fy, u = P.z * fz;
      dist = dist + t;
      dist = dist + u;
      dist = dist + P.w;
      if (__builtin_fabsf(dist) <= threshold) {
        const double X = fx, Y = fy, Z = fz;
        acc[0] += 1.0; acc[1] += X; acc[2] += Y; acc[3] += Z;
        acc[4] += X * X; acc[5] += X * Y; acc[6] += X * Z; acc[7] += Y * Y; acc[8] += Y * Z; acc[9] += Z * Z;
      }
    }
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kPlaneMoments; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) scratch[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kPlaneMoments) {
    double r = scratch[0][threadIdx.x];
    for (int w = 1; w < kBlockThreads / 64; ++w) r += scratch[w][threadIdx.x];
    partials[static_cast<size_t>(blockIdx.x) * kPlaneMoments + threadIdx.x] = r;
  }
}

void launch_plane_hypotheses(void* stream, const float* scene, uint32_t n, unsigned long long seed, uint32_t n_hyp, float* planes4,
                             uint32_t* counts) {
  hipLaunchKernelGGL(plane_hypotheses_kernel, dim3((n_hyp + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), scene, n, seed, n_hyp,
                     reinterpret_cast<float4*>(planes4), counts);
}
void launch_plane_support(void* stream, const float* scene, uint32_t n, const float* planes4, uint32_t n_hyp, float threshold, uint32_t* counts) {
  const uint32_t blocks = (n + 4 * kBlockThreads - 1) / (4 * kBlockThreads);
  if (!blocks) return;
  hipLaunchKernelGGL(plane_support_kernel, dim3(blocks), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), scene, n,
                     reinterpret_cast<const float4*>(planes4), n_hyp, threshold, counts, make_float4(0.f, 0.f, 0.f, 0.f));
}
void launch_plane_support_one(void* stream, const float* scene, uint32_t n, const float* plane4, float threshold, uint32_t* count) {
  const uint32_t blocks = (n + 4 * kBlockThreads - 1) / (4 * kBlockThreads);
  if (!blocks) return;
  hipLaunchKernelGGL(plane_support_kernel, dim3(blocks), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), scene, n,
                     static_cast<const float4*>(nullptr), 1u, threshold, count, make_float4(plane4[0], plane4[1], plane4[2], plane4[3]));
}
void launch_plane_moments(void* stream, const float* scene, uint32_t n, const float* plane4, float threshold, double* partials, uint32_t n_blocks) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(plane_moments_kernel, dim3(n_blocks), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), scene, n,
                     make_float4(plane4[0], plane4[1], plane4[2], plane4[3]), threshold, partials);
}

void launch_rows_to_planes(void* stream, const float* rows, uint32_t fstride, uint32_t n, float* planes, float* normals, size_t stride,
                           uint32_t off, uint32_t n_padded, float pad, uint32_t* flags, uint32_t cloud) {
  if (!n_padded) return;
  hipLaunchKernelGGL(rows_to_planes_kernel, dim3((n_padded + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), rows, fstride, n, planes, normals, stride, off, n_padded, pad, flags, cloud);
}

void launch_scene_compact(void* stream, const float* in, uint32_t n, const SceneFilterSet& filters, float* out,
                          unsigned long long* state, uint32_t* counters, uint32_t n_tiles) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(scene_compact_kernel, dim3(n_tiles), dim3(kSceneThreads), 0, static_cast<hipStream_t>(stream), in, n,
                     filters, out, state, counters, n_tiles);
}

#ifdef MI355_ICP_DIAG
extern "C" int mi355_icp_diag_scene_begin(size_t n_tiles) {
  unsigned long long* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), n_tiles * 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemset(buf, 0, n_tiles * 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_scene), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_scene_read(unsigned long long* out, size_t n_tiles) {
  unsigned long long* buf = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_diag_scene), sizeof(buf)) != hipSuccess || !buf) return 1;
  return hipMemcpy(out, buf, n_tiles * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

void launch_scene_to_planes(void* stream, const float* in, uint32_t n, float* planes, size_t stride, uint32_t off,
                            uint32_t n_padded, float pad, uint32_t* counters) {
  if (!n_padded) return;
  hipLaunchKernelGGL(scene_to_planes_kernel, dim3((n_padded + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), in, n, planes, stride, off, n_padded, pad, counters);
}

}  // namespace mi355

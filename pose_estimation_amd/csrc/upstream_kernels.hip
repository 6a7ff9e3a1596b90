// upstream_kernels.hip — the reductions behind upstream's pre-normalisation (MI355_ICP_FLAG_UPSTREAM): what
// cv::ppf_match_3d::ICP::registerModelToScene computes before its level loop (reference call site
// pose_estimation/src/opencv_surface_match.cpp:94; upstream's source is absent, restated as recalled in
// oracle/icp_oracle.c, "UPSTREAM-FAITHFUL variant"):
//   computeMeanCols      -> cloud_sums_kernel: coordinate sums of a cloud (once per upload; the host divides)
//   computeDistToOrigin  -> centre_dists_kernel: per candidate, sum |T0 s - c| over the model and sum |q - c| over the scene
// Both are HBM-streaming sums of 12 B per point, microseconds at the clouds of a request, and both are DETERMINISTIC: a cloud
// is cut into kCloudSumBlocks fixed chunks, a chunk is summed by one workgroup in a fixed order, and the host adds the chunk
// sums in chunk order - the same cloud always gives the same bits (atomics would not).
// Float32 arithmetic follows the oracle operation for operation (-ffp-contract=off like every kernel file; sqrtf is the
// correctly rounded one, whose expansion uses fused multiply-adds - which is why this file is not part of `make check-isa`).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_internal.h"

namespace mi355 {

namespace {

constexpr int kSumVals = 3;

// fixed-order sum of NV doubles per thread over the workgroup; thread 0 returns the totals in out[]
template <int NV>
__device__ __forceinline__ void block_sum_fixed(double (&v)[NV], double* out) {
  __shared__ double wave_part[kBlockThreads / 64][NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    v[k] = x;
  }
  if ((threadIdx.x & 63u) == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) wave_part[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double t = wave_part[0][k];
#pragma unroll
      for (int w = 1; w < kBlockThreads / 64; ++w) t += wave_part[w][k];
      out[k] = t;
    }
  }
}

// chunk b of a cloud of n points cut into kCloudSumBlocks chunks: [lo, hi)
__device__ __forceinline__ void chunk_range(uint32_t n, uint32_t b, uint32_t& lo, uint32_t& hi) {
  const uint32_t per = (n + kCloudSumBlocks - 1) / kCloudSumBlocks;
  lo = b * per < n ? b * per : n;
  hi = lo + per < n ? lo + per : n;
}

}  // namespace

// grid = kCloudSumBlocks, block = 256: partials[b][3] = sum of x, y, z over chunk b
__global__ __launch_bounds__(kBlockThreads) void cloud_sums_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ z, uint32_t off, uint32_t n,
                                                                   double* __restrict__ partials) {
  uint32_t lo, hi;
  chunk_range(n, blockIdx.x, lo, hi);
  double acc[kSumVals] = {0.0, 0.0, 0.0};
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kBlockThreads) {
    acc[0] += static_cast<double>(x[off + i]);
    acc[1] += static_cast<double>(y[off + i]);
    acc[2] += static_cast<double>(z[off + i]);
  }
  block_sum_fixed<kSumVals>(acc, partials + static_cast<size_t>(blockIdx.x) * kSumVals);
}

// upstream's computeDistToOrigin on a centred point: sqrt(r0*r0 + r1*r1 + r2*r2) as a float expression
__device__ __forceinline__ double dist_to_centre(float px, float py, float pz, const float* c) {
  const float r0 = px - c[0], r1 = py - c[1], r2 = pz - c[2];
  float a = r0 * r0;
  const float b = r1 * r1, e = r2 * r2;
  a = a + b;
  a = a + e;
  return static_cast<double>(sqrtf(a));
}

// grid = (kCloudSumBlocks, jobs of the launch, 2), block = 256: partials[(first + job) * 2 + which][b]
__global__ __launch_bounds__(kBlockThreads) void centre_dists_kernel(NormArgs A, const float* __restrict__ sx, const float* __restrict__ sy,
                                                                     const float* __restrict__ sz, const float* __restrict__ tx,
                                                                     const float* __restrict__ ty, const float* __restrict__ tz,
                                                                     double* __restrict__ partials) {
  const NormJob& J = A.jobs[blockIdx.y];
  const bool scene = blockIdx.z != 0;
  const uint32_t n = scene ? J.tgt_n : J.src_n, off = scene ? J.tgt_off : J.src_off;
  uint32_t lo, hi;
  chunk_range(n, blockIdx.x, lo, hi);
  double acc[1] = {0.0};
  if (scene) {
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kBlockThreads) acc[0] += dist_to_centre(tx[off + i], ty[off + i], tz[off + i], J.centre);
  } else {
    const float* m = J.pose;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kBlockThreads) {
      const float X = sx[off + i], Y = sy[off + i], Z = sz[off + i];
      float a, b, px, py, pz;  // the contract's float32 transform (icp_kernels.hip: transform_point)
      a = m[0] * X; b = m[1] * Y; a = a + b; b = m[2] * Z; a = a + b; px = a + m[3];
      a = m[4] * X; b = m[5] * Y; a = a + b; b = m[6] * Z; a = a + b; py = a + m[7];
      a = m[8] * X; b = m[9] * Y; a = a + b; b = m[10] * Z; a = a + b; pz = a + m[11];
      acc[0] += dist_to_centre(px, py, pz, J.centre);
    }
  }
  block_sum_fixed<1>(acc, partials + (static_cast<size_t>(A.first + blockIdx.y) * 2 + (scene ? 1 : 0)) * kCloudSumBlocks + blockIdx.x);
}

void launch_cloud_sums(void* stream, DevicePlanes planes, uint32_t off, uint32_t n, double* partials) {
  hipLaunchKernelGGL(cloud_sums_kernel, dim3(kCloudSumBlocks), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), planes.x, planes.y,
                     planes.z, off, n, partials);
}

void launch_centre_dists(void* stream, const NormArgs& args, DevicePlanes src, DevicePlanes tgt, double* partials) {
  if (!args.n) return;
  hipLaunchKernelGGL(centre_dists_kernel, dim3(kCloudSumBlocks, args.n, 2), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), args,
                     src.x, src.y, src.z, tgt.x, tgt.y, tgt.z, partials);
}

}  // namespace mi355

// Does the f32-input MFMA share the vector ALU, or can the two pipes overlap?  Three loops, same trip count:
//   A: 4 independent v_mfma_f32_32x32x2_f32 chains           (matrix pipe only)
//   B: 32 independent v_fma_f32 per MFMA of A                 (vector ALU only, ~same duration as A when run alone)
//   C: A and B interleaved in one wave                        (separate pipes: ~max(A, B); shared hardware: ~A + B)
// also D: v_mfma_f32_32x32x16_bf16 alone and E: bf16 MFMA + the same VALU work, for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  bf16x8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {8, 7, 6, 5, 4, 3, 2, 1};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
    if (MODE == 3 || MODE == 4) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc3, 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2 || MODE == 4) {
      const int reps = (MODE == 4) ? 4 : 8;  // 4 MFMA f32 = 256 cycles ~ 128 VALU ops; 4 bf16 MFMA = 128 cycles ~ 64 VALU ops
#pragma unroll
      for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b, a);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int grid, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, iters / 4);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  const int iters = 200000;
  for (int wps : {1, 2}) {
    const int grid = 256 * wps;
    const float a = run<0>(out, grid, iters), b = run<1>(out, grid, iters), c = run<2>(out, grid, iters);
    const float d = run<3>(out, grid, iters), e = run<4>(out, grid, iters);
    printf("waves/SIMD=%d  f32-MFMA alone %.1f ms | VALU alone %.1f ms | f32-MFMA + VALU %.1f ms  (sum %.1f, max %.1f)\n", wps, a, b, c, a + b,
           a > b ? a : b);
    printf("              bf16-MFMA alone %.1f ms | bf16-MFMA + half the VALU work %.1f ms\n", d, e);
  }
  return 0;
}

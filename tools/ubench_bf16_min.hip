// ubench_bf16_min.hip — what one step of nn_scan_xbf16_kernel costs: a v_mfma_f32_32x32x16_bf16 (fresh accumulator)
// plus the unsigned minimum over the 16 results of an EARLIER MFMA, at 1 and 2 waves per SIMD.
//   M: MFMA only   V2: 16 v_min_u32 only   V3: the v_min_u32 / v_min3_u32 mix hipcc makes of the same tree   B2/B3: MFMA + minima alternating
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t m3(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t t = a < b ? a : b;
  return t < c ? t : c;
}
// kPlain: 16 two-operand minima as hipcc mixes them (v_min_u32 / v_min3_u32); otherwise the 8-deep v_min3_u32 tree
template <bool kPlain>
__device__ __forceinline__ uint32_t min16(const f32x16& acc, uint32_t seed) {
  uint32_t b[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) b[k] = __float_as_uint(acc[k]);
  if (!kPlain) {
    const uint32_t t0 = m3(b[0], b[1], b[2]), t1 = m3(b[3], b[4], b[5]), t2 = m3(b[6], b[7], b[8]), t3 = m3(b[9], b[10], b[11]);
    const uint32_t t4 = m3(b[12], b[13], b[14]), t5 = m3(b[15], t0, t1), t6 = m3(t2, t3, t4);
    return m3(t5, t6, seed);
  }
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) b[k] = b[k] < b[k + w] ? b[k] : b[k + w];
  return b[0] < seed ? b[0] : seed;
}

// 14 registers + seed through 7 v_min3_u32; the other two go to a per-lane LDS slot with ds_min_u32 (LDS pipe, no VALU slot)
__device__ __forceinline__ uint32_t min14_lds2(const f32x16& acc, uint32_t seed, uint32_t* slot) {
  uint32_t b[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) b[k] = __float_as_uint(acc[k]);
  __hip_atomic_fetch_min(slot, b[14], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_min(slot, b[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t t0 = m3(b[0], b[1], b[2]), t1 = m3(b[3], b[4], b[5]), t2 = m3(b[6], b[7], b[8]), t3 = m3(b[9], b[10], b[11]);
  const uint32_t t4 = m3(b[12], b[13], seed), t5 = m3(t0, t1, t2);
  return m3(t3, t4, t5);
}

// float minima written as asm, so hipcc adds no canonicalising v_max (IEEE mode) in front of them
__device__ __forceinline__ float fmin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fmin2(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fminimum3(float a, float b, float c) {
  float r;
  asm("v_minimum3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int KIND>  // 0: 8 v_min3_f32, 1: 16 v_min_f32, 2: 8 v_minimum3_f32 (gfx950)
__device__ __forceinline__ uint32_t fmin16(const f32x16& a, uint32_t seed_bits) {
  const float seed = __uint_as_float(seed_bits);
  if (KIND == 1) {
    float r = seed;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {  // four independent chains keep the adds apart
      r = fmin2(r, a[k]);
    }
    float r1 = a[1], r2 = a[2], r3 = a[3];
#pragma unroll
    for (int k = 4; k < 16; k += 4) {
      r1 = fmin2(r1, a[k + 1]);
      r2 = fmin2(r2, a[k + 2]);
      r3 = fmin2(r3, a[k + 3]);
    }
    return __float_as_uint(fmin2(fmin2(r, r1), fmin2(r2, r3)));
  }
  auto f = [](float x, float y, float z) { return KIND == 0 ? fmin3(x, y, z) : fminimum3(x, y, z); };
  const float t0 = f(a[0], a[1], a[2]), t1 = f(a[3], a[4], a[5]), t2 = f(a[6], a[7], a[8]), t3 = f(a[9], a[10], a[11]);
  const float t4 = f(a[12], a[13], a[14]), t5 = f(a[15], t0, t1), t6 = f(t2, t3, t4);
  return __float_as_uint(f(t5, t6, seed));
}

template <int MODE>  // 0 M, 1 V2, 2 V3, 3 B2, 4 B3, 5 B3 with two values per MFMA through ds_min_u32; 6/7/8: MFMA + float minima (fmin16<0/1/2>); 9/10/11: those minima alone
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const u32x4* in, long long* cyc = nullptr) {
  const long long c0 = clock64();  // s_memtime: ticks at the shader clock (profiles/r01a_clock_probe.txt)
  f32x16 acc[4];
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = threadIdx.x + i + m;
  u32x4 fa = in[threadIdx.x], fb = in[threadIdx.x + 256];
  uint32_t cm[4] = {0x7F800000u, 0x7F800000u, 0x7F800000u, 0x7F800000u};
  __shared__ uint32_t slots[4][256];
  for (int m = 0; m < 4; ++m) slots[m][threadIdx.x] = 0x7F800000u;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 8) asm volatile("" : "+v"(fa), "+v"(acc[(m + 2) & 3]), "+v"(cm[(m + 3) & 3]));
      if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 8)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), zero, 0, 0, 0);
      if (MODE == 1 || MODE == 3) cm[m] = min16<true>(acc[(m + 2) & 3], cm[m]);
      if (MODE == 2 || MODE == 4) cm[m] = min16<false>(acc[(m + 2) & 3], cm[m]);
      if (MODE == 5) cm[m] = min14_lds2(acc[(m + 2) & 3], cm[m], &slots[m][threadIdx.x]);
      if (MODE == 6 || MODE == 9) cm[m] = fmin16<0>(acc[(m + 2) & 3], cm[m]);
      if (MODE == 7 || MODE == 10) cm[m] = fmin16<1>(acc[(m + 2) & 3], cm[m]);
      if (MODE == 8 || MODE == 11) cm[m] = fmin16<2>(acc[(m + 2) & 3], cm[m]);
      if (MODE == 1 || MODE == 2 || MODE >= 9) asm volatile("" : "+v"(acc[(m + 2) & 3]));  // keep the minima in the loop
    }
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][7] + __uint_as_float(cm[m]) + (MODE == 5 ? __uint_as_float(slots[m][threadIdx.x]) : 0.f);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = clock64() - c0;
}

// the same loop in SHADER cycles (clock64 inside the kernel): what the wall-clock figures above are at the clock the chip
// really holds under that load - the number the scan kernel's GRBM_GUI_ACTIVE / MFMA count compares with
template <int MODE>
double run_cycles(float* out, const u32x4* in, int grid, int iters, long long* cyc_dev, float* ms_out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, iters / 4, in, cyc_dev);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, in, cyc_dev);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(ms_out, e0, e1);
  static long long h[4096];
  (void)hipMemcpy(h, cyc_dev, sizeof(long long) * grid, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int b = 0; b < grid; ++b) sum += static_cast<double>(h[b]);
  return sum / grid;  // cycles one workgroup (= one wave per SIMD) spent on iters x 4 steps
}

template <int MODE>
float run(float* out, const u32x4* in, int grid, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, iters / 4, in);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, in);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  u32x4* in;
  if (hipMalloc(&out, sizeof(float) * 256 * 8 * 256) != hipSuccess || hipMalloc(&in, 16 * 512) != hipSuccess) return 1;
  (void)hipMemset(in, 0x3c, 16 * 512);
  const int iters = 100000;  // x 4 steps
  for (int wps : {1, 2, 3}) {
    const int grid = 256 * wps;
    const float m = run<0>(out, in, grid, iters), v2 = run<1>(out, in, grid, iters), v3 = run<2>(out, in, grid, iters);
    const float b2 = run<3>(out, in, grid, iters), b3 = run<4>(out, in, grid, iters), b5 = run<5>(out, in, grid, iters);
    const double steps = 4.0 * iters * wps;  // steps per SIMD
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / steps; };
    printf("waves/SIMD=%d  cycles per step per SIMD at 2.4 GHz:  (MFMA only %.1f) | pairwise tree (hipcc: min/min3 mix) %.1f | 8 v_min3_u32 %.1f | MFMA + mix %.1f | MFMA + 8 v_min3_u32 %.1f\n",
           wps, cyc(m), cyc(v2), cyc(v3), cyc(b2), cyc(b3));
    printf("              MFMA + 7 v_min3_u32 + 2 ds_min_u32 (two of the 16 results through the LDS pipe): %.1f\n", cyc(b5));
    const float f6 = run<6>(out, in, grid, iters), f7 = run<7>(out, in, grid, iters), f8 = run<8>(out, in, grid, iters);
    const float f9 = run<9>(out, in, grid, iters), f10 = run<10>(out, in, grid, iters), f11 = run<11>(out, in, grid, iters);
    printf("              float minima (asm, no canonicalise): MFMA + 8 v_min3_f32 %.1f | MFMA + 16 v_min_f32 %.1f | MFMA + 8 v_minimum3_f32 %.1f || alone: %.1f | %.1f | %.1f\n",
           cyc(f6), cyc(f7), cyc(f8), cyc(f9), cyc(f10), cyc(f11));
  }
  long long* cyc_dev;
  if (hipMalloc(&cyc_dev, sizeof(long long) * 4096) != hipSuccess) return 1;
  for (int wps : {1, 2, 3, 4}) {
    const int grid = 256 * wps;
    float ms_v = 0, ms_b = 0;
    const double cv = run_cycles<2>(out, in, grid, iters, cyc_dev, &ms_v), cb = run_cycles<4>(out, in, grid, iters, cyc_dev, &ms_b);
    // a workgroup's wave shares its SIMD with wps - 1 others: steps per SIMD in that time = wps x its own
    const double steps_wave = 4.0 * iters;
    printf("waves/SIMD=%d  SHADER cycles per step per SIMD:  8 v_min3_u32 %.1f (clock %.2f GHz) | MFMA + 8 v_min3_u32 %.1f (clock %.2f GHz)\n", wps,
           cv / steps_wave / wps, cv / (ms_v * 1e-3) * 1e-9, cb / steps_wave / wps, cb / (ms_b * 1e-3) * 1e-9);
  }
  return 0;
}

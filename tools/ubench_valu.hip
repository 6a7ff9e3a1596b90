// ubench_valu.hip — measures FP32 VALU issue rates on gfx950 to price the NN scan's inner loop
// (design input for csrc/icp_kernels.hip; results are recorded in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o build/ubench_valu && ./build/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(S) S S S S S S S S S S S S S S S S

enum Kind { ADD, MUL, FMA, MIN3, PK_ADD, PK_MUL, PK_FMA, MIX_SCALAR, MIX_PK, NKIND };
static const char* kNames[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_min3_f32", "v_pk_add_f32", "v_pk_mul_f32",
                               "v_pk_fma_f32", "dist-mix scalar (3sub 3mul 2add 0.5min3 per pair)",
                               "dist-mix packed (pk sub/mul/add + min3)"};
// lane-ops per instruction (packed = 2)
static const int kLaneOps[] = {1, 1, 1, 1, 2, 2, 2, 1, 1};

template <int K>
__global__ __launch_bounds__(256) void bench(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f + blockIdx.x * 1e-9f, c = 0.9999f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f2 pb = {b, c};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if constexpr (K == ADD) {
      asm volatile(REP16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (K == MUL) {
      asm volatile(REP16("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (K == FMA) {
      asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (K == MIN3) {
      asm volatile(REP16("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                         "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (K == PK_ADD) {
      asm volatile(REP16("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
    } else if constexpr (K == PK_MUL) {
      asm volatile(REP16("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
    } else if constexpr (K == PK_FMA) {
      asm volatile(REP16("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
    } else if constexpr (K == MIX_SCALAR) {
      // two (source,target) pairs per group: 6 sub, 6 mul, 4 add, 1 min3 = 17 instructions
      asm volatile(REP16("v_sub_f32 %4, %0, %8\n v_sub_f32 %5, %1, %8\n v_sub_f32 %6, %2, %8\n"
                         "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n"
                         "v_add_f32 %4, %4, %5\n v_add_f32 %4, %4, %6\n"
                         "v_sub_f32 %5, %0, %9\n v_sub_f32 %6, %1, %9\n v_sub_f32 %7, %2, %9\n"
                         "v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                         "v_add_f32 %5, %5, %6\n v_add_f32 %5, %5, %7\n"
                         "v_min3_f32 %3, %3, %4, %5\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (K == MIX_PK) {
      // the same two pairs with packed ops: 3 pk_add(sub) 3 pk_mul 2 pk_add 1 min3 = 9 instructions
      asm volatile(REP16("v_pk_add_f32 %4, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %6, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n"
                         "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %4, %4, %6\n"
                         "v_min3_f32 %9, %9, %10, %11\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(pb), "+v"(a3) : "v"(a4), "v"(a5));
    }
  }
  long long t1 = clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x +
            p5.y + p6.x + p6.y + p7.x + p7.y + pb.x + pb.y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int K>
void run(int blocks_per_cu, int n_cu, float* out, long long* cyc, int iters) {
  const int instr_per_iter = (K == MIX_SCALAR) ? 17 * 16 : (K == MIX_PK) ? 9 * 16 : 8 * 16;
  const int grid = n_cu * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  bench<K><<<grid, 256>>>(out, cyc, iters / 8);  // warm-up
  hipEventRecord(e0);
  bench<K><<<grid, 256>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += double(v);
  mean /= h.size();
  const double wave_instr = double(grid) * 4 * double(iters) * instr_per_iter;
  const double per_simd_per_s = wave_instr / (n_cu * 4.0) / (ms * 1e-3);
  // clock64 ticks at the shader clock (s_memtime); in-kernel cycles per wave-instruction
  printf("%-52s waves/SIMD=%d  %8.3f ms  %7.2f G wave-instr/s/SIMD  in-wave cyc/instr=%6.2f  lane-ops/clk/CU(@2.4GHz)=%7.1f",
         kNames[K], blocks_per_cu, ms, per_simd_per_s * 1e-9, mean / (double(iters) * instr_per_iter),
         per_simd_per_s * 64.0 * kLaneOps[K] * 4 / 2.4e9);
  if (K == MIX_SCALAR || K == MIX_PK) printf("  => %.3e pairs/s chip", wave_instr / instr_per_iter * 16 * 2 * 64 / (ms * 1e-3) / 16);
  printf("\n");
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

// Sustained-load clock: runs the scalar distance mix for ~0.3 s and reads s_memtime (shader clock)
// against s_memrealtime (100 MHz) inside the kernel (MI355X_MICROARCH.md "DVFS give-back" item 6).
__global__ __launch_bounds__(256) void clock_probe(float* out, long long* ticks, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f + blockIdx.x * 1e-9f, c = 0.9999f;
  long long t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_sub_f32 %4, %0, %8\n v_sub_f32 %5, %1, %8\n v_sub_f32 %6, %2, %8\n"
                       "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n"
                       "v_add_f32 %4, %4, %5\n v_add_f32 %4, %4, %6\n"
                       "v_sub_f32 %5, %0, %9\n v_sub_f32 %6, %1, %9\n v_sub_f32 %7, %2, %9\n"
                       "v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                       "v_add_f32 %5, %5, %6\n v_add_f32 %5, %5, %7\n"
                       "v_min_f32 %3, %3, %4\n v_min_f32 %3, %3, %5\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
  }
  long long t1 = clock64(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same loop with the fused surrogate d2 = fma(dz,dz, fma(dy,dy, dx*dx)): 3 sub, 1 mul, 2 fma, 1 min per pair.
__global__ __launch_bounds__(256) void clock_probe_fma(float* out, long long* ticks, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f + blockIdx.x * 1e-9f, c = 0.9999f;
  long long t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_sub_f32 %4, %0, %8\n v_sub_f32 %5, %1, %8\n v_sub_f32 %6, %2, %8\n"
                       "v_mul_f32 %4, %4, %4\n v_fma_f32 %4, %5, %5, %4\n v_fma_f32 %4, %6, %6, %4\n"
                       "v_sub_f32 %5, %0, %9\n v_sub_f32 %6, %1, %9\n v_sub_f32 %7, %2, %9\n"
                       "v_mul_f32 %5, %5, %5\n v_fma_f32 %5, %6, %6, %5\n v_fma_f32 %5, %7, %7, %5\n"
                       "v_min_f32 %3, %3, %4\n v_min_f32 %3, %3, %5\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
  }
  long long t1 = clock64(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

static void run_clock_probe_fma(int n_cu, float* out, long long* cyc) {
  const int grid = n_cu * 4, iters = 400000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    clock_probe_fma<<<grid, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { sc += double(h[2 * i]); rt += double(h[2 * i + 1]); }
    const double ghz = sc / rt * 0.1;
    const double pairs = double(grid) * 256.0 * iters * 16.0 * 2.0;
    printf("fused-surrogate probe (3 sub, 1 mul, 2 fma, 1 min per pair): %.1f ms, clock %.3f GHz, %.3e pairs/s, slots/pair = %.2f\n",
           ms, ghz, pairs / (ms * 1e-3), ghz * 1e9 / 2.0 * n_cu * 4 * 64 / (pairs / (ms * 1e-3)));
  }
}

static void run_clock_probe(int n_cu, float* out, long long* cyc) {
  const int grid = n_cu * 4, iters = 400000;  // 4 waves/SIMD, 18 instr x 16 x iters per wave
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    clock_probe<<<grid, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { sc += double(h[2 * i]); rt += double(h[2 * i + 1]); }
    const double ghz = sc / rt * 0.1;  // realtime counter = 100 MHz
    const double pairs = double(grid) * 256.0 * iters * 16.0 * 2.0;
    printf("clock probe (v_min_f32 variant, 4 waves/SIMD): %.1f ms, shader clock %.3f GHz under load, %.3e pairs/s, "
           "slots/pair at that clock = %.2f\n", ms, ghz, pairs / (ms * 1e-3),
           ghz * 1e9 / 2.0 * n_cu * 4 * 64 / (pairs / (ms * 1e-3)));
  }
}

// Same probe as a train of short kernels (~1.5 ms each, host sync in between) - the shape of a 100k<->100k ICP loop.
static void run_short_probe(int n_cu, float* out, long long* cyc) {
  const int grid = n_cu * 4, iters = 1500;
  for (int rep = 0; rep < 60; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    clock_probe<<<grid, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep % 10 == 9) {
      std::vector<long long> h(grid * 2);
      hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      double sc = 0, rt = 0, rtmax = 0;
      for (int i = 0; i < grid; ++i) { sc += double(h[2 * i]); rt += double(h[2 * i + 1]); if (h[2*i+1] > rtmax) rtmax = double(h[2*i+1]); }
      const double pairs = double(grid) * 256.0 * iters * 16.0 * 2.0;
      printf("short probe launch %2d: %.3f ms event time, in-kernel span mean %.3f ms max %.3f ms, shader clock %.3f GHz, %.3e pairs/s\n",
             rep, ms, rt / grid * 1e-5, rtmax * 1e-5, sc / rt * 0.1, pairs / (ms * 1e-3));
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("device %s %s CUs=%d clock=%d kHz\n", prop.name, prop.gcnArchName, n_cu, prop.clockRate);
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * n_cu * 8 * 256);
  hipMalloc(&cyc, sizeof(long long) * n_cu * 8 * 4);
  const int iters = 20000;
  if (!getenv("UBENCH_SKIP_SHORT")) run_short_probe(n_cu, out, cyc);
  run_clock_probe(n_cu, out, cyc);
  run_clock_probe_fma(n_cu, out, cyc);
  if (getenv("UBENCH_CLOCK_ONLY")) return 0;
  for (int w : {1, 2, 4, 8}) {
    run<ADD>(w, n_cu, out, cyc, iters);
    run<MUL>(w, n_cu, out, cyc, iters);
    run<FMA>(w, n_cu, out, cyc, iters);
    run<MIN3>(w, n_cu, out, cyc, iters);
    run<PK_ADD>(w, n_cu, out, cyc, iters);
    run<PK_MUL>(w, n_cu, out, cyc, iters);
    run<PK_FMA>(w, n_cu, out, cyc, iters);
    run<MIX_SCALAR>(w, n_cu, out, cyc, iters);
    run<MIX_PK>(w, n_cu, out, cyc, iters);
  }
  return 0;
}

// probe_mfma_bf16.hip — how accurately does v_mfma_f32_32x32x16_bf16 accumulate its 16 products?
// The split-bf16 matrix-core filter (nn_scan_xbf16_kernel) budgets 4 u (u = 2^-24) of relative error per accumulation
// step against sum |terms|; this measures the real figure on random operands with heavy cancellation, including the
// magnitude pattern the filter produces (a few large terms that cancel, small corrections, one bias term).
//   hipcc --offload-arch=gfx950 -O2 -o build/probe_mfma_bf16 tools/probe_mfma_bf16.hip && ./build/probe_mfma_bf16
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const uint4* a, const uint4* b, float* d) {
  const uint32_t w = blockIdx.x, l = threadIdx.x;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[w * 64 + l]), __builtin_bit_cast(bf16x8, b[w * 64 + l]), zero, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[(w * 64 + l) * 16 + r] = acc[r];
}

static uint16_t bf16_of(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  return static_cast<uint16_t>(u);
}
static float val(uint16_t h) {
  uint32_t u = static_cast<uint32_t>(h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int waves = 2048;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::uniform_int_distribution<int> EX(-12, 0);
  // A[w][row 0..31][k 0..15], B[w][k][col]
  std::vector<uint16_t> A(static_cast<size_t>(waves) * 32 * 16), B(A.size());
  for (int w = 0; w < waves; ++w) {
    const int mode = w % 3;  // 0: uniform magnitudes; 1: wide exponent spread; 2: the filter's pattern
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < 16; ++k) {
        float a = U(rng), b = U(rng);
        if (mode == 1) { a = std::ldexp(a, EX(rng)); b = std::ldexp(b, EX(rng)); }
        if (mode == 2) {  // hi/lo pieces: slots 4c..4c+3 = (th, th, tl, tl) x (qh, ql, qh, ql); 12-14 = tt pieces x 1; 15 = 1 x bias
          const int c = k / 4, s = k % 4;
          if (c < 3) { a = (s < 2 ? 1.f : 1.f / 256) * a; b = ((s & 1) ? 1.f / 256 : 1.f) * b; }
          else if (k < 15) { a = std::fabs(a) * std::ldexp(1.f, -8 * (k - 12)); b = 1.f; }
          else { a = 1.f; b = std::fabs(b) * 3.f; }
        }
        A[(static_cast<size_t>(w) * 32 + i) * 16 + k] = bf16_of(a);
        B[(static_cast<size_t>(w) * 32 + i) * 16 + k] = bf16_of(b);  // B[w][col i][k]
      }
  }
  // fragments: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r]
  std::vector<uint4> fa(static_cast<size_t>(waves) * 64), fb(fa.size());
  for (int w = 0; w < waves; ++w)
    for (int l = 0; l < 64; ++l) {
      const int r = l & 31, h = l >> 5;
      uint32_t pa[4], pb[4];
      for (int j = 0; j < 4; ++j) {
        const size_t base = (static_cast<size_t>(w) * 32 + r) * 16 + 8 * h + 2 * j;
        pa[j] = A[base] | (static_cast<uint32_t>(A[base + 1]) << 16);
        pb[j] = B[base] | (static_cast<uint32_t>(B[base + 1]) << 16);
      }
      fa[static_cast<size_t>(w) * 64 + l] = make_uint4(pa[0], pa[1], pa[2], pa[3]);
      fb[static_cast<size_t>(w) * 64 + l] = make_uint4(pb[0], pb[1], pb[2], pb[3]);
    }
  uint4 *da, *db;
  float* dd;
  if (hipMalloc(&da, fa.size() * 16) != hipSuccess || hipMalloc(&db, fb.size() * 16) != hipSuccess ||
      hipMalloc(&dd, fa.size() * 16 * 4) != hipSuccess) { std::puts("hipMalloc failed"); return 1; }
  hipMemcpy(da, fa.data(), fa.size() * 16, hipMemcpyHostToDevice);
  hipMemcpy(db, fb.data(), fb.size() * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, da, db, dd);
  std::vector<float> D(fa.size() * 16);
  if (hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::puts("kernel failed"); return 1; }
  const double u = std::ldexp(1.0, -24);
  double worst_abs[3] = {0, 0, 0}, worst_res[3] = {0, 0, 0};
  long exact_rn[3] = {0, 0, 0}, total[3] = {0, 0, 0};
  for (int w = 0; w < waves; ++w)
    for (int l = 0; l < 64; ++l)
      for (int reg = 0; reg < 16; ++reg) {
        const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        double sum = 0, mag = 0;
        for (int k = 0; k < 16; ++k) {
          const double p = static_cast<double>(val(A[(static_cast<size_t>(w) * 32 + row) * 16 + k])) * val(B[(static_cast<size_t>(w) * 32 + col) * 16 + k]);
          sum += p;
          mag += std::fabs(p);
        }
        const double got = D[(static_cast<size_t>(w) * 64 + l) * 16 + reg];
        const int m = w % 3;
        const double e_mag = std::fabs(got - sum) / (mag * u);                       // in u of sum |terms|
        const double e_res = sum != 0 ? std::fabs(got - sum) / (std::fabs(sum) * u) : 0;  // in u of the result
        worst_abs[m] = std::max(worst_abs[m], e_mag);
        worst_res[m] = std::max(worst_res[m], e_res);
        exact_rn[m] += static_cast<float>(sum) == static_cast<float>(got);
        total[m] += 1;
      }
  const char* names[3] = {"uniform magnitudes", "wide exponent spread", "filter pattern"};
  for (int m = 0; m < 3; ++m)
    std::printf("%-22s %8ld results: max |err| = %.3f u of sum|terms| (%.3f u of |result|); %.2f %% equal the correctly rounded sum\n",
                names[m], total[m], worst_abs[m], worst_res[m], 100.0 * exact_rn[m] / total[m]);
  return 0;
}

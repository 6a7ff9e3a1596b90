// icp_ref_kernels.hip — gfx950 kernels of the "reference mode" pass: what
// cv::ppf_match_3d::ICP(iterations, tolerance, rejectionScale, numLevels) adds around the exact NN scan —
// resolution pyramid (strided subsampling), median + MAD rejection, "picky" one-to-one dedupe and the
// linearised point-to-plane normal equations (reference call site: opencv_surface_match.cpp:85,94).
// The NN scan itself is nn_scan_kernel / resolve_accumulate_kernel of icp_kernels.hip, run on the level's
// compacted planes.  Arithmetic contract: oracle/icp_oracle.c, second half (bit-exact threshold, mask and
// winners; sums in double).  Compile with -ffp-contract=off like icp_kernels.hip.
//
//   subsample_kernel          level planes:  out[i] = in[i * step], padded - every missing level of a cloud in one launch
//   rejection_select_reg_kernel  per job, ONE workgroup, values in registers (jobs of up to 32k points): exact rank
//                             select (11-bit radix digits from the highest bit that differs, LDS histograms, the last
//                             <= 256 candidates ranked by counting) of d2, then of |d2 - med|  ->  threshold
//   rejection_select_kernel   the same with the values in LDS / global memory (3-pass 11/11/10-bit radix), up to 64k points
//   select_hist/pick_kernel   the same selection across many workgroups, for jobs of >= 64k points
//   picky_scatter_kernel      accepted pairs: atomicMin(owner[target], d2 bits << 32 | source index)
//   ref_accumulate_kernel     winners: 21 + 6 normal-equation sums, sum |p-q|^2, count (29 doubles)
//   ref_final_reduce_kernel   fixed-order reduction of the per-workgroup partials
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>

#include "icp_internal.h"

namespace mi355 {

#ifdef MI355_ICP_DIAG
// Diagnostic build only (make diag): phase stamps (100 MHz) of the reference-mode kernels of ONE workgroup per launch
// (tools/ref_kernel_phases.py).  Record = [kernel id, n_stamps, t0, t1, ...] in slots of 16 words.
__device__ unsigned long long* g_diag_ref = nullptr;
__device__ unsigned int g_diag_ref_n = 0;
struct RefDiag {
  unsigned long long t[14];
  int n = 0;
  bool on = false;
  __device__ void begin(bool who) { on = who && g_diag_ref != nullptr; n = 0; stamp(); }
  __device__ void stamp() { if (on && n < 14) t[n++] = wall_clock64(); }
  __device__ void end(unsigned long long id) {
    if (!on) return;
    stamp();
    const unsigned int slot = atomicAdd(&g_diag_ref_n, 1u);
    if (slot >= 4096u) return;
    g_diag_ref[slot * 16ull] = id;
    g_diag_ref[slot * 16ull + 1] = static_cast<unsigned long long>(n);
    for (int k = 0; k < n; ++k) g_diag_ref[slot * 16ull + 2 + k] = t[k];
  }
};
#define REF_DIAG_BEGIN(who) RefDiag rd; rd.begin(who)
#define REF_DIAG_STAMP() rd.stamp()
#define REF_DIAG_STAMP_WAIT() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); rd.stamp(); } while (0)  // loads landed
#define REF_DIAG_END(id) rd.end(id)
#define REF_DIAG_ARG , RefDiag& rd
#define REF_DIAG_PASS , rd
#else
#define REF_DIAG_BEGIN(who) ((void)0)
#define REF_DIAG_STAMP() ((void)0)
#define REF_DIAG_STAMP_WAIT() ((void)0)
#define REF_DIAG_END(id) ((void)0)
#define REF_DIAG_ARG
#define REF_DIAG_PASS
#endif

namespace {

__device__ __forceinline__ void transform_point(const float* __restrict__ m, float x, float y, float z, float& ox,
                                                float& oy, float& oz) {
  float a, b;
  a = m[0] * x; b = m[1] * y; a = a + b; b = m[2] * z; a = a + b; ox = a + m[3];
  a = m[4] * x; b = m[5] * y; a = a + b; b = m[6] * z; a = a + b; oy = a + m[7];
  a = m[8] * x; b = m[9] * y; a = a + b; b = m[10] * z; a = a + b; oz = a + m[11];
}


}  // namespace

// ---------------------------------------------------------------------------------------------
// Several pyramid levels of one cloud in ONE launch: level q keeps the level-0 rows in_off + i * step_q (any step >= 1: the
// round 1-3 form samples with 2^l, the upstream-faithful one with cvRound(n / cvRound(n / 2^l)) of the model size); thread i
// writes point i of every level that has one; the grid covers the longest padded level.
// normals != nullptr (clouds uploaded with normals): a level with `rows` also gets its rows (target_rows_kernel's records:
// the level's coordinates with the normal of the level-0 row each point came from).
__global__ __launch_bounds__(kBlockThreads) void subsample_kernel(const float* __restrict__ in, size_t in_stride, SubsampleArgs A, float pad,
                                                                   const float* __restrict__ normals) {
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
#pragma unroll 1
  for (int q = 0; q < A.n_levels; ++q) {
    const SubsampleLevel& L = A.lv[q];
    if (i >= L.padded) continue;
    float v[3];
    const size_t row = L.in_off + static_cast<size_t>(i) * L.step;
#pragma unroll
    for (int plane = 0; plane < 3; ++plane) {
      v[plane] = i < L.n ? in[plane * in_stride + row] : pad;
      L.out[static_cast<size_t>(plane) * L.out_stride + L.out_off + i] = v[plane];
    }
    if (normals && L.rows && i < L.n) {
      float4* rows = static_cast<float4*>(L.rows);
      rows[2 * static_cast<size_t>(L.out_off + i)] = make_float4(v[0], v[1], v[2], normals[row]);
      rows[2 * static_cast<size_t>(L.out_off + i) + 1] = make_float4(normals[in_stride + row], normals[2 * in_stride + row], 0.f, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Exact rank select of non-negative float32 values by their bit pattern (monotone for x >= 0).
// kSelThreads threads cooperate; `value_of(i)` yields element i.  Returns the element of 0-based rank k.
constexpr int kSelThreads = 1024;

// After a pass the digit's bin holds the only values that can still be the answer.  Once there are at most 64 of them the
// remaining passes (a histogram clear, a walk with LDS atomics, a 2048-bin scan and four barriers each: ~1.2 us + the walk) are
// replaced by ONE walk that collects the candidates and one wave that ranks them against each other by counting: the value whose
// rank interval [#smaller, #smaller-or-equal) holds k.  Typically that happens after the first pass on the coarse pyramid levels
// and after the second anywhere (22 of 32 bits fixed).
template <typename F>
__device__ uint32_t radix_select(uint32_t n, uint32_t k, uint32_t* hist /* 2048 */, uint32_t* bcast /* 3 */,
                                 uint32_t* wave_tot /* 16 */, uint32_t* cand /* 64 */, uint32_t* cand_n, F value_of REF_DIAG_ARG) {
  uint32_t prefix = 0, mask = 0;
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const uint32_t nb = 1u << widths[pass];
    for (uint32_t b = threadIdx.x; b < 2048; b += kSelThreads) hist[b] = 0;
    if (threadIdx.x == 0) *cand_n = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kSelThreads) {
      const uint32_t bits = __float_as_uint(value_of(i));
      if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    REF_DIAG_STAMP();  // histogram walk
    // Which digit holds rank k?  Parallel scan over the 2048 bins (2 per thread): wave64 shuffle scan, wave totals
    // through LDS.  (A serial walk by one lane costs ~100 us per pass: one dependent LDS read per bin.)
    {
      const uint32_t c0 = hist[2 * threadIdx.x], c1 = hist[2 * threadIdx.x + 1];
      const uint32_t mine = c0 + c1;
      uint32_t incl = mine;
      const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= static_cast<uint32_t>(off)) incl += up;
      }
      if (lane == 63) wave_tot[wave] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (uint32_t w = 0; w < wave; ++w) base += wave_tot[w];
      const uint32_t excl = base + incl - mine;
      if (mine && k >= excl && k < excl + mine) {  // exactly one thread
        const bool second = k >= excl + c0;
        bcast[0] = 2 * threadIdx.x + (second ? 1u : 0u);
        bcast[1] = k - excl - (second ? c0 : 0u);
        bcast[2] = second ? c1 : c0;  // values in the chosen bin
      }
    }
    __syncthreads();
    prefix |= bcast[0] << shift;
    mask |= (nb - 1) << shift;
    k = bcast[1];
    const uint32_t in_bin = bcast[2];
    __syncthreads();
    REF_DIAG_STAMP();  // bin scan
    if (pass < 2 && in_bin <= 64u) {  // wave-uniform: everyone read the same words
      for (uint32_t i = threadIdx.x; i < n; i += kSelThreads) {
        const uint32_t bits = __float_as_uint(value_of(i));
        if ((bits & mask) == prefix) cand[atomicAdd(cand_n, 1u)] = bits;  // exactly in_bin appends
      }
      __syncthreads();
      REF_DIAG_STAMP();  // candidates collected
      if (threadIdx.x < 64) {
        const uint32_t c = threadIdx.x < in_bin ? cand[threadIdx.x] : 0xFFFFFFFFu;
        uint32_t less = 0, leq = 0;
        for (uint32_t j = 0; j < in_bin; ++j) {
          const uint32_t cj = cand[j];  // one address for the whole wave: an LDS broadcast
          less += cj < c ? 1u : 0u;
          leq += cj <= c ? 1u : 0u;
        }
        if (threadIdx.x < in_bin && less <= k && k < leq) bcast[0] = c;  // equal values all qualify and write the same bits
      }
      __syncthreads();
      const uint32_t answer = bcast[0];
      __syncthreads();
      return answer;
    }
  }
  return prefix;
}

// grid = n_jobs, block = kSelThreads.  thr[job] = scale * (1.48257968f * median|d2 - med|) + med, float32 ops as the oracle.
// The six radix passes re-read the job's distances; the first `cache_n` of them are kept in LDS (dynamic, up to 128 KiB of
// the CU's 160) so that global memory is walked once instead of six times (20k points: 31 -> ~15 us).
// With `owner` given the workgroup goes on to offer the job's accepted pairs to the picky owner table (what
// picky_scatter_kernel does in a launch of its own): the distances are still in LDS, and a pass of the coarse pyramid
// levels is a chain of dependent launches whose number is what it costs.
__global__ __launch_bounds__(kSelThreads) void rejection_select_kernel(const JobDesc* __restrict__ jobs,
                                                                        const float* __restrict__ dist2, float scale,
                                                                        float* __restrict__ thr_out, uint32_t cache_n,
                                                                        const RefJobDesc* __restrict__ rjobs,
                                                                        const int32_t* __restrict__ index,
                                                                        unsigned long long* __restrict__ owner) {
  extern __shared__ float cache[];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t bcast[3];
  __shared__ uint32_t wave_tot[kSelThreads / 64];
  __shared__ uint32_t cand[64];
  __shared__ uint32_t cand_n;
  __shared__ float thr_shared;
  REF_DIAG_BEGIN(blockIdx.x == 0 && threadIdx.x == 0);
  const JobDesc& J = jobs[blockIdx.x];
  const uint32_t n = J.src_n;
  if (n == 0) {
    if (threadIdx.x == 0) thr_out[blockIdx.x] = __builtin_inff();
    return;
  }
  const float* __restrict__ d = dist2 + J.out_off;
  const uint32_t cached = n < cache_n ? n : cache_n;
  for (uint32_t i = threadIdx.x; i < cached; i += kSelThreads) cache[i] = d[i];
  __syncthreads();
  REF_DIAG_STAMP();  // 1: distances in LDS
  const uint32_t k = (n - 1) / 2;
  const float med = __uint_as_float(radix_select(n, k, hist, bcast, wave_tot, cand, &cand_n, [&](uint32_t i) { return i < cached ? cache[i] : d[i]; } REF_DIAG_PASS));
  REF_DIAG_STAMP();  // 2: median
  const float mad = __uint_as_float(radix_select(n, k, hist, bcast, wave_tot, cand, &cand_n, [&](uint32_t i) {
    const float v = i < cached ? cache[i] : d[i];
    return static_cast<float>(__builtin_fabs(static_cast<double>(v) - static_cast<double>(med)));
  } REF_DIAG_PASS));
  REF_DIAG_STAMP();  // 3: MAD
  if (threadIdx.x == 0) {
    const float s = 1.48257968f * mad;
    const float t = scale * s;
    thr_out[blockIdx.x] = t + med;
    thr_shared = t + med;
  }
  if (!owner) {
    REF_DIAG_END(1ull | (static_cast<unsigned long long>(n) << 8));
    return;
  }
  __syncthreads();
  const float t = thr_shared;
  const RefJobDesc& RJ = rjobs[blockIdx.x];
  const int32_t* __restrict__ idx = index + J.out_off;
  for (uint32_t i = threadIdx.x; i < n; i += kSelThreads) {
    const float v = i < cached ? cache[i] : d[i];
    if (v < t) atomicMin(&owner[RJ.owner_off + static_cast<uint32_t>(idx[i])], picky_key(RJ.owner_prefix, v, RJ.src_index_base + i));
  }
  REF_DIAG_END(2ull | (static_cast<unsigned long long>(n) << 8));
}

// ---------------------------------------------------------------------------------------------
// The same selection with the job's values in REGISTERS (jobs of up to kVpt x 1024 points: the coarse and middle pyramid
// levels and a request's model at full resolution).  Phase stamps of the kernel above at 20k values (round 3,
// tools/ref_kernel_phases.py): every walk over the LDS copy - a loop of dependent LDS reads - costs 2-4 us and a select
// makes three of them, its first histogram walk on a digit (sign, exponent, two mantissa bits) that puts half the values in
// two or three bins.  Here a walk is an unrolled run over kVpt registers; the first digit starts at the highest bit in which
// the job's smallest and largest value DIFFER (their common prefix decides nothing), which spreads 20k squared distances over
// hundreds of bins; up to 256 values left in the chosen bin are ranked against each other by counting, all 1024 threads
// sharing the comparisons.  The selected element is the same one - the value of rank k in unsigned order of the bits.
constexpr uint32_t kSelCand = 256;

struct SelectShared {
  uint32_t hist[2048];
  uint32_t bcast[2][3];
  uint32_t wave_lo[kSelThreads / 64], wave_hi[kSelThreads / 64], wave_tot[kSelThreads / 64];
  uint32_t cand[kSelCand];
  uint32_t cand_n;
  uint32_t answer;
};

// The value of rank k among cand[0, m), m <= kSelCand, by counting (a barrier lies between the writes of cand and this call):
// candidate c = tid / 4 against every fourth candidate from tid % 4 on; the four partial counts meet in neighbouring lanes.
__device__ uint32_t rank_candidates(SelectShared& S, uint32_t m, uint32_t k) {
  const uint32_t tid = threadIdx.x, c = tid >> 2, q = tid & 3u;
  const uint32_t mine_v = c < m ? S.cand[c] : 0xFFFFFFFFu;
  uint32_t less = 0, leq = 0;
  for (uint32_t j = q; j < m; j += 4) {
    const uint32_t cj = S.cand[j];
    less += cj < mine_v ? 1u : 0u;
    leq += cj <= mine_v ? 1u : 0u;
  }
  less += __shfl_xor(less, 1, 64);
  leq += __shfl_xor(leq, 1, 64);
  less += __shfl_xor(less, 2, 64);
  leq += __shfl_xor(leq, 2, 64);
  if (q == 0 && c < m && less <= k && k < leq) S.answer = mine_v;  // equal values all qualify and write the same bits
  __syncthreads();
  return S.answer;
}

// hist[] all zero and cand_n == 0 on entry and on return.  v[j] belongs to element threadIdx.x + j * kSelThreads (< n).
// [lo, hi] (the same words in every thread) encloses the values, and every value has lo's bits above the highest bit in which
// lo and hi differ: the exact smallest and largest value do, and so does (0, largest value).
template <int kVpt>
__device__ uint32_t register_select(const uint32_t (&v)[kVpt], uint32_t n, uint32_t k, uint32_t lo, uint32_t hi, SelectShared& S REF_DIAG_ARG) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (lo == hi) return lo;  // (block-uniform) one distinct value
  int undecided = 32 - __builtin_clz(lo ^ hi);                                   // bits [undecided - 1, 0] can differ
  uint32_t mask = undecided >= 32 ? 0u : ~((1u << undecided) - 1u), prefix = lo & mask;  // the bits above them: common to all
  int parity = 0;
#pragma unroll 1
  while (true) {
    const int width = undecided < 11 ? undecided : 11, shift = undecided - width;
    const uint32_t digits = (1u << width) - 1u;
#pragma unroll
    for (int j = 0; j < kVpt; ++j)
      if (tid + j * kSelThreads < n && (v[j] & mask) == prefix) atomicAdd(&S.hist[(v[j] >> shift) & digits], 1u);
    __syncthreads();
    REF_DIAG_STAMP();  // histogram
    // which digit holds rank k: scan over the 2048 bins, two per thread (cleared on the way for the next user)
    const uint32_t c0 = S.hist[2 * tid], c1 = S.hist[2 * tid + 1];
    S.hist[2 * tid] = 0;
    S.hist[2 * tid + 1] = 0;
    const uint32_t mine = c0 + c1;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off, 64);
      if (lane >= static_cast<uint32_t>(off)) incl += up;
    }
    if (lane == 63) S.wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / 64; ++w) base += static_cast<uint32_t>(w) < wave ? S.wave_tot[w] : 0u;
    const uint32_t excl = base + incl - mine;
    if (mine && k >= excl && k < excl + mine) {  // exactly one thread
      const bool second = k >= excl + c0;
      S.bcast[parity][0] = 2 * tid + (second ? 1u : 0u);
      S.bcast[parity][1] = k - excl - (second ? c0 : 0u);
      S.bcast[parity][2] = second ? c1 : c0;  // values in the chosen bin
    }
    __syncthreads();
    prefix |= S.bcast[parity][0] << shift;
    mask |= digits << shift;
    k = S.bcast[parity][1];
    const uint32_t in_bin = S.bcast[parity][2];
    parity ^= 1;  // the next pass writes the other triple: no barrier needed between this read and that write
    undecided = shift;
    REF_DIAG_STAMP();  // bin found
    if (undecided == 0) return prefix;  // (block-uniform, like every branch here: all threads read the same words)
    if (in_bin <= kSelCand) {
#pragma unroll
      for (int j = 0; j < kVpt; ++j)
        if (tid + j * kSelThreads < n && (v[j] & mask) == prefix) S.cand[atomicAdd(&S.cand_n, 1u)] = v[j];  // exactly in_bin appends
      __syncthreads();
      if (tid == 0) S.cand_n = 0;
      const uint32_t answer = rank_candidates(S, in_bin, k);
      REF_DIAG_STAMP();  // candidates ranked
      return answer;
    }
  }
}

// grid = n_jobs, block = kSelThreads; every job of the launch has at most kVpt * kSelThreads points.
template <int kVpt>
__global__ __launch_bounds__(kSelThreads) void rejection_select_reg_kernel(const JobDesc* __restrict__ jobs, const float* __restrict__ dist2,
                                                                            float scale, float* __restrict__ thr_out,
                                                                            const RefJobDesc* __restrict__ rjobs, const int32_t* __restrict__ index,
                                                                            unsigned long long* __restrict__ owner) {
  __shared__ SelectShared S;
  REF_DIAG_BEGIN(blockIdx.x == 0 && threadIdx.x == 0);
  const JobDesc& J = jobs[blockIdx.x];
  const uint32_t n = J.src_n, tid = threadIdx.x;
  if (n == 0) {
    if (tid == 0) thr_out[blockIdx.x] = __builtin_inff();
    return;
  }
  const float* __restrict__ d = dist2 + J.out_off;
  uint32_t v[kVpt];
#pragma unroll
  for (int j = 0; j < kVpt; ++j) {
    const uint32_t i = tid + j * kSelThreads;
    v[j] = i < n ? __float_as_uint(d[i]) : 0xFFFFFFFFu;
  }
  const uint32_t k = (n - 1) / 2;
  float med, mad;
  if (n <= kSelCand) {  // (block-uniform) the coarsest pyramid levels: every value is a candidate, no histogram at all
    if (tid < n) S.cand[tid] = v[0];
    __syncthreads();
    med = __uint_as_float(rank_candidates(S, n, k));
    if (tid < n) S.cand[tid] = __float_as_uint(static_cast<float>(__builtin_fabs(static_cast<double>(__uint_as_float(v[0])) - static_cast<double>(med))));
    __syncthreads();
    mad = __uint_as_float(rank_candidates(S, n, k));
  } else {
  S.hist[2 * tid] = 0;
  S.hist[2 * tid + 1] = 0;
  if (tid == 0) S.cand_n = 0;
  // smallest and largest distance of the job (the selections start at the highest bit in which they differ)
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (int j = 0; j < kVpt; ++j)
    if (tid + j * kSelThreads < n) {
      lo = v[j] < lo ? v[j] : lo;
      hi = v[j] > hi ? v[j] : hi;
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((tid & 63u) == 0) {
    S.wave_lo[tid >> 6] = lo;
    S.wave_hi[tid >> 6] = hi;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kSelThreads / 64; ++w) {
    const uint32_t a = S.wave_lo[w], b = S.wave_hi[w];
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  REF_DIAG_STAMP();  // 1: distances in registers, their range known
  med = __uint_as_float(register_select<kVpt>(v, n, k, lo, hi, S REF_DIAG_PASS));
  __syncthreads();  // the second selection reuses the words the first one's last readers may still be reading
  auto deviation = [&](uint32_t bits) {
    return __float_as_uint(static_cast<float>(__builtin_fabs(static_cast<double>(__uint_as_float(bits)) - static_cast<double>(med))));
  };
  uint32_t dev[kVpt];
#pragma unroll
  for (int j = 0; j < kVpt; ++j) dev[j] = deviation(v[j]);
  // the largest deviation is the smallest or the largest distance's (the rounded difference is monotone in |d2 - med|)
  const uint32_t dev_lo = deviation(lo), dev_hi = deviation(hi);
  mad = __uint_as_float(register_select<kVpt>(dev, n, k, 0u, dev_lo > dev_hi ? dev_lo : dev_hi, S REF_DIAG_PASS));
  }
  const float mad_s = 1.48257968f * mad;  // every thread: the same float32 operations on the same operands, as the kernel above
  const float mad_t = scale * mad_s;
  const float t = mad_t + med;
  if (tid == 0) thr_out[blockIdx.x] = t;
  if (!owner) {
    REF_DIAG_END(1ull | (static_cast<unsigned long long>(n) << 8));
    return;
  }
  const RefJobDesc& RJ = rjobs[blockIdx.x];
  const int32_t* __restrict__ idx = index + J.out_off;
#pragma unroll
  for (int j = 0; j < kVpt; ++j) {
    const uint32_t i = tid + j * kSelThreads;
    const float val = __uint_as_float(v[j]);
    if (i < n && val < t) atomicMin(&owner[RJ.owner_off + static_cast<uint32_t>(idx[i])], picky_key(RJ.owner_prefix, val, RJ.src_index_base + i));
  }
  REF_DIAG_END(2ull | (static_cast<unsigned long long>(n) << 8));
}

// ---------------------------------------------------------------------------------------------
// The same selection across MANY workgroups, for large jobs (one workgroup walks a million values six times in 1.4 ms).
// Per radix pass two launches: select_hist_kernel - every workgroup histograms a strided share of the job's values that
// still match the chosen prefix into LDS and adds it to the job's global histogram - and select_pick_kernel - one
// workgroup per job finds the digit that holds rank k, extends the prefix, clears the histogram; after pass 3 it has
// the median (stage 0) and the following three passes select the median of |d2 - med| (stage 1), after which it writes
// the threshold.  Counts are integers: the selected element, hence the threshold, is the one-workgroup kernel's bit for bit.
struct SelectState {
  uint32_t prefix, mask, k;
  float med;
};
// scratch: the jobs' 2048-bin histograms form one contiguous block (a slab-sharded pass adds it up across ranks between
// the two kernels), their SelectStates another
__device__ __forceinline__ uint32_t* select_hist_of(uint32_t* hists, uint32_t job) { return hists + static_cast<size_t>(job) * 2048u; }
__device__ __forceinline__ SelectState* select_state_of(uint32_t* states, uint32_t job) {
  return reinterpret_cast<SelectState*>(states + static_cast<size_t>(job) * 4u);
}

__device__ __forceinline__ float select_value(const float* __restrict__ d, uint32_t i, int stage, float med) {
  const float v = d[i];
  return stage == 0 ? v : static_cast<float>(__builtin_fabs(static_cast<double>(v) - static_cast<double>(med)));
}

// grid = (chunks, n_jobs), block = kSelThreads.  pass = 0..5 (3 per stage)
__global__ __launch_bounds__(kSelThreads) void select_hist_kernel(const JobDesc* __restrict__ jobs, const float* __restrict__ dist2,
                                                                   uint32_t* __restrict__ work, uint32_t* __restrict__ states, int pass) {
  __shared__ uint32_t hist[2048];
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t n = J.src_n;
  if (blockIdx.x * kSelThreads >= n) return;
  const int stage = pass / 3, sub = pass % 3;
  const int shift = sub == 0 ? 21 : (sub == 1 ? 10 : 0);
  const uint32_t nb = sub == 2 ? 1024u : 2048u;
  const SelectState st = *select_state_of(states, blockIdx.y);
  uint32_t* ghist = select_hist_of(work, blockIdx.y);
  const uint32_t prefix = sub == 0 ? 0u : st.prefix, mask = sub == 0 ? 0u : st.mask;
  for (uint32_t b = threadIdx.x; b < 2048; b += kSelThreads) hist[b] = 0;
  __syncthreads();
  const float* __restrict__ d = dist2 + J.out_off;
  for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n; i += gridDim.x * kSelThreads) {
    const uint32_t bits = __float_as_uint(select_value(d, i, stage, st.med));
    if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += kSelThreads)
    if (hist[b]) atomicAdd(&ghist[b], hist[b]);
}

// grid = n_jobs, block = kSelThreads
__global__ __launch_bounds__(kSelThreads) void select_pick_kernel(const JobDesc* __restrict__ jobs, uint32_t* __restrict__ work,
                                                                   uint32_t* __restrict__ states, int pass, float scale,
                                                                   float* __restrict__ thr_out) {
  __shared__ uint32_t bcast[2];
  __shared__ uint32_t wave_tot[kSelThreads / 64];
  const JobDesc& J = jobs[blockIdx.x];
  const uint32_t n = J.src_n_total ? J.src_n_total : J.src_n;  // slab-sharded: the rank is taken among ALL ranks' values
  if (n == 0) {
    if (threadIdx.x == 0 && pass == 5) thr_out[blockIdx.x] = __builtin_inff();
    return;
  }
  const int sub = pass % 3;
  const int shift = sub == 0 ? 21 : (sub == 1 ? 10 : 0);
  const uint32_t nb = sub == 2 ? 1024u : 2048u;
  SelectState st = *select_state_of(states, blockIdx.x);
  const uint32_t k = sub == 0 ? (n - 1) / 2 : st.k;
  uint32_t* h = select_hist_of(work, blockIdx.x);
  const uint32_t c0 = h[2 * threadIdx.x], c1 = h[2 * threadIdx.x + 1];
  const uint32_t mine = c0 + c1;
  uint32_t incl = mine;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; ++w) base += wave_tot[w];
  const uint32_t excl = base + incl - mine;
  if (mine && k >= excl && k < excl + mine) {  // exactly one thread
    const bool second = k >= excl + c0;
    bcast[0] = 2 * threadIdx.x + (second ? 1u : 0u);
    bcast[1] = k - excl - (second ? c0 : 0u);
  }
  h[2 * threadIdx.x] = 0;  // ready for the next pass
  h[2 * threadIdx.x + 1] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    st.prefix = (sub == 0 ? 0u : st.prefix) | (bcast[0] << shift);
    st.mask = (sub == 0 ? 0u : st.mask) | ((nb - 1) << shift);
    st.k = bcast[1];
    if (pass == 2) st.med = __uint_as_float(st.prefix);
    if (pass == 5) {
      const float mad = __uint_as_float(st.prefix);
      const float s = 1.48257968f * mad;
      const float t = scale * s;
      thr_out[blockIdx.x] = t + st.med;
    }
    *select_state_of(states, blockIdx.x) = st;
  }
}

// ---------------------------------------------------------------------------------------------
// grid = (max src_blocks, n_jobs).  owner[] was set to ~0 before the launch.
__global__ __launch_bounds__(kBlockThreads) void picky_scatter_kernel(const JobDesc* __restrict__ jobs,
                                                                       const RefJobDesc* __restrict__ rjobs,
                                                                       const int32_t* __restrict__ index,
                                                                       const float* __restrict__ dist2,
                                                                       const float* __restrict__ thr,
                                                                       unsigned long long* __restrict__ owner) {
  const JobDesc& J = jobs[blockIdx.y];
  if (blockIdx.x >= J.src_blocks) return;
  const float t = thr[blockIdx.y];
  const RefJobDesc& RJ = rjobs[blockIdx.y];
  const uint32_t owner_off = RJ.owner_off;
  // distances and indices of the lane's four points go out together (one round trip, not two per point one after the other)
  float d[kResolvePointsPerLane];
  uint32_t j[kResolvePointsPerLane];
#pragma unroll
  for (int p = 0; p < kResolvePointsPerLane; ++p) {
    const uint32_t local = blockIdx.x * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    const uint32_t at = J.out_off + (local < J.src_n ? local : 0u);
    d[p] = dist2[at];
    j[p] = static_cast<uint32_t>(index[at]);
  }
#pragma unroll
  for (int p = 0; p < kResolvePointsPerLane; ++p) {
    const uint32_t local = blockIdx.x * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    if (local < J.src_n && d[p] < t) atomicMin(&owner[owner_off + j[p]], picky_key(RJ.owner_prefix, d[p], RJ.src_index_base + local));
  }
}

// ---------------------------------------------------------------------------------------------
// Slab-sharded passes whose model is small against the scene: instead of min-reducing the WHOLE owner table across ranks
// (8 B x level targets per pass), every rank lists the pairs it OFFERS (<= its accepted slab points), the lists are
// all-gathered and every rank applies the others' offers to its own table.  atomicMin is order-free, so the table ends up the
// same as under the table-wide reduction whatever order the lists are written and applied in.
// block: [0] = number of records (as uint64), then records {key, owner slot}.  grid = (max src_blocks, n_jobs).
__global__ __launch_bounds__(kBlockThreads) void picky_offer_kernel(const JobDesc* __restrict__ jobs, const RefJobDesc* __restrict__ rjobs,
                                                                     const int32_t* __restrict__ index, const float* __restrict__ dist2,
                                                                     const float* __restrict__ thr, unsigned long long* __restrict__ owner,
                                                                     unsigned long long* __restrict__ block, uint32_t capacity) {
  const JobDesc& J = jobs[blockIdx.y];
  if (blockIdx.x >= J.src_blocks) return;
  const float t = thr[blockIdx.y];
  const RefJobDesc& RJ = rjobs[blockIdx.y];
#pragma unroll
  for (int p = 0; p < kResolvePointsPerLane; ++p) {
    const uint32_t local = blockIdx.x * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    if (local >= J.src_n) continue;
    const float d = dist2[J.out_off + local];
    if (!(d < t)) continue;
    const uint32_t slot = RJ.owner_off + static_cast<uint32_t>(index[J.out_off + local]);
    const unsigned long long key = picky_key(RJ.owner_prefix, d, RJ.src_index_base + local);
    atomicMin(&owner[slot], key);
    const unsigned long long at = atomicAdd(&block[0], 1ull);
    if (at < capacity) {  // (the capacity covers every point of the slab: never exceeded)
      block[1 + 2 * at] = key;
      block[2 + 2 * at] = slot;
    }
  }
}

// all: n_ranks blocks of `block_words` uint64 each (the gathered lists); every record of every OTHER rank is offered to this
// rank's table.  grid = (ceil(capacity / 256), n_ranks).
__global__ __launch_bounds__(kBlockThreads) void picky_apply_kernel(const unsigned long long* __restrict__ all, uint32_t block_words, int my_rank,
                                                                     uint32_t capacity, unsigned long long* __restrict__ owner) {
  if (static_cast<int>(blockIdx.y) == my_rank) return;
  const unsigned long long* __restrict__ block = all + static_cast<size_t>(blockIdx.y) * block_words;
  const unsigned long long n = block[0] < capacity ? block[0] : capacity;
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n) return;
  atomicMin(&owner[block[2 + 2 * static_cast<size_t>(i)]], block[1 + 2 * static_cast<size_t>(i)]);
}

void launch_picky_offer(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs, uint32_t max_src_blocks, const int32_t* index,
                        const float* dist2, const float* thr, unsigned long long* owner, unsigned long long* block, uint32_t capacity) {
  hipLaunchKernelGGL(picky_offer_kernel, dim3(max_src_blocks, static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream),
                     jobs_dev, rjobs_dev, index, dist2, thr, owner, block, capacity);
}
void launch_picky_apply(void* stream, const unsigned long long* all, uint32_t block_words, int n_ranks, int my_rank, uint32_t capacity,
                        unsigned long long* owner) {
  if (!capacity || n_ranks < 2) return;
  hipLaunchKernelGGL(picky_apply_kernel, dim3((capacity + kBlockThreads - 1) / kBlockThreads, static_cast<uint32_t>(n_ranks)), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), all, block_words, my_rank, capacity, owner);
}

// ---------------------------------------------------------------------------------------------
// Rows of the pass result: [n x 29 sums][n overflow-excess flags][n thresholds][n overflow counts], n = jobs of the pass.
__device__ __forceinline__ void write_pass_tail(double* sums, uint32_t n_jobs, uint32_t job, const float* thr,
                                                uint32_t* overflow_count, uint32_t overflow_cap) {
  double* tail = sums + static_cast<size_t>(n_jobs) * kRefSums;
  const uint32_t n = overflow_count ? overflow_count[job] : 0u;
  tail[job] = n > overflow_cap ? 1.0 : 0.0;
  tail[n_jobs + job] = static_cast<double>(thr[job]);  // float -> double is exact
  tail[2 * n_jobs + job] = static_cast<double>(n);
  if (overflow_count) overflow_count[job] = 0u;
}

// The target rows the accumulate gathers, one 32-byte record per point of a pyramid level: (x, y, z, nx | ny, nz, 0, 0) -
// coordinates of the level's compacted planes, normal of the level-0 row the point came from.  Built once per upload.
// The accumulate is bound by the ISSUE of its scattered loads (64 different lines per instruction; phase stamps, round 3:
// 6 of a launch's 11 us at 6 x 20k pairs), and a record is two such loads where six planes were six.
// grid = ceil(n / 256), block = 256.
__global__ __launch_bounds__(kBlockThreads) void target_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                    const float* __restrict__ z, uint32_t level_off, uint32_t n,
                                                                    const float* __restrict__ nx, const float* __restrict__ ny,
                                                                    const float* __restrict__ nz, uint32_t normal_off, uint32_t step,
                                                                    float4* __restrict__ rows) {
  const uint32_t j = blockIdx.x * kBlockThreads + threadIdx.x;
  if (j >= n) return;
  const size_t nj = static_cast<size_t>(normal_off) + static_cast<size_t>(j) * step;
  rows[2 * static_cast<size_t>(level_off + j)] = make_float4(x[level_off + j], y[level_off + j], z[level_off + j], nx[nj]);
  rows[2 * static_cast<size_t>(level_off + j) + 1] = make_float4(ny[nj], nz[nj], 0.f, 0.f);
}

// grid = (max src_blocks, n_jobs).  src are the LEVEL planes, rows the level's target records (target_rows_kernel).
// With tail.done set the launch also ENDS the pass (single-GPU reference mode): finish_pass, icp_internal.h.
// kUpstream (MI355_ICP_FLAG_UPSTREAM; oracle/icp_oracle.c, "UPSTREAM-FAITHFUL variant"): the model point comes with its normal
// from the level's SOURCE rows (srows: one 32-byte record, the same layout as the target's), the plane equations are written for
// p - centre, and sum 29 collects |n_p - n_q|^2 with n_p the model normal under the level-start rotation, re-normalised and
// rounded to float32 the way upstream's transformPCPose leaves it.  Otherwise sum 29 stays 0.
template <bool kUpstream>
__global__ __launch_bounds__(kBlockThreads) void ref_accumulate_kernel(
    const JobDesc* __restrict__ jobs, const RefJobDesc* __restrict__ rjobs, const float* __restrict__ sx,
    const float* __restrict__ sy, const float* __restrict__ sz, const float4* __restrict__ srows, const float4* __restrict__ rows,
    const int32_t* __restrict__ index,
    const float* __restrict__ dist2, const float* __restrict__ thr, const unsigned long long* __restrict__ owner,
    uint8_t* __restrict__ selected, double* partials, PassTail tail) {
  __shared__ double scratch[kBlockThreads / 64][kRefSums];
  __shared__ uint32_t last_flag;
  REF_DIAG_BEGIN(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
  const JobDesc& J = jobs[blockIdx.y];
  const uint32_t sb = blockIdx.x;
  if (sb >= J.src_blocks) return;
  // raised by earlier launches of the call only: read now, needed at the very end
  const uint32_t raised = tail.done ? __hip_atomic_load(tail.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const RefJobDesc& RJ = rjobs[blockIdx.y];
  float pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = RJ.pose_start[k];
  const float t = thr[blockIdx.y];
  double acc[kRefSums];
#pragma unroll
  for (int k = 0; k < kRefSums; ++k) acc[k] = 0.0;
  // A point is a chain of dependent gathers (distance and index -> owner slot -> source, target and normal rows).  The
  // lane's four points walk it side by side - every load of a stage is issued before the first one is used - so the
  // launch waits for three round trips, not twelve: on the coarse pyramid levels that IS the launch.
  constexpr int kP = kResolvePointsPerLane;
  uint32_t local[kP], jt[kP];
  float dd[kP];
  bool live[kP];
#pragma unroll
  for (int p = 0; p < kP; ++p) {
    local[p] = sb * kSrcPerBlock + p * kBlockThreads + threadIdx.x;
    live[p] = local[p] < J.src_n;
    const uint32_t at = J.out_off + (live[p] ? local[p] : 0u);
    dd[p] = dist2[at];
    jt[p] = static_cast<uint32_t>(index[at]);
  }
  REF_DIAG_STAMP_WAIT();  // 1: descriptors read, distances and indices arrived
  // the owner keys and the rows of the pairs that pass the threshold go out together (neither address depends on the other's
  // data): one round trip for both.  A pair the threshold rejects needs no rows - its lanes sit the gathers out.
  unsigned long long own[kP];
  bool near[kP];
  float S0[kP], S1[kP], S2[kP], Qx[kP], Qy[kP], Qz[kP], Nx[kP], Ny[kP], Nz[kP];
  float M0[kP], M1[kP], M2[kP];  // kUpstream: the model point's normal
#pragma unroll
  for (int p = 0; p < kP; ++p) {
    near[p] = live[p] && (dd[p] < t);
    own[p] = 0ull;
    S0[p] = S1[p] = S2[p] = Qx[p] = Qy[p] = Qz[p] = Nx[p] = Ny[p] = Nz[p] = 0.f;
    M0[p] = M1[p] = M2[p] = 0.f;
    if (near[p]) {
      own[p] = owner[RJ.owner_off + jt[p]];
      const uint32_t i = J.src_off + local[p];
      if constexpr (kUpstream) {
        const float4* srec = srows + 2 * static_cast<size_t>(i);
        const float4 sa = srec[0];
        const float2 sb2 = *reinterpret_cast<const float2*>(srec + 1);
        S0[p] = sa.x; S1[p] = sa.y; S2[p] = sa.z;
        M0[p] = sa.w; M1[p] = sb2.x; M2[p] = sb2.y;
      } else {
        S0[p] = sx[i]; S1[p] = sy[i]; S2[p] = sz[i];
      }
      const float4* rec = rows + 2 * static_cast<size_t>(J.tgt_off + jt[p]);
      const float4 a = rec[0];
      const float2 b = *reinterpret_cast<const float2*>(rec + 1);
      Qx[p] = a.x; Qy[p] = a.y; Qz[p] = a.z;
      Nx[p] = a.w; Ny[p] = b.x; Nz[p] = b.y;
    }
  }
  REF_DIAG_STAMP();  // 2: owner keys and rows requested
  bool sel[kP];
#pragma unroll
  for (int p = 0; p < kP; ++p) sel[p] = near[p] && own[p] == picky_key(RJ.owner_prefix, dd[p], RJ.src_index_base + local[p]);
  REF_DIAG_STAMP_WAIT();  // 3: source, target and normal rows arrived
#pragma unroll
  for (int p = 0; p < kP; ++p) {
    if (selected && live[p]) selected[J.out_off + local[p]] = sel[p] ? 1 : 0;  // the per-pass diagnostic only (mi355_icp_reference_pass)
    if (sel[p]) {
      float px, py, pz;
      transform_point(pose, S0[p], S1[p], S2[p], px, py, pz);  // the level-start point
      const double P0 = px, P1 = py, P2 = pz, Q0 = Qx[p], Q1 = Qy[p], Q2 = Qz[p];
      const double N0 = Nx[p], N1 = Ny[p], N2 = Nz[p];
      // the point the cross product is taken with: p itself, or p - centre (upstream linearises the rotation about its joint mean)
      const double C0 = kUpstream ? P0 - static_cast<double>(RJ.centre[0]) : P0;
      const double C1 = kUpstream ? P1 - static_cast<double>(RJ.centre[1]) : P1;
      const double C2 = kUpstream ? P2 - static_cast<double>(RJ.centre[2]) : P2;
      const double a[6] = {C1 * N2 - C2 * N1, C2 * N0 - C0 * N2, C0 * N1 - C1 * N0, N0, N1, N2};
      const double dx = Q0 - P0, dy = Q1 - P1, dz = Q2 - P2;
      const double b = dx * N0 + dy * N1 + dz * N2;
      int k = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) acc[k++] += a[r] * a[c];
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[21 + r] += a[r] * b;
      acc[27] += dx * dx + dy * dy + dz * dz;
      acc[28] += 1.0;
      if constexpr (kUpstream) {
        // n_p = (R n) / |R n|, R the float32 level-start pose, products and sums in double, rounded to float32; zero below FLT_EPSILON
        const double m0 = M0[p], m1 = M1[p], m2 = M2[p];
        const double r0 = (static_cast<double>(pose[0]) * m0 + static_cast<double>(pose[1]) * m1) + static_cast<double>(pose[2]) * m2;
        const double r1 = (static_cast<double>(pose[4]) * m0 + static_cast<double>(pose[5]) * m1) + static_cast<double>(pose[6]) * m2;
        const double r2 = (static_cast<double>(pose[8]) * m0 + static_cast<double>(pose[9]) * m1) + static_cast<double>(pose[10]) * m2;
        const double nn = __builtin_sqrt((r0 * r0 + r1 * r1) + r2 * r2);
        const bool ok = nn > 1.192092896e-07;
        const double e0 = (ok ? static_cast<double>(static_cast<float>(r0 / nn)) : 0.0) - N0;
        const double e1 = (ok ? static_cast<double>(static_cast<float>(r1 / nn)) : 0.0) - N1;
        const double e2 = (ok ? static_cast<double>(static_cast<float>(r2 / nn)) : 0.0) - N2;
        acc[29] += (e0 * e0 + e1 * e1) + e2 * e2;
      }
    }
  }
  REF_DIAG_STAMP();  // 4: products accumulated (gathers arrived)
  const double r = pass_block_sum<kRefSums>(acc, scratch);
  REF_DIAG_STAMP();  // 5: block sum
  REF_DIAG_END(3ull | (static_cast<unsigned long long>(J.src_n) << 8) | (static_cast<unsigned long long>(J.src_blocks) << 40));
  double* job_rows = partials + static_cast<size_t>(J.partial_off) * kRefSums;
  if (!tail.done) {
    if (threadIdx.x < kRefSums) job_rows[static_cast<size_t>(sb) * kRefSums + threadIdx.x] = r;
    return;
  }
  const float threshold = thr[blockIdx.y];
  finish_pass<kRefSums>(r, acc, scratch, &last_flag, job_rows, sb, J.src_blocks, tail, raised, [&](double* t, uint32_t job, uint32_t n_jobs) {
    publish_overflow(tail, job, t + job, t + 2 * n_jobs + job);
    __hip_atomic_store(t + n_jobs + job, static_cast<double>(threshold), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // float -> double is exact
  });
}

// Tail of `sums` behind the n = gridDim.x rows: [n flags][n thresholds][n counts] - see final_reduce_kernel; the rejection
// thresholds ride along so that the pass needs ONE device-to-host copy.
__global__ __launch_bounds__(kBlockThreads) void ref_final_reduce_kernel(const JobDesc* __restrict__ jobs,
                                                                          const double* __restrict__ partials,
                                                                          double* __restrict__ sums, const float* __restrict__ thr,
                                                                          uint32_t* __restrict__ overflow_count, uint32_t overflow_cap) {
  __shared__ double scratch[kBlockThreads / 64][kRefSums];
  const JobDesc& J = jobs[blockIdx.x];
  double acc[kRefSums];
#pragma unroll
  for (int k = 0; k < kRefSums; ++k) acc[k] = 0.0;
  for (uint32_t b = threadIdx.x; b < J.src_blocks; b += kBlockThreads) {
    const double* __restrict__ row = partials + static_cast<size_t>(J.partial_off + b) * kRefSums;
#pragma unroll
    for (int k = 0; k < kRefSums; ++k) acc[k] += row[k];
  }
  const double r = pass_block_sum<kRefSums>(acc, scratch);
  if (threadIdx.x < kRefSums) sums[static_cast<size_t>(blockIdx.x) * kRefSums + threadIdx.x] = r;
  if (threadIdx.x == 0) write_pass_tail(sums, gridDim.x, blockIdx.x, thr, overflow_count, overflow_cap);
}

// ---------------------------------------------------------------------------------------------
void launch_subsample_levels(void* stream, const float* in, size_t in_stride, const SubsampleArgs& args, float pad, const float* normals) {
  uint32_t threads = 0;
  for (int q = 0; q < args.n_levels; ++q) threads = std::max(threads, args.lv[q].padded);
  if (!threads) return;
  hipLaunchKernelGGL(subsample_kernel, dim3((threads + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), in, in_stride, args, pad, normals);
}

// max_n: largest job of the launch.  hists / states: scratch zeroed once at allocation (the pick kernel leaves the
// histograms clear); small launches keep the one-workgroup kernel (a dozen launches cost more).
// Returns true when the launch also offered the accepted pairs to `owner` (the one-workgroup kernel does when asked to:
// rjobs_dev, index and owner given); false = the caller still has to launch picky_scatter_kernel.
bool launch_rejection_select(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_n, const float* dist2, float scale,
                             float* thr, uint32_t* hists, uint32_t* states, const RefJobDesc* rjobs_dev, const int32_t* index,
                             unsigned long long* owner) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (max_n <= 32u * kSelThreads) {  // the jobs' values fit the registers of one workgroup each
    // one workgroup offering 20k pairs takes longer than the launch it saves (26 + 7 -> 58 us); a few thousand do not
    const bool scatter = rjobs_dev && index && owner && max_n <= 4096u;
    auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(n_jobs)), dim3(kSelThreads), 0, st, jobs_dev, dist2, scale, thr, rjobs_dev, index,
                         scatter ? owner : nullptr);
    };
    if (max_n <= 4u * kSelThreads) go(rejection_select_reg_kernel<4>);
    else if (max_n <= 8u * kSelThreads) go(rejection_select_reg_kernel<8>);
    else if (max_n <= 20u * kSelThreads) go(rejection_select_reg_kernel<20>);
    else go(rejection_select_reg_kernel<32>);
    return scatter;
  }
  if (max_n < 65536u || !hists) {
    // the attribute belongs to the DEVICE the kernel is loaded on: asked once per device (a multi-device handle launches this
    // from one thread per device, each with its own current device)
    static std::atomic<unsigned long long> asked{0}, granted{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (static_cast<unsigned>(dev) & 63u);
    if (!(asked.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(rejection_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess)
        granted.fetch_or(bit, std::memory_order_release);
      asked.fetch_or(bit, std::memory_order_release);
    }
    const bool big_lds = (granted.load(std::memory_order_acquire) & bit) != 0;
    const uint32_t cache_n = std::min<uint32_t>(max_n, big_lds ? 32768u : 12288u);
    // one workgroup offering 20k pairs takes longer than the launch it saves (26 + 7 -> 58 us); a few thousand do not
    const bool scatter = rjobs_dev && index && owner && max_n <= 4096u;
    hipLaunchKernelGGL(rejection_select_kernel, dim3(static_cast<uint32_t>(n_jobs)), dim3(kSelThreads), cache_n * sizeof(float), st, jobs_dev,
                       dist2, scale, thr, cache_n, rjobs_dev, index, scatter ? owner : nullptr);
    return scatter;
  }
  for (int pass = 0; pass < 6; ++pass) {
    launch_select_hist(stream, jobs_dev, n_jobs, max_n, dist2, hists, states, pass);
    launch_select_pick(stream, jobs_dev, n_jobs, hists, states, pass, scale, thr);
  }
  return false;
}

// the two halves of one radix pass of the many-workgroup selection, separately: a slab-sharded pass adds the histograms
// up across ranks in between
void launch_select_hist(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t max_n, const float* dist2, uint32_t* hists,
                        uint32_t* states, int pass) {
  const uint32_t chunks = std::max<uint32_t>(1u, std::min<uint32_t>(256u, (max_n + 8u * kSelThreads - 1) / (8u * kSelThreads)));  // >= 8 values per thread
  hipLaunchKernelGGL(select_hist_kernel, dim3(chunks, static_cast<uint32_t>(n_jobs)), dim3(kSelThreads), 0, static_cast<hipStream_t>(stream),
                     jobs_dev, dist2, hists, states, pass);
}
void launch_select_pick(void* stream, const JobDesc* jobs_dev, int n_jobs, uint32_t* hists, uint32_t* states, int pass, float scale,
                        float* thr) {
  hipLaunchKernelGGL(select_pick_kernel, dim3(static_cast<uint32_t>(n_jobs)), dim3(kSelThreads), 0, static_cast<hipStream_t>(stream), jobs_dev,
                     hists, states, pass, scale, thr);
}

void launch_picky_scatter(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs,
                          uint32_t max_src_blocks, const int32_t* index, const float* dist2, const float* thr,
                          unsigned long long* owner) {
  hipLaunchKernelGGL(picky_scatter_kernel, dim3(max_src_blocks, static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), jobs_dev, rjobs_dev, index, dist2, thr, owner);
}

void launch_target_rows(void* stream, DevicePlanes level, uint32_t level_off, uint32_t n, DevicePlanes normals, uint32_t normal_off,
                        uint32_t step, void* rows) {
  if (!n) return;
  hipLaunchKernelGGL(target_rows_kernel, dim3((n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream),
                     level.x, level.y, level.z, level_off, n, normals.x, normals.y, normals.z, normal_off, step, static_cast<float4*>(rows));
}

void launch_ref_accumulate(void* stream, const JobDesc* jobs_dev, const RefJobDesc* rjobs_dev, int n_jobs,
                           uint32_t max_src_blocks, DevicePlanes src, const void* source_rows, const void* target_rows,
                           const int32_t* index, const float* dist2, const float* thr, const unsigned long long* owner,
                           uint8_t* selected, double* partials, const PassTail& tail) {
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(max_src_blocks, static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0,
                       static_cast<hipStream_t>(stream), jobs_dev, rjobs_dev, src.x, src.y, src.z, static_cast<const float4*>(source_rows),
                       static_cast<const float4*>(target_rows), index, dist2, thr, owner, selected, partials, tail);
  };
  if (source_rows) go(ref_accumulate_kernel<true>);
  else go(ref_accumulate_kernel<false>);
}

void launch_ref_final_reduce(void* stream, const JobDesc* jobs_dev, int n_jobs, const double* partials, double* sums, const float* thr,
                             uint32_t* overflow_count, uint32_t overflow_cap) {
  hipLaunchKernelGGL(ref_final_reduce_kernel, dim3(static_cast<uint32_t>(n_jobs)), dim3(kBlockThreads), 0,
                     static_cast<hipStream_t>(stream), jobs_dev, partials, sums, thr, overflow_count, overflow_cap);
}

#ifdef MI355_ICP_DIAG
extern "C" int mi355_icp_diag_ref_begin(void) {
  unsigned long long* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), 4096 * 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemset(buf, 0, 4096 * 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  const unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_ref_n), &zero, sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_ref), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int mi355_icp_diag_ref_read(unsigned long long* out, unsigned int* n) {
  unsigned long long* buf = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_diag_ref), sizeof(buf)) != hipSuccess || !buf) return 1;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_diag_ref_n), sizeof(*n)) != hipSuccess) return 1;
  return hipMemcpy(out, buf, 4096 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace mi355

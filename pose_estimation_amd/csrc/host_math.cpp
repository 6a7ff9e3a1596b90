// host_math.cpp — the host half of one ICP iteration: Kabsch (3x3 SVD in
// double), the convergence rule, and pose packing.  The design keeps these on
// the CPU (BASELINE.json north_star: "the 3x3 SVD done on the host"): they see
// 17 doubles per job per iteration.  In the reference this arithmetic is inside
// cv::ppf_match_3d::ICP::registerModelToScene
// (pose_estimation/src/opencv_surface_match.cpp:94); the only in-repo
// centroid+SVD is plane_fit (chessboard_pose_estimator.cpp:82-124).
#include <cmath>
#include <cstdlib>
#include <string>
#include <cstring>

#include "../../include/mi355_icp.h"
#include "icp_internal.h"

namespace mi355 {

// Which points of the slab [begin, begin + count) of a cloud of n_total points belong to the pyramid level of the WHOLE
// cloud that keeps every step-th point (the global indices j * step, j < n_total / step): the first one's offset inside the
// slab, how many, the level index j of the first, and the size of the whole level.
void level_slab(uint64_t n_total, uint64_t begin, uint64_t count, uint64_t step, uint64_t* first_local, uint64_t* n_local, uint64_t* index_base,
                uint64_t* n_level_total) {
  const uint64_t n_lvl = n_total / step;
  const uint64_t j_lo = (begin + step - 1) / step;
  uint64_t j_hi = (begin + count + step - 1) / step;  // first j with j * step >= begin + count
  if (j_hi > n_lvl) j_hi = n_lvl;
  *n_local = j_hi > j_lo ? j_hi - j_lo : 0;
  *first_local = *n_local ? j_lo * step - begin : 0;
  *index_base = j_lo < n_lvl ? j_lo : n_lvl;  // a slab past the level's last point numbers from the level's end
  *n_level_total = n_lvl;
}


// cvRound as upstream evaluates it (cvtsd2si / lrint under the default rounding mode): half to even.
static long cv_round(double v) { return std::lrint(v); }

uint32_t upstream_step(uint64_t n_model, int level) {
  const double div = std::pow(2.0, static_cast<double>(level));
  const long num = cv_round(static_cast<double>(n_model) / div);
  if (num < 1) return 0;
  const long step = cv_round(static_cast<double>(n_model) / static_cast<double>(num));
  return step < 1 ? 1u : static_cast<uint32_t>(step);
}

int upstream_max_iterations(int iterations, int level) { return static_cast<int>(cv_round(static_cast<double>(iterations) / static_cast<double>(level + 1))); }

// One-sided (Hestenes) Jacobi SVD of a 3x3: rotates column pairs of G = H V until
// they are mutually orthogonal; then column norms are the singular values and
// the normalised columns are U.  Robust for rank-deficient H (coplanar /
// collinear clouds): vanishing columns are completed by cross products.
static void svd3(const double H[3][3], double U[3][3], double S[3], double V[3][3]) {
  double G[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      G[i][j] = H[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; ++k) {
          alpha += G[k][p] * G[k][p];
          beta += G[k][q] * G[k][q];
          gamma += G[k][p] * G[k][q];
        }
        // columns count as orthogonal at 1e-15 of their norms' product - a few ulps of what the dot product can resolve.  (Rounds
        // 1-3 asked for 1e-17, BELOW that noise: the loop then ran until the noise happened to vanish, 35 sweeps on average
        // instead of 6, 3.3 us per solve - 0.21 ms of host time per iteration of the 64-object batch, where the device waits.)
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; ++k) {
          double gp = G[k][p], gq = G[k][q];
          G[k][p] = c * gp - s * gq;
          G[k][q] = s * gp + c * gq;
          double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double nrm[3];
  for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
  // order columns by decreasing singular value
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (nrm[ord[j]] > nrm[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double Vo[3][3], Go[3][3];
  for (int c = 0; c < 3; ++c) {
    S[c] = nrm[ord[c]];
    for (int r = 0; r < 3; ++r) { Vo[r][c] = V[r][ord[c]]; Go[r][c] = G[r][ord[c]]; }
  }
  std::memcpy(V, Vo, sizeof(Vo));
  const double tiny = 1e-12 * S[0];
  int rank = 0;
  for (int c = 0; c < 3; ++c) {
    if (S[0] > 0.0 && S[c] > tiny) {
      for (int r = 0; r < 3; ++r) U[r][c] = Go[r][c] / S[c];
      rank = c + 1;
    } else {
      break;
    }
  }
  auto cross_into = [&](int a, int b, int out) {
    U[0][out] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
    U[1][out] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
    U[2][out] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
  };
  if (rank == 0) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) U[r][c] = (r == c) ? 1.0 : 0.0;
  } else if (rank == 1) {
    int m = 0;
    if (std::fabs(U[1][0]) < std::fabs(U[m][0])) m = 1;
    if (std::fabs(U[2][0]) < std::fabs(U[m][0])) m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    double dot = U[m][0], b[3], nb = 0;
    for (int r = 0; r < 3; ++r) { b[r] = e[r] - dot * U[r][0]; nb += b[r] * b[r]; }
    nb = std::sqrt(nb);
    for (int r = 0; r < 3; ++r) U[r][1] = b[r] / nb;
    cross_into(0, 1, 2);
  } else if (rank == 2) {
    cross_into(0, 1, 2);
  }
}

static double det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

int solve_rigid(const double* sums, double* pose) {
  for (int k = 0; k < MI355_ICP_NSUMS; ++k)
    if (!std::isfinite(sums[k])) return MI355_ICP_ERR_NUMERIC;
  const double n = sums[0];
  if (n < 3.0) return MI355_ICP_ERR_TOO_FEW_PAIRS;
  double sbar[3], tbar[3], H[3][3];
  for (int a = 0; a < 3; ++a) {
    sbar[a] = sums[1 + a] / n;
    tbar[a] = sums[4 + a] / n;
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) H[a][b] = sums[7 + 3 * a + b] - sums[1 + a] * sums[4 + b] / n;
  double U[3][3], S[3], V[3][3];
  svd3(H, U, S, V);
  const double sgn = (det3(V) * det3(U) < 0.0) ? -1.0 : 1.0;
  double out[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + sgn * V[i][2] * U[j][2];
    out[4 * i + 3] = tbar[i] - (out[4 * i + 0] * sbar[0] + out[4 * i + 1] * sbar[1] + out[4 * i + 2] * sbar[2]);
  }
  out[12] = out[13] = out[14] = 0.0;
  out[15] = 1.0;
  for (int k = 0; k < 16; ++k)
    if (!std::isfinite(out[k])) return MI355_ICP_ERR_NUMERIC;
  std::memcpy(pose, out, sizeof(out));
  return MI355_ICP_OK;
}

int host_update(const mi355_icp_params* p, const double* sums, int iteration, double* pose, double* prev_mse,
                int* out_done, int* out_converged) {
  double next[16];
  int st = solve_rigid(sums, next);
  if (st != MI355_ICP_OK) return st;
  const double mse = sums[16] / sums[0];
  // incremental transform D = P_next * P_prev^-1 = [Rn Ro^T | tn - Rn Ro^T to]
  double D[3][3];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k)
      D[i][k] = next[4 * i + 0] * pose[4 * k + 0] + next[4 * i + 1] * pose[4 * k + 1] + next[4 * i + 2] * pose[4 * k + 2];
  double dt2 = 0.0;
  for (int i = 0; i < 3; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += D[i][k] * pose[4 * k + 3];
    double d = next[4 * i + 3] - acc;
    dt2 += d * d;
  }
  const double one_minus_cos = 1.0 - 0.5 * ((D[0][0] + D[1][1] + D[2][2]) - 1.0);
  int conv = 0;
  if (p->transformation_epsilon > 0.0 && dt2 <= p->transformation_epsilon && one_minus_cos <= p->rotation_epsilon)
    conv = 1;
  if (p->relative_mse_epsilon > 0.0 && iteration > 0 &&
      std::fabs(mse - *prev_mse) <= p->relative_mse_epsilon * (*prev_mse))
    conv = 1;
  std::memcpy(pose, next, sizeof(next));
  *prev_mse = mse;
  *out_converged = conv;
  *out_done = (conv || iteration + 1 >= p->max_iterations) ? 1 : 0;
  return MI355_ICP_OK;
}

// ---- reference mode: 6x6 point-to-plane normal equations -------------------------------------------------
// Minimum-norm least-squares solution through the eigen-decomposition of the symmetric 6x6 (cyclic Jacobi),
// which is what cv::solve(A, b, x, DECOMP_SVD) returns for A^T A x = A^T b up to rounding (upstream solves the
// tall system; reference call site opencv_surface_match.cpp:94).
int solve_point_to_plane(const double* sums, double* x) {
  for (int k = 0; k < kRefSolveSums; ++k)
    if (!std::isfinite(sums[k])) return MI355_ICP_ERR_NUMERIC;
  constexpr int n = 6;
  double M[n][n], E[n][n], g[n];
  int k = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) M[i][j] = M[j][i] = sums[k++];
  for (int i = 0; i < n; ++i) {
    g[i] = sums[21 + i];
    for (int j = 0; j < n; ++j) E[i][j] = (i == j) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += std::fabs(M[i][i]);
      for (int j = i + 1; j < n; ++j) off += std::fabs(M[i][j]);
    }
    if (off <= 1e-300 || off <= 1e-30 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (M[p][q] == 0.0) continue;
        const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int r = 0; r < n; ++r) {
          const double a = M[r][p], b = M[r][q];
          M[r][p] = c * a - s * b;
          M[r][q] = s * a + c * b;
        }
        for (int r = 0; r < n; ++r) {
          const double a = M[p][r], b = M[q][r];
          M[p][r] = c * a - s * b;
          M[q][r] = s * a + c * b;
        }
        for (int r = 0; r < n; ++r) {
          const double a = E[r][p], b = E[r][q];
          E[r][p] = c * a - s * b;
          E[r][q] = s * a + c * b;
        }
      }
  }
  double wmax = 0.0;
  for (int i = 0; i < n; ++i) wmax = std::fmax(wmax, std::fabs(M[i][i]));
  for (int i = 0; i < n; ++i) x[i] = 0.0;
  for (int e = 0; e < n; ++e) {
    const double w = M[e][e];
    if (!(w > 1e-12 * wmax)) continue;  // rank-deficient direction: leave it out (minimum-norm solution)
    double proj = 0.0;
    for (int i = 0; i < n; ++i) proj += E[i][e] * g[i];
    for (int i = 0; i < n; ++i) x[i] += E[i][e] * proj / w;
  }
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(x[i])) return MI355_ICP_ERR_NUMERIC;
  return MI355_ICP_OK;
}

// The same solve for up to 8 systems AT ONCE, one per SIMD lane (GCC vector extensions, compiled for AVX-512; other CPUs take
// the scalar loop).  A reference-mode pass ends with one 6x6 solve per candidate pose while the device
// has nothing to do; 6 of them in sequence were 14 us of every ~65 us coarse-level pass.  Every lane performs EXACTLY the
// operations of solve_point_to_plane in the same order (+, -, *, /, sqrt are correctly rounded in every instruction set,
// contraction is off): the scalar version's branches become masks - a skipped rotation or a lane that has already
// converged keeps its values through a blend, never through an "identity rotation" - so results are bit-identical lane
// for lane (tests/test_reference_mode_cpu.py compares the two on well-conditioned, rank-deficient and non-finite input).
#pragma GCC diagnostic ignored "-Wpsabi"  // the vector helpers below are always inlined: no 64-byte value crosses a call
namespace {
typedef double vd8 __attribute__((vector_size(64)));
typedef long long vm8 __attribute__((vector_size(64)));  // comparison results: -1 / 0 per lane

__attribute__((always_inline)) inline vd8 splat(double v) { return vd8{v, v, v, v, v, v, v, v}; }
__attribute__((always_inline)) inline vd8 vabs(vd8 v) { return v < splat(0.0) ? -v : v; }  // fabs for everything but the sign of zero, which is never looked at
__attribute__((always_inline)) inline vd8 vsqrt(vd8 v) {
  vd8 r;
  for (int l = 0; l < 8; ++l) r[l] = std::sqrt(v[l]);
  return r;
}
__attribute__((always_inline)) inline bool any(vm8 m) {
  long long acc = 0;
  for (int l = 0; l < 8; ++l) acc |= m[l];
  return acc != 0;
}
}  // namespace

namespace {
__attribute__((always_inline)) inline void solve_batch_body(const double* sums, int n, double* x, int* status, size_t stride) {
  constexpr int N = 6;
  for (int base = 0; base < n; base += 8) {
    const int lanes = n - base < 8 ? n - base : 8;
    vd8 M[N][N], E[N][N], g[N];
    vm8 bad = vm8{0, 0, 0, 0, 0, 0, 0, 0};
    {
      double in[kRefSolveSums][8];
      for (int k = 0; k < kRefSolveSums; ++k)
        for (int l = 0; l < 8; ++l) in[k][l] = l < lanes ? sums[static_cast<size_t>(base + l) * stride + k] : (k == 0 || k == 6 || k == 11 || k == 15 || k == 18 || k == 20 ? 1.0 : 0.0);
      for (int k = 0; k < kRefSolveSums; ++k)
        for (int l = 0; l < 8; ++l)
          if (!std::isfinite(in[k][l])) bad[l] = -1;
      int k = 0;
      for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) {
          vd8 v;
          for (int l = 0; l < 8; ++l) v[l] = in[k][l];
          M[i][j] = M[j][i] = v;
          ++k;
        }
      for (int i = 0; i < N; ++i) {
        for (int l = 0; l < 8; ++l) g[i][l] = in[21 + i][l];
        for (int j = 0; j < N; ++j) E[i][j] = splat(i == j ? 1.0 : 0.0);
      }
    }
    vm8 act = ~bad;  // lanes still sweeping
    for (int sweep = 0; sweep < 100; ++sweep) {
      vd8 off = splat(0.0), diag = splat(0.0);
      for (int i = 0; i < N; ++i) {
        diag += vabs(M[i][i]);
        for (int j = i + 1; j < N; ++j) off += vabs(M[i][j]);
      }
      act &= ~((off <= splat(1e-300)) | (off <= splat(1e-30) * diag));
      if (!any(act)) break;
      for (int p = 0; p < N - 1; ++p)
        for (int q = p + 1; q < N; ++q) {
          const vm8 go = act & (M[p][q] != splat(0.0));
          if (!any(go)) continue;
          const vd8 theta = (M[q][q] - M[p][p]) / (splat(2.0) * M[p][q]);
          const vd8 t = (theta >= splat(0.0) ? splat(1.0) : splat(-1.0)) / (vabs(theta) + vsqrt(theta * theta + splat(1.0)));
          const vd8 c = splat(1.0) / vsqrt(t * t + splat(1.0)), sn = t * c;
          for (int r = 0; r < N; ++r) {
            const vd8 a = M[r][p], b = M[r][q];
            M[r][p] = go ? c * a - sn * b : a;
            M[r][q] = go ? sn * a + c * b : b;
          }
          for (int r = 0; r < N; ++r) {
            const vd8 a = M[p][r], b = M[q][r];
            M[p][r] = go ? c * a - sn * b : a;
            M[q][r] = go ? sn * a + c * b : b;
          }
          for (int r = 0; r < N; ++r) {
            const vd8 a = E[r][p], b = E[r][q];
            E[r][p] = go ? c * a - sn * b : a;
            E[r][q] = go ? sn * a + c * b : b;
          }
        }
    }
    vd8 wmax = splat(0.0);
    for (int i = 0; i < N; ++i) {
      const vd8 w = vabs(M[i][i]);
      wmax = (wmax < w || wmax != wmax) ? w : wmax;  // std::fmax(wmax, w) with wmax never NaN in a live lane
    }
    vd8 xs[N];
    for (int i = 0; i < N; ++i) xs[i] = splat(0.0);
    for (int e = 0; e < N; ++e) {
      const vd8 w = M[e][e];
      const vm8 use = w > splat(1e-12) * wmax;  // rank-deficient direction: left out (minimum-norm solution)
      vd8 proj = splat(0.0);
      for (int i = 0; i < N; ++i) proj += E[i][e] * g[i];
      for (int i = 0; i < N; ++i) xs[i] = use ? xs[i] + E[i][e] * proj / w : xs[i];
    }
    for (int l = 0; l < lanes; ++l) {
      int st = bad[l] ? MI355_ICP_ERR_NUMERIC : MI355_ICP_OK;
      for (int i = 0; i < N; ++i) {
        x[static_cast<size_t>(base + l) * N + i] = xs[i][l];
        if (!std::isfinite(xs[i][l])) st = MI355_ICP_ERR_NUMERIC;
      }
      status[base + l] = st;
    }
  }
}
// The body is worth having with 512-bit registers only: GCC lowers its 64-byte vector compares and blends to scalar code on
// AVX2 and below (measured: 3-8 x SLOWER than six scalar solves), so without AVX-512 the batch is the scalar loop.
__attribute__((target("avx512f,avx512dq,avx512vl,avx512bw"))) void solve_batch_avx512(const double* sums, int n, double* x, int* status, size_t stride) {
  solve_batch_body(sums, n, x, status, stride);
}
void solve_batch_scalar(const double* sums, int n, double* x, int* status, size_t stride) {
  for (int k = 0; k < n; ++k) status[k] = solve_point_to_plane(sums + static_cast<size_t>(k) * stride, x + static_cast<size_t>(k) * 6);
}
}  // namespace

void solve_point_to_plane_batch(const double* sums, int n, double* x, int* status, size_t stride) {
  static void (*const impl)(const double*, int, double*, int*, size_t) = [] {
    __builtin_cpu_init();
    const char* want = std::getenv("MI355_ICP_SOLVE_ISA");  // "scalar": one system at a time whatever the CPU has (A/B, tests)
    if (want && std::string(want) == "scalar") return &solve_batch_scalar;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512bw"))
      return &solve_batch_avx512;
    return &solve_batch_scalar;
  }();
  impl(sums, n, x, status, stride);
}

// ---- RANSAC plane fit, host half (mi355_icp_scene_fit_plane; reference pose_estimation.cpp:285-297) ----------------
// PCL's RandomSampleConsensus::computeModel replayed over the support counts the device produced for all hypotheses at
// once: a hypothesis is looked at only while iterations < k; more support than any before makes it the best and
// shortens k = log(1 - probability) / log(1 - w^3).  Returns the best hypothesis (-1: none) and the iterations spent.
int plane_choose(const uint32_t* counts, const float* planes4, uint32_t n_hyp, uint32_t n_points, double probability, int* iterations) {
  int best = -1, it = 0;
  uint32_t best_count = 0;
  double k = 1.0;
  const double log_probability = std::log(1.0 - probability);
  const double eps = 2.220446049250313e-16;
  for (uint32_t h = 0; h < n_hyp; ++h) {
    if (!(static_cast<double>(it) < k)) break;
    if (planes4[4 * h + 3] != planes4[4 * h + 3]) continue;  // no valid triple: skipped, not an iteration (as PCL)
    if (best < 0 || counts[h] > best_count) {
      best = static_cast<int>(h);
      best_count = counts[h];
      const double w = static_cast<double>(best_count) / static_cast<double>(n_points);
      double p_no_outliers = 1.0 - w * w * w;
      p_no_outliers = std::fmax(eps, p_no_outliers);
      p_no_outliers = std::fmin(1.0 - eps, p_no_outliers);
      k = log_probability / std::log(p_no_outliers);
    }
    ++it;
  }
  if (iterations) *iterations = it;
  return best;
}

// Least-squares plane of the inliers summarised by (n, sum p, sum p p^T upper triangle): normal = eigenvector of the
// smallest eigenvalue of the covariance (cyclic Jacobi on the symmetric 3x3), pointing the way the hypothesis does.
int plane_refine(const double* m, const float* hyp, float* out) {
  for (int k = 0; k < 10; ++k)
    if (!std::isfinite(m[k])) return MI355_ICP_ERR_NUMERIC;
  if (m[0] < 3.0) return MI355_ICP_ERR_TOO_FEW_PAIRS;
  const double inv = 1.0 / m[0];
  const double c[3] = {m[1] * inv, m[2] * inv, m[3] * inv};
  double A[3][3] = {{m[4] * inv - c[0] * c[0], m[5] * inv - c[0] * c[1], m[6] * inv - c[0] * c[2]},
                    {0, m[7] * inv - c[1] * c[1], m[8] * inv - c[1] * c[2]},
                    {0, 0, m[9] * inv - c[2] * c[2]}};
  A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    const double diag = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-30 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int r = 0; r < 3; ++r) {
          const double a = A[r][p], b = A[r][q];
          A[r][p] = cs * a - sn * b;
          A[r][q] = sn * a + cs * b;
        }
        for (int r = 0; r < 3; ++r) {
          const double a = A[p][r], b = A[q][r];
          A[p][r] = cs * a - sn * b;
          A[q][r] = sn * a + cs * b;
        }
        for (int r = 0; r < 3; ++r) {
          const double a = V[r][p], b = V[r][q];
          V[r][p] = cs * a - sn * b;
          V[r][q] = sn * a + cs * b;
        }
      }
  }
  int e = 0;
  if (A[1][1] < A[e][e]) e = 1;
  if (A[2][2] < A[e][e]) e = 2;
  double v[3] = {V[0][e], V[1][e], V[2][e]};
  const double nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (!(nrm > 0.0)) return MI355_ICP_ERR_NUMERIC;
  const double sign = (v[0] * hyp[0] + v[1] * hyp[1] + v[2] * hyp[2]) < 0.0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) v[k] = sign * v[k] / nrm;
  out[0] = static_cast<float>(v[0]);
  out[1] = static_cast<float>(v[1]);
  out[2] = static_cast<float>(v[2]);
  out[3] = static_cast<float>(-(v[0] * c[0] + v[1] * c[1] + v[2] * c[2]));
  return MI355_ICP_OK;
}

// R = Rx(x0) * (Ry(x1) * Rz(x2)), t = (x3, x4, x5): upstream's getTransformMat / eulerToDCM convention as recalled.
void euler_pose(const double* x, double* P) {
  const double c0 = std::cos(x[0]), s0 = std::sin(x[0]), c1 = std::cos(x[1]), s1 = std::sin(x[1]), c2 = std::cos(x[2]),
               s2 = std::sin(x[2]);
  const double RyRz[9] = {c1 * c2, -c1 * s2, s1, s2, c2, 0.0, -s1 * c2, s1 * s2, c1};
  const double Rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      P[4 * i + j] = Rx[3 * i] * RyRz[j] + Rx[3 * i + 1] * RyRz[3 + j] + Rx[3 * i + 2] * RyRz[6 + j];
    P[4 * i + 3] = x[3 + i];
  }
  P[12] = P[13] = P[14] = 0.0;
  P[15] = 1.0;
}

void mat4_mul(const double* a, const double* b, double* c) {
  double t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += a[4 * i + k] * b[4 * k + j];
      t[4 * i + j] = acc;
    }
  std::memcpy(c, t, sizeof(t));
}

// 0.5 * copysign(1, sign_arg) * sqrt(a - b - c + 1) with float arithmetic inside,
// as the reference evaluates it on xt::xarray<float>
// (chessboard_pose_estimator.cpp:196-207).  Deviation: the sqrt argument is
// clamped at 0; the reference returns NaN when rounding takes it below 0.
static float quat_axis(float sign_arg, float a, float b, float c) {
  float arg = ((a - b) - c) + 1.0f;
  if (arg < 0.0f) arg = 0.0f;
  const float r = std::sqrt(arg);
  return static_cast<float>(0.5 * std::copysign(1.0, static_cast<double>(sign_arg)) * static_cast<double>(r));
}

}  // namespace mi355

extern "C" {

unsigned mi355_icp_abi_version(void) { return MI355_ICP_ABI_VERSION; }

void mi355_icp_default_params(mi355_icp_params* p) {
  if (!p) return;
  p->max_iterations = 50;
  p->max_correspondence_distance = 0.0f;
  p->transformation_epsilon = 0.0;
  p->rotation_epsilon = 1e-5;
  p->relative_mse_epsilon = 0.0;
  p->flags = 0;
}

int mi355_icp_params_from_reference(int iterations, float tolerance, float rejection_scale, int num_levels,
                                    mi355_icp_params* p) {
  if (!p || iterations < 1) return MI355_ICP_ERR_INVALID_ARGUMENT;
  (void)rejection_scale;  // point-to-plane / MAD-rejection mode: SURVEY.md 8f rank 1
  (void)num_levels;       // pyramid: same
  mi355_icp_default_params(p);
  p->max_iterations = iterations;
  p->relative_mse_epsilon = tolerance > 0.0f ? static_cast<double>(tolerance) : 0.0;
  return MI355_ICP_OK;
}

int mi355_icp_slab_range(size_t n, int rank, int n_ranks, size_t* begin, size_t* count) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !begin || !count) return MI355_ICP_ERR_INVALID_ARGUMENT;
  const size_t base = n / static_cast<size_t>(n_ranks), extra = n % static_cast<size_t>(n_ranks);
  const size_t r = static_cast<size_t>(rank);
  *begin = r * base + (r < extra ? r : extra);
  *count = base + (r < extra ? 1 : 0);
  return MI355_ICP_OK;
}

int mi355_icp_level_slab(size_t n_total, size_t begin, size_t count, int level, size_t* first_local, size_t* n_level_local,
                         size_t* level_index_base, size_t* n_level_total) {
  if (level < 0 || level > 31 || begin > n_total || count > n_total - begin) return MI355_ICP_ERR_INVALID_ARGUMENT;
  return mi355_icp_level_slab_step(n_total, begin, count, static_cast<size_t>(1) << level, first_local, n_level_local, level_index_base, n_level_total);
}

int mi355_icp_level_slab_step(size_t n_total, size_t begin, size_t count, size_t step, size_t* first_local, size_t* n_level_local,
                              size_t* level_index_base, size_t* n_level_total) {
  if (step < 1 || begin > n_total || count > n_total - begin) return MI355_ICP_ERR_INVALID_ARGUMENT;
  uint64_t f = 0, n = 0, base = 0, all = 0;
  mi355::level_slab(n_total, begin, count, step, &f, &n, &base, &all);
  if (first_local) *first_local = static_cast<size_t>(f);
  if (n_level_local) *n_level_local = static_cast<size_t>(n);
  if (level_index_base) *level_index_base = static_cast<size_t>(base);
  if (n_level_total) *n_level_total = static_cast<size_t>(all);
  return MI355_ICP_OK;
}

unsigned mi355_icp_upstream_step(size_t n_model, int level) {
  if (level < 0 || level > 62) return 0;
  return mi355::upstream_step(n_model, level);
}

int mi355_icp_solve_rigid(const double sums[MI355_ICP_NSUMS], double out_pose[16]) {
  if (!sums || !out_pose) return MI355_ICP_ERR_INVALID_ARGUMENT;
  return mi355::solve_rigid(sums, out_pose);
}

int mi355_icp_host_update(const mi355_icp_params* params, const double sums[MI355_ICP_NSUMS], int iteration,
                          double pose[16], double* prev_mse, int* out_done, int* out_converged) {
  if (!params || !sums || !pose || !prev_mse || !out_done || !out_converged || iteration < 0 ||
      params->max_iterations < 1)
    return MI355_ICP_ERR_INVALID_ARGUMENT;
  return mi355::host_update(params, sums, iteration, pose, prev_mse, out_done, out_converged);
}

int mi355_icp_pose_to_pose7(const double pose[16], float out7[7], int reference_packing) {
  if (!pose || !out7) return MI355_ICP_ERR_INVALID_ARGUMENT;
  float h[3][4];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) h[i][j] = static_cast<float>(pose[4 * i + j]);
  const float ox = mi355::quat_axis(h[2][1] - h[1][2], h[0][0], h[1][1], h[2][2]);
  const float oy = mi355::quat_axis(h[0][2] - h[2][0], h[1][1], h[2][2], h[0][0]);
  const float oz = mi355::quat_axis(h[1][0] - h[0][1], h[2][2], h[0][0], h[1][1]);
  float warg = ((h[0][0] + h[1][1]) + h[2][2]) + 1.0f;
  if (warg < 0.0f) warg = 0.0f;
  const float w = static_cast<float>(0.5 * static_cast<double>(std::sqrt(warg)));
  out7[0] = h[0][3];
  out7[1] = h[1][3];
  out7[2] = h[2][3];
  out7[3] = ox;
  out7[4] = oy;
  if (reference_packing) {
    // opencv_surface_match.cpp:138-141: pose[3]=q[1]; pose[4]=q[2]; pose[6]=q[3]; pose[6]=q[0];
    // (q = {w,x,y,z}) -> orientation.z is never written, qz is overwritten by qw.
    out7[5] = 0.0f;
    out7[6] = w;
  } else {
    out7[5] = oz;
    out7[6] = w;
  }
  return MI355_ICP_OK;
}

}  // extern "C"

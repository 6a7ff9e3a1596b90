/*
 * icp_oracle.c — CPU restatement of the point-to-point ICP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pose_estimation_amd/ may include,
 * link, import or execute this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker and the CPU baseline.
 *
 * PARITY UNPINNED.  The reference (yumi-crew/pose_estimation) holds no tests,
 * golden vectors or sample clouds for this path, and the arithmetic it runs
 * lives in an un-vendored, un-pinned third-party module
 * (opencv_contrib surface_matching, cv::ppf_match_3d::ICP; call site
 * pose_estimation/src/opencv_surface_match.cpp:85,94; README.md:14-22 clones
 * the default branch) that is not present in the build container, so it can be
 * neither compiled nor run here.  This file therefore restates the algorithm
 * the design brief (BASELINE.json north_star) mandates,
 * anchored on the reference's call-site contract:
 *   - clouds are float32 xyz rows            (pose_estimation.cpp:263-276)
 *   - poses are row-major 4x4 double, model->scene, refined in place together
 *     with a scalar residual                 (opencv_surface_match.cpp:80-94)
 *   - 7-float pose {x,y,z,qx,qy,qz,qw}       (pose_estimation.hpp:83-88) with the
 *     matrix->quaternion formula of          chessboard_pose_estimator.cpp:196-207
 *     and the packing of                     opencv_surface_match.cpp:133-142
 * and cross-checked in tests/ against scipy.spatial.cKDTree (exact NN) and
 * numpy.linalg.svd (Kabsch), which are independent of this file.
 *
 * Numeric definition (the thing the GPU path must reproduce):
 *   pose -> float32 (round to nearest) 3x4 [R|t]
 *   s'   = ((r0*x + r1*y) + r2*z) + t      every product and sum rounded to float32, no FMA
 *   d2   = ((dx*dx) + (dy*dy)) + (dz*dz)   dx = s'x - tx etc., float32, no FMA
 *          (the accumulation order of FLANN's L2 functor on a non-FMA x86 build)
 *   nn   = lowest target index among the float32 minima of d2
 *   accept pair if d2 <= max_d2 (float32), max_d2 = maxdist*maxdist in float32
 *   sums in double over accepted pairs (layout: include/mi355_icp.h MI355_ICP_NSUMS)
 *   Kabsch: H = sum s0 t^T - (sum s0)(sum t)^T / n, H = U S V^T,
 *           R = V diag(1,1,det(V U^T)) U^T, t = tbar - R s0bar
 *   convergence: see oracle_host_update
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NSUMS 17

enum { OR_OK = 0, OR_INVALID = 1, OR_TOO_FEW = 5, OR_NUMERIC = 6 };

typedef struct oracle_params {
  int32_t max_iterations;
  float max_correspondence_distance;
  double transformation_epsilon;
  double rotation_epsilon;
  double relative_mse_epsilon;
  int32_t flags;
} oracle_params;

/* ------------------------------------------------------------------------ */
/* float32 building blocks (volatile-free; -ffp-contract=off keeps them exact) */

void oracle_pose_to_f32(const double pose[16], float out12[12]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) out12[4 * r + c] = (float)pose[4 * r + c];
}

static inline void xform(const float m[12], float x, float y, float z, float* ox, float* oy, float* oz) {
  float a, b;
  a = m[0] * x; b = m[1] * y; a = a + b; b = m[2] * z; a = a + b; *ox = a + m[3];
  a = m[4] * x; b = m[5] * y; a = a + b; b = m[6] * z; a = a + b; *oy = a + m[7];
  a = m[8] * x; b = m[9] * y; a = a + b; b = m[10] * z; a = a + b; *oz = a + m[11];
}

static inline float dist2(float qx, float qy, float qz, float tx, float ty, float tz) {
  float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  float a = dx * dx, b = dy * dy, c = dz * dz;
  a = a + b;
  return a + c;
}

void oracle_transform(const float pose12[12], const float* src_xyz, size_t n, float* out_xyz) {
  for (size_t i = 0; i < n; ++i)
    xform(pose12, src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2], &out_xyz[3 * i], &out_xyz[3 * i + 1],
          &out_xyz[3 * i + 2]);
}

/* ------------------------------------------------------------------------ */
/* exact brute-force 1-NN: lowest index among equal float32 minima           */

void oracle_nn_brute(const float* q_xyz, size_t nq, const float* t_xyz, size_t nt, int32_t* idx, float* d2,
                     int nthreads) {
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (long long i = 0; i < (long long)nq; ++i) {
    float qx = q_xyz[3 * i], qy = q_xyz[3 * i + 1], qz = q_xyz[3 * i + 2];
    float best = INFINITY;
    int32_t bj = 0;
    for (size_t j = 0; j < nt; ++j) {
      float d = dist2(qx, qy, qz, t_xyz[3 * j], t_xyz[3 * j + 1], t_xyz[3 * j + 2]);
      if (d < best) { best = d; bj = (int32_t)j; }
    }
    idx[i] = bj;
    d2[i] = best;
  }
  (void)nthreads;
}

/* ------------------------------------------------------------------------ */
/* exact KD-tree 1-NN with the same float32 distance and the same tie rule.
 * Bounding-box lower bounds are evaluated with the SAME rounded operations as
 * dist2, and rounding is monotone, so lb <= d2 of every point in the box holds
 * in floating point, not just in exact arithmetic; a subtree is skipped only
 * when lb > best (strict), and equal distances prefer the lower index, so the
 * result equals oracle_nn_brute bit for bit.                                */

typedef struct kd_node {
  float lo[3], hi[3];
  int32_t left, right; /* children, or -1 */
  int32_t begin, end;  /* leaf range into perm */
} kd_node;

typedef struct oracle_kdtree {
  size_t n;
  float* pts;    /* reordered xyz, leaf-contiguous */
  int32_t* perm; /* original index of reordered point */
  kd_node* nodes;
  int32_t n_nodes, cap_nodes;
} oracle_kdtree;

#define KD_LEAF 12

static int32_t kd_new_node(oracle_kdtree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(kd_node));
  }
  return t->n_nodes++;
}

static void kd_select(const float* xyz, int32_t* perm, int32_t lo, int32_t hi, int32_t k, int dim) {
  /* quickselect on perm[lo,hi) by coordinate dim, ties by index for determinism */
  while (hi - lo > 1) {
    int32_t mid = lo + (hi - lo) / 2;
    int32_t pv = perm[mid];
    float pvv = xyz[3 * (size_t)pv + dim];
    perm[mid] = perm[hi - 1];
    perm[hi - 1] = pv;
    int32_t store = lo;
    for (int32_t i = lo; i < hi - 1; ++i) {
      float v = xyz[3 * (size_t)perm[i] + dim];
      if (v < pvv || (v == pvv && perm[i] < pv)) {
        int32_t tmp = perm[i]; perm[i] = perm[store]; perm[store] = tmp;
        ++store;
      }
    }
    int32_t tmp = perm[store]; perm[store] = perm[hi - 1]; perm[hi - 1] = tmp;
    if (store == k) return;
    if (k < store) hi = store; else lo = store + 1;
  }
}

static int32_t kd_build_rec(oracle_kdtree* t, const float* xyz, int32_t lo, int32_t hi) {
  int32_t id = kd_new_node(t);
  float bl[3] = {INFINITY, INFINITY, INFINITY}, bh[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int32_t i = lo; i < hi; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = xyz[3 * (size_t)t->perm[i] + d];
      if (v < bl[d]) bl[d] = v;
      if (v > bh[d]) bh[d] = v;
    }
  for (int d = 0; d < 3; ++d) { t->nodes[id].lo[d] = bl[d]; t->nodes[id].hi[d] = bh[d]; }
  t->nodes[id].begin = lo;
  t->nodes[id].end = hi;
  t->nodes[id].left = t->nodes[id].right = -1;
  if (hi - lo > KD_LEAF) {
    int dim = 0;
    float ext = bh[0] - bl[0];
    if (bh[1] - bl[1] > ext) { ext = bh[1] - bl[1]; dim = 1; }
    if (bh[2] - bl[2] > ext) { ext = bh[2] - bl[2]; dim = 2; }
    if (ext > 0.0f) {
      int32_t mid = lo + (hi - lo) / 2;
      kd_select(xyz, t->perm, lo, hi, mid, dim);
      int32_t l = kd_build_rec(t, xyz, lo, mid);
      int32_t r = kd_build_rec(t, xyz, mid, hi);
      t->nodes[id].left = l;
      t->nodes[id].right = r;
    }
  }
  return id;
}

oracle_kdtree* oracle_kdtree_build(const float* t_xyz, size_t nt) {
  oracle_kdtree* t = (oracle_kdtree*)calloc(1, sizeof(*t));
  t->n = nt;
  t->perm = (int32_t*)malloc((nt ? nt : 1) * sizeof(int32_t));
  t->pts = (float*)malloc((nt ? nt : 1) * 3 * sizeof(float));
  for (size_t i = 0; i < nt; ++i) t->perm[i] = (int32_t)i;
  if (nt) kd_build_rec(t, t_xyz, 0, (int32_t)nt);
  for (size_t i = 0; i < nt; ++i) memcpy(&t->pts[3 * i], &t_xyz[3 * (size_t)t->perm[i]], 3 * sizeof(float));
  return t;
}

void oracle_kdtree_free(oracle_kdtree* t) {
  if (!t) return;
  free(t->perm); free(t->pts); free(t->nodes); free(t);
}

static inline float box_lb(const kd_node* nd, float qx, float qy, float qz) {
  float q[3] = {qx, qy, qz}, b[3];
  for (int d = 0; d < 3; ++d) {
    float v = 0.0f;
    if (q[d] < nd->lo[d]) v = q[d] - nd->lo[d];
    else if (q[d] > nd->hi[d]) v = q[d] - nd->hi[d];
    b[d] = v;
  }
  float a = b[0] * b[0], bb = b[1] * b[1], c = b[2] * b[2];
  a = a + bb;
  return a + c;
}

static void kd_query_one(const oracle_kdtree* t, float qx, float qy, float qz, int32_t* out_idx, float* out_d2) {
  /* same initial state and update rule as oracle_nn_brute: (index 0, +inf), replaced only by a strictly
   * smaller distance, or by an equal distance at a lower original index (the tree visits out of order). */
  float best = INFINITY;
  int32_t bj = 0;
  int32_t stack[128];
  int sp = 0;
  if (t->n) stack[sp++] = 0;
  while (sp) {
    const kd_node* nd = &t->nodes[stack[--sp]];
    if (box_lb(nd, qx, qy, qz) > best) continue;
    if (nd->left < 0) {
      for (int32_t i = nd->begin; i < nd->end; ++i) {
        float d = dist2(qx, qy, qz, t->pts[3 * (size_t)i], t->pts[3 * (size_t)i + 1], t->pts[3 * (size_t)i + 2]);
        int32_t j = t->perm[i];
        if (d < best || (d == best && j < bj)) { best = d; bj = j; }
      }
    } else {
      const kd_node* l = &t->nodes[nd->left];
      const kd_node* r = &t->nodes[nd->right];
      float ll = box_lb(l, qx, qy, qz), lr = box_lb(r, qx, qy, qz);
      /* visit the nearer child first (push it last) */
      if (ll <= lr) { stack[sp++] = nd->right; stack[sp++] = nd->left; }
      else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
    }
  }
  *out_idx = bj;
  *out_d2 = best;
}

void oracle_kdtree_query(const oracle_kdtree* t, const float* q_xyz, size_t nq, int32_t* idx, float* d2,
                         int nthreads) {
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
#endif
  for (long long i = 0; i < (long long)nq; ++i)
    kd_query_one(t, q_xyz[3 * i], q_xyz[3 * i + 1], q_xyz[3 * i + 2], &idx[i], &d2[i]);
  (void)nthreads;
}

/* ------------------------------------------------------------------------ */
/* covariance / centroid accumulation (sequential, index order, double)      */

void oracle_accumulate(const float* src0_xyz, const float* t_xyz, const int32_t* idx, const float* d2, size_t n,
                       float max_corr_dist, double sums[NSUMS]) {
  float max_d2 = INFINITY;
  if (max_corr_dist > 0.0f) max_d2 = max_corr_dist * max_corr_dist;
  for (int k = 0; k < NSUMS; ++k) sums[k] = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (!(d2[i] <= max_d2)) continue;
    const float* s = &src0_xyz[3 * i];
    const float* t = &t_xyz[3 * (size_t)idx[i]];
    sums[0] += 1.0;
    for (int a = 0; a < 3; ++a) {
      sums[1 + a] += (double)s[a];
      sums[4 + a] += (double)t[a];
      for (int b = 0; b < 3; ++b) sums[7 + 3 * a + b] += (double)s[a] * (double)t[b];
    }
    sums[16] += (double)d2[i];
  }
}

/* ------------------------------------------------------------------------ */
/* Kabsch via eigen-decomposition of H^T H (cyclic Jacobi), then U = H V / s
 * with Gram-Schmidt completion for vanishing singular values.               */

static void jacobi_eig3(double A[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; ++k) { /* A <- A J */
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- J^T A */
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

static double det3(double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

int oracle_solve_rigid(const double sums[NSUMS], double pose[16]) {
  for (int k = 0; k < NSUMS; ++k)
    if (!isfinite(sums[k])) return OR_NUMERIC;
  double n = sums[0];
  if (n < 3.0) return OR_TOO_FEW;
  double sb[3], tb[3], H[3][3];
  for (int a = 0; a < 3; ++a) { sb[a] = sums[1 + a] / n; tb[a] = sums[4 + a] / n; }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) H[a][b] = sums[7 + 3 * a + b] - sums[1 + a] * sums[4 + b] / n;
  /* H^T H = V S^2 V^T */
  double A[3][3], V[3][3], w[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += H[k][i] * H[k][j];
      A[i][j] = acc;
    }
  jacobi_eig3(A, V, w);
  /* sort descending */
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[ord[j]] > w[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double Vs[3][3], U[3][3], sv[3];
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][ord[c]];
    sv[c] = w[ord[c]] > 0 ? sqrt(w[ord[c]]) : 0.0;
  }
  double smax = sv[0];
  int rank = 0;
  for (int c = 0; c < 3; ++c) {
    double u[3];
    for (int r = 0; r < 3; ++r) u[r] = H[r][0] * Vs[0][c] + H[r][1] * Vs[1][c] + H[r][2] * Vs[2][c];
    /* orthogonalise against accepted columns (twice for safety) */
    for (int pass = 0; pass < 2; ++pass)
      for (int p = 0; p < rank; ++p) {
        double dot = u[0] * U[0][p] + u[1] * U[1][p] + u[2] * U[2][p];
        for (int r = 0; r < 3; ++r) u[r] -= dot * U[r][p];
      }
    double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (smax > 0 && nrm > 1e-9 * smax && c == rank) {
      for (int r = 0; r < 3; ++r) U[r][rank] = u[r] / nrm;
      ++rank;
    } else {
      break;
    }
  }
  /* complete U to an orthonormal basis */
  if (rank == 0) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { U[r][c] = (r == c); Vs[r][c] = (r == c); }
  } else if (rank == 1) {
    double a[3] = {U[0][0], U[1][0], U[2][0]};
    int m = 0;
    if (fabs(a[1]) < fabs(a[m])) m = 1;
    if (fabs(a[2]) < fabs(a[m])) m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    double dot = a[m];
    double b[3] = {e[0] - dot * a[0], e[1] - dot * a[1], e[2] - dot * a[2]};
    double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int r = 0; r < 3; ++r) U[r][1] = b[r] / nb;
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  } else if (rank == 2) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
  /* R = V diag(1,1,det(V U^T)) U^T */
  double dV = det3(Vs), dU = det3(U);
  double sgn = (dV * dU < 0) ? -1.0 : 1.0;
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + sgn * Vs[i][2] * U[j][2];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pose[4 * i + j] = R[i][j];
    pose[4 * i + 3] = tb[i] - (R[i][0] * sb[0] + R[i][1] * sb[1] + R[i][2] * sb[2]);
  }
  pose[12] = pose[13] = pose[14] = 0.0;
  pose[15] = 1.0;
  for (int k = 0; k < 16; ++k)
    if (!isfinite(pose[k])) return OR_NUMERIC;
  return OR_OK;
}

/* ------------------------------------------------------------------------ */
/* iteration controller                                                      */

int oracle_host_update(const oracle_params* p, const double sums[NSUMS], int iteration, double pose[16],
                       double* prev_mse, int* done, int* converged) {
  double np[16];
  int st = oracle_solve_rigid(sums, np);
  if (st != OR_OK) return st;
  double mse = sums[16] / sums[0];
  /* incremental transform D = Pnew * Pold^-1 ; Pold^-1 = [R^T | -R^T t] */
  double Ro[3][3], to[3], Rn[3][3], tn[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) { Ro[i][j] = pose[4 * i + j]; Rn[i][j] = np[4 * i + j]; }
    to[i] = pose[4 * i + 3];
    tn[i] = np[4 * i + 3];
  }
  double trace = 0.0, dt[3];
  for (int i = 0; i < 3; ++i) {
    /* (Rn Ro^T)_ii */
    trace += Rn[i][0] * Ro[i][0] + Rn[i][1] * Ro[i][1] + Rn[i][2] * Ro[i][2];
  }
  for (int i = 0; i < 3; ++i) {
    double acc = 0.0; /* (Rn Ro^T to)_i */
    for (int k = 0; k < 3; ++k) {
      double dik = Rn[i][0] * Ro[k][0] + Rn[i][1] * Ro[k][1] + Rn[i][2] * Ro[k][2];
      acc += dik * to[k];
    }
    dt[i] = tn[i] - acc;
  }
  double one_minus_cos = 1.0 - 0.5 * (trace - 1.0);
  double dt2 = dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2];
  int conv = 0;
  if (p->transformation_epsilon > 0.0 && dt2 <= p->transformation_epsilon && one_minus_cos <= p->rotation_epsilon)
    conv = 1;
  if (p->relative_mse_epsilon > 0.0 && iteration > 0 && fabs(mse - *prev_mse) <= p->relative_mse_epsilon * (*prev_mse))
    conv = 1;
  memcpy(pose, np, sizeof(np));
  *prev_mse = mse;
  *converged = conv;
  *done = conv || (iteration + 1 >= p->max_iterations);
  return OR_OK;
}

/* ------------------------------------------------------------------------ */
/* full ICP.  Optional trace: per iteration NSUMS sums, 16 pose entries (the
 * pose the pass was evaluated under) — arrays sized max_iterations*...       */

int oracle_icp(const float* src_xyz, size_t ns, const float* tgt_xyz, size_t nt, const double init_pose[16],
               const oracle_params* p, int use_kdtree, int nthreads, double out_pose[16], double* out_residual,
               int* out_iters, int* out_converged, double* trace_sums, double* trace_pose, int32_t* last_idx,
               float* last_d2) {
  if (!p || p->max_iterations < 1 || !ns || !nt) return OR_INVALID;
  double pose[16];
  memcpy(pose, init_pose, sizeof(pose));
  float* moved = (float*)malloc(ns * 3 * sizeof(float));
  int32_t* idx = (int32_t*)malloc(ns * sizeof(int32_t));
  float* d2 = (float*)malloc(ns * sizeof(float));
  oracle_kdtree* tree = use_kdtree ? oracle_kdtree_build(tgt_xyz, nt) : NULL;
  double prev_mse = 0.0, residual = 0.0;
  int iters = 0, conv = 0, st = OR_OK;
  for (int it = 0; it < p->max_iterations; ++it) {
    float pf[12];
    double sums[NSUMS];
    oracle_pose_to_f32(pose, pf);
    oracle_transform(pf, src_xyz, ns, moved);
    if (tree) oracle_kdtree_query(tree, moved, ns, idx, d2, nthreads);
    else oracle_nn_brute(moved, ns, tgt_xyz, nt, idx, d2, nthreads);
    oracle_accumulate(src_xyz, tgt_xyz, idx, d2, ns, p->max_correspondence_distance, sums);
    if (trace_sums) memcpy(&trace_sums[(size_t)it * NSUMS], sums, sizeof(sums));
    if (trace_pose) memcpy(&trace_pose[(size_t)it * 16], pose, sizeof(pose));
    int done = 0;
    st = oracle_host_update(p, sums, it, pose, &prev_mse, &done, &conv);
    if (st != OR_OK) break;
    iters = it + 1;
    residual = sqrt(prev_mse);
    if (done) break;
  }
  if (last_idx) memcpy(last_idx, idx, ns * sizeof(int32_t));
  if (last_d2) memcpy(last_d2, d2, ns * sizeof(float));
  memcpy(out_pose, pose, sizeof(pose));
  if (out_residual) *out_residual = residual;
  if (out_iters) *out_iters = iters;
  if (out_converged) *out_converged = conv;
  oracle_kdtree_free(tree);
  free(moved); free(idx); free(d2);
  return st;
}

/* ------------------------------------------------------------------------ */
/* pose packing: chessboard_pose_estimator.cpp:196-207 formula (evaluated in
 * float like the reference's xt::xarray<float>), and the reference packing of
 * opencv_surface_match.cpp:133-142.                                         */

static float quat_axis(float sign_arg, float a, float b, float c) {
  /* 0.5 * copysign(1.0, sign_arg) * sqrt(a - b - c + 1): float arithmetic inside, as on xt::xarray<float>.
   * The argument is clamped at 0: the reference formula returns NaN when rounding drives it just below 0. */
  float arg = ((a - b) - c) + 1.0f;
  if (arg < 0.0f) arg = 0.0f;
  float r = sqrtf(arg);
  return (float)(0.5 * copysign(1.0, (double)sign_arg) * (double)r);
}

void oracle_pose_to_pose7(const double pose[16], float out7[7], int reference_packing) {
  float h[3][4];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) h[i][j] = (float)pose[4 * i + j];
  float ox = quat_axis(h[2][1] - h[1][2], h[0][0], h[1][1], h[2][2]);
  float oy = quat_axis(h[0][2] - h[2][0], h[1][1], h[2][2], h[0][0]);
  float oz = quat_axis(h[1][0] - h[0][1], h[2][2], h[0][0], h[1][1]);
  float warg = ((h[0][0] + h[1][1]) + h[2][2]) + 1.0f;
  if (warg < 0.0f) warg = 0.0f;
  float w = (float)(0.5 * (double)sqrtf(warg));
  out7[0] = h[0][3]; out7[1] = h[1][3]; out7[2] = h[2][3];
  if (reference_packing) {
    /* pose[3]=q[1]; pose[4]=q[2]; pose[6]=q[3]; pose[6]=q[0]; with q = (w,x,y,z): pose[5] never set */
    out7[3] = ox; out7[4] = oy; out7[5] = 0.0f; out7[6] = oz; out7[6] = w;
  } else {
    out7[3] = ox; out7[4] = oy; out7[5] = oz; out7[6] = w;
  }
}

int oracle_omp_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ======================================================================== */
/* "Reference mode": the structure of what the reference really runs through
 * cv::ppf_match_3d::ICP(iterations, tolerance, rejectionScale, numLevels)
 * (pose_estimation/src/opencv_surface_match.cpp:85,94):
 * resolution pyramid, exact 1-NN, median+MAD rejection, "picky" one-to-one
 * dedupe, linearised point-to-plane least squares on the scene normals, and the
 * error-ratio stop rule.  The upstream source is NOT available offline, so this
 * restates the published structure as recalled; every numeric choice upstream
 * leaves to its implementation is pinned here (PARITY UNPINNED, see header):
 *   level l = L-1..0: both clouds subsampled with step 2^l (rows 0, step, 2*step, ...;
 *     floor(n/step) rows), at most round(iterations/(l+1)) passes, stop when
 *     fval/fval_old lies in (1 - tol*(l+1)^2, 1 + tol*(l+1)^2)
 *   pass: pose_cur = PoseX * pose; s' = transform_f32(pose_cur, s); exact NN -> (j, d2)
 *     rejection (scale > 0): med = lower median of d2 (rank (n-1)/2), dev_i = (float)|d2_i - med|
 *       (difference in double), thr = scale*(1.48257968f*median(dev)) + med in float32; keep d2 < thr
 *     picky: a target hit by several kept sources keeps the one with the smallest d2 (lowest source index on ties)
 *     fewer than 6 pairs: leave the level
 *     p = transform_f32(pose, s) (the LEVEL-START point), q, n = target point and normal:
 *       row a = [p x n, n], b = (q - p).n; x = least-squares solution (normal equations in double,
 *       symmetric-eigen pseudo-inverse); PoseX = [Rx(x0) Ry(x1) Rz(x2) | x3..5]
 *     fval = sqrt(sum |p - q|^2) / n_level
 *   after the level: pose = PoseX * pose, residual = smallest fval of the level
 * Deviations of THIS form from upstream (rounds 1-3; kept behind a flag since round 4): no pre-normalisation of the
 * clouds (the residual is in cloud units, and the plane equations are linearised about the cloud origin, not about
 * upstream's joint mean), fval over xyz only, step 2^l whatever the model size, round-half-up of the pass limit.  The
 * UPSTREAM-FAITHFUL variant further down (oracle_icp_reference_upstream) restates those four as well.   */

#define NREF 29 /* 21 upper-triangle A^T A, 6 A^T b, sum |p-q|^2, count */

typedef struct oracle_ref_params {
  int32_t iterations;
  float tolerance;
  float rejection_scale;
  int32_t num_levels;
} oracle_ref_params;

static int cmp_f32(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* element of 0-based rank k in sorted order (test oracle: full sort is fine) */
float oracle_select_rank(const float* v, size_t n, size_t k) {
  float* t = (float*)malloc(n * sizeof(float));
  memcpy(t, v, n * sizeof(float));
  qsort(t, n, sizeof(float), cmp_f32);
  float r = t[k];
  free(t);
  return r;
}

float oracle_rejection_threshold(const float* d2, size_t n, float scale) {
  float med = oracle_select_rank(d2, n, (n - 1) / 2);
  float* dev = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) dev[i] = (float)fabs((double)d2[i] - (double)med);
  float s = 1.48257968f * oracle_select_rank(dev, n, (n - 1) / 2);
  free(dev);
  float thr = scale * s;
  return thr + med;
}

static void jacobi_eig_n(double* A, double* V, int n) { /* symmetric A (n x n, row-major) -> eigenvalues on diag, V columns */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) {
      diag += fabs(A[i * n + i]);
      for (int j = i + 1; j < n; ++j) off += fabs(A[i * n + j]);
    }
    if (off <= 1e-300 || off <= 1e-30 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}

/* x = pinv(N) g for the 6x6 normal equations packed in sums (upper triangle row-major, then g) */
int oracle_solve_point_to_plane(const double sums[NREF], double x[6]) {
  for (int k = 0; k < NREF; ++k)
    if (!isfinite(sums[k])) return OR_NUMERIC;
  double N[36], V[36], g[6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) { N[i * 6 + j] = N[j * 6 + i] = sums[k++]; }
  for (int i = 0; i < 6; ++i) g[i] = sums[21 + i];
  jacobi_eig_n(N, V, 6);
  double wmax = 0;
  for (int i = 0; i < 6; ++i)
    if (fabs(N[i * 6 + i]) > wmax) wmax = fabs(N[i * 6 + i]);
  for (int i = 0; i < 6; ++i) x[i] = 0.0;
  for (int e = 0; e < 6; ++e) {
    double w = N[e * 6 + e];
    if (!(w > 1e-12 * wmax)) continue; /* rank-deficient direction: minimum-norm solution */
    double proj = 0;
    for (int i = 0; i < 6; ++i) proj += V[i * 6 + e] * g[i];
    for (int i = 0; i < 6; ++i) x[i] += V[i * 6 + e] * proj / w;
  }
  for (int i = 0; i < 6; ++i)
    if (!isfinite(x[i])) return OR_NUMERIC;
  return OR_OK;
}

void oracle_euler_pose(const double x[6], double P[16]) { /* R = Rx(x0) * (Ry(x1) * Rz(x2)), t = x3..5 */
  double c0 = cos(x[0]), s0 = sin(x[0]), c1 = cos(x[1]), s1 = sin(x[1]), c2 = cos(x[2]), s2 = sin(x[2]);
  double Rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0};
  double Ry[9] = {c1, 0, s1, 0, 1, 0, -s1, 0, c1};
  double Rz[9] = {c2, -s2, 0, s2, c2, 0, 0, 0, 1};
  double T[9], R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = Ry[3 * i] * Rz[j] + Ry[3 * i + 1] * Rz[3 + j] + Ry[3 * i + 2] * Rz[6 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = Rx[3 * i] * T[j] + Rx[3 * i + 1] * T[3 + j] + Rx[3 * i + 2] * T[6 + j];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) P[4 * i + j] = R[3 * i + j];
    P[4 * i + 3] = x[3 + i];
  }
  P[12] = P[13] = P[14] = 0;
  P[15] = 1;
}

static void mat4_mul(const double A[16], const double B[16], double C[16]) {
  double T[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0;
      for (int k = 0; k < 4; ++k) acc += A[4 * i + k] * B[4 * k + j];
      T[4 * i + j] = acc;
    }
  memcpy(C, T, sizeof(T));
}

/* one reference-mode pass on already subsampled clouds; returns the number of selected pairs.
 * src_xyz: level source (ns x 3), tgt6: level target (nt x 6, xyz + normal).  Outputs optional. */
size_t oracle_ref_pass(const float* src_xyz, size_t ns, const float* tgt6, size_t nt, const double pose_start[16],
                       const double pose_cur[16], float rejection_scale, int nthreads, double sums[NREF],
                       int32_t* out_idx, float* out_d2, uint8_t* out_selected, float* out_threshold) {
  float *tgt_xyz = (float*)malloc(nt * 3 * sizeof(float)), *moved = (float*)malloc(ns * 3 * sizeof(float));
  float* start = (float*)malloc(ns * 3 * sizeof(float));
  int32_t* idx = (int32_t*)malloc(ns * sizeof(int32_t));
  float* d2 = (float*)malloc(ns * sizeof(float));
  for (size_t j = 0; j < nt; ++j) memcpy(&tgt_xyz[3 * j], &tgt6[6 * j], 3 * sizeof(float));
  float pf[12];
  oracle_pose_to_f32(pose_cur, pf);
  oracle_transform(pf, src_xyz, ns, moved);
  oracle_pose_to_f32(pose_start, pf);
  oracle_transform(pf, src_xyz, ns, start);
  oracle_kdtree* tree = oracle_kdtree_build(tgt_xyz, nt);
  oracle_kdtree_query(tree, moved, ns, idx, d2, nthreads);
  oracle_kdtree_free(tree);
  float thr = INFINITY;
  if (rejection_scale > 0.0f) thr = oracle_rejection_threshold(d2, ns, rejection_scale);
  /* picky: per target the kept source with the smallest d2, lowest index on ties */
  int32_t* owner = (int32_t*)malloc(nt * sizeof(int32_t));
  for (size_t j = 0; j < nt; ++j) owner[j] = -1;
  for (size_t i = 0; i < ns; ++i) {
    if (!(d2[i] < thr)) continue;
    int32_t j = idx[i];
    if (owner[j] < 0 || d2[i] < d2[owner[j]]) owner[j] = (int32_t)i; /* ascending i: ties keep the lower index */
  }
  for (int k = 0; k < NREF; ++k) sums[k] = 0.0;
  size_t nsel = 0;
  for (size_t i = 0; i < ns; ++i) {
    int sel = (d2[i] < thr) && owner[idx[i]] == (int32_t)i;
    if (out_selected) out_selected[i] = (uint8_t)sel;
    if (!sel) continue;
    ++nsel;
    const float* p = &start[3 * i];
    const float* q = &tgt6[6 * (size_t)idx[i]];
    double P[3] = {p[0], p[1], p[2]}, Q[3] = {q[0], q[1], q[2]}, Nn[3] = {q[3], q[4], q[5]};
    double a[6] = {P[1] * Nn[2] - P[2] * Nn[1], P[2] * Nn[0] - P[0] * Nn[2], P[0] * Nn[1] - P[1] * Nn[0], Nn[0], Nn[1], Nn[2]};
    double dx = Q[0] - P[0], dy = Q[1] - P[1], dz = Q[2] - P[2];
    double b = dx * Nn[0] + dy * Nn[1] + dz * Nn[2];
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) sums[k++] += a[r] * a[c];
    for (int r = 0; r < 6; ++r) sums[21 + r] += a[r] * b;
    sums[27] += dx * dx + dy * dy + dz * dz;
    sums[28] += 1.0;
  }
  if (out_idx) memcpy(out_idx, idx, ns * sizeof(int32_t));
  if (out_d2) memcpy(out_d2, d2, ns * sizeof(float));
  if (out_threshold) *out_threshold = thr;
  free(tgt_xyz); free(moved); free(start); free(idx); free(d2); free(owner);
  return nsel;
}

/* host controller of one level (shared shape with the product's): returns 1 when the level loop must stop */
int oracle_icp_reference(const float* src_xyz, size_t ns, const float* tgt6, size_t nt, const double init_pose[16],
                         const oracle_ref_params* rp, int nthreads, double out_pose[16], double* out_residual,
                         int* out_passes) {
  if (!rp || rp->iterations < 1 || rp->num_levels < 1 || !ns || !nt) return OR_INVALID;
  double pose[16];
  memcpy(pose, init_pose, sizeof(pose));
  double residual = 0.0;
  int passes = 0;
  for (int level = rp->num_levels - 1; level >= 0; --level) {
    const size_t step = (size_t)1 << level;
    const size_t ns_l = ns / step, nt_l = nt / step;
    if (ns_l < 6 || nt_l < 1) continue;
    float* s_l = (float*)malloc(ns_l * 3 * sizeof(float));
    float* t_l = (float*)malloc(nt_l * 6 * sizeof(float));
    for (size_t i = 0; i < ns_l; ++i) memcpy(&s_l[3 * i], &src_xyz[3 * i * step], 3 * sizeof(float));
    for (size_t j = 0; j < nt_l; ++j) memcpy(&t_l[6 * j], &tgt6[6 * j * step], 6 * sizeof(float));
    const double tolp = (double)rp->tolerance * (double)(level + 1) * (double)(level + 1);
    const int maxit = (int)floor((double)rp->iterations / (double)(level + 1) + 0.5);
    double posex[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double fval_old = 9999999999.0, fval_perc = 0.0, fval_min = 9999999999.0;
    int i = 0;
    while (!(fval_perc < 1.0 + tolp && fval_perc > 1.0 - tolp) && i < maxit) {
      double cur[16], sums[NREF], x[6];
      mat4_mul(posex, pose, cur);
      size_t nsel = oracle_ref_pass(s_l, ns_l, t_l, nt_l, pose, cur, rp->rejection_scale, nthreads, sums, NULL, NULL, NULL, NULL);
      ++passes;
      if (nsel < 6) break;
      if (oracle_solve_point_to_plane(sums, x) != OR_OK) break;
      oracle_euler_pose(x, posex);
      double fval = sqrt(sums[27]) / (double)ns_l;
      fval_perc = fval / fval_old;
      fval_old = fval;
      if (fval < fval_min) fval_min = fval;
      ++i;
    }
    mat4_mul(posex, pose, pose);
    residual = fval_min;
    free(s_l); free(t_l);
  }
  memcpy(out_pose, pose, sizeof(pose));
  if (out_residual) *out_residual = residual;
  if (out_passes) *out_passes = passes;
  return OR_OK;
}

/* ======================================================================== */
/* Reference mode, UPSTREAM-FAITHFUL variant (round 4; MI355_ICP_FLAG_UPSTREAM in the product).
 *
 * The form above leaves out four things upstream's cv::ppf_match_3d::ICP::registerModelToScene does that change its
 * CONTROL FLOW (how many passes a level runs, hence the pose and the residual the reference's best-pose rule reads at
 * opencv_surface_match.cpp:114-118).  This variant restates them.  Everything below is RECALLED from upstream's
 * published source (opencv_contrib modules/surface_matching/src/icp.cpp, c_utils.hpp, ppf_helpers.cpp); that source is
 * absent from /root/reference and from this image and cannot be checked offline - PARITY UNPINNED, as above.
 *
 *  (a) the model is N x 6 (opencv_surface_match.cpp:22): transformPCPose rotates and re-normalises its normals with the
 *      cloud, and the matched rows carry them;
 *  (b) fval = cv::norm(Src_Match, Dst_Match) / rows(level source): the L2 norm over ALL SIX columns of the selected
 *      (level-start model row, scene row) pairs - position differences in the normalised units of (c) and the difference
 *      of the rotated model normal and the scene normal;
 *  (c) before the level loop: meanAvg = (mean(model') + mean(scene)) / 2 with model' = the model under the candidate pose;
 *      both clouds minus (float)meanAvg; scale = n / ((sum |model'_i| + sum |scene_j|) / 2) over the centred clouds
 *      (computeDistToOrigin returns the SUM of the distances, its comment notwithstanding), xyz columns times scale;
 *      the level loop runs from the identity on those clouds; afterwards t <- t / scale + meanAvg - R meanAvg, and the
 *      wrapper appends that pose to the candidate's (Pose3D::appendPose);
 *  (d) level step: numSamples = cvRound(n / 2^level), sampleStep = cvRound(n / numSamples), n the MODEL's row count, used
 *      for BOTH clouds (samplePCUniform: rows 0, step, 2 step, ..., rows / step of them); cvRound rounds half to even
 *      (also in MaxIterationsPyr = cvRound(iterations / (level + 1)): 250 / 4 -> 62, where the form above takes 63).
 *
 * How (c) is restated: ALGEBRAICALLY.  Materialising upstream's clouds would mean one centred and scaled copy of the SCENE
 * per candidate pose (meanAvg and scale depend on the candidate).  Instead the clouds stay as they are and the similarity
 * x' = scale (x - c), c = (float)meanAvg, is carried through the formulas - exact in real arithmetic:
 *   - nearest neighbours, the MAD threshold test and the picky winners are invariant under a common similarity;
 *   - the plane equations are written for p - c and q - c (row [(p - c) x n, n], rhs (q - p).n): the linearisation of the
 *     rotation is about upstream's origin, which is what makes the iterates differ from a solve about the camera origin
 *     (second order in the rotation, times |c|); upstream's unknowns are (rpy, scale t), so x[3..5] here is its t / scale;
 *   - poses live in the centred frame (PoseX_c, pose_c); the pose a pass is evaluated under in cloud coordinates is
 *     Tr(c) X_c Tr(-c) T0 with T0 the candidate pose, which is also what is returned (upstream: appendPose of the
 *     un-normalised pose);
 *   - fval = sqrt(scale^2 sum |p - q|^2 + sum |n_p - n_q|^2) / n_level, the residual is upstream's number.
 * What this cannot reproduce: the float32 roundings of upstream's intermediate clouds (it rounds the
 * candidate-transformed, centred, scaled and level-start clouds to float32 one after the other; here the original float32
 * model goes through ONE float32 transform by the composed pose, the contract of the header) - differences of a few
 * 1e-7 relative in coordinates, which move a nearest neighbour or a threshold decision only at near-ties; FLANN's tie
 * order; picky's indexing quirk (above).  tests/upstream_literal.py materialises upstream's clouds step by step in numpy
 * and is compared with this restatement (tests/test_upstream_variant_cpu.py).
 *
 * Numeric definitions pinned here (the GPU path reproduces them):
 *   mean(model') = R0 mean(model) + t0 in double (means: sequential double sums of the float32 coordinates / rows)
 *   c = (float)(0.5 (mean(model') + mean(scene)))
 *   |x - c| = sqrtf(((r0 r0) + (r1 r1)) + (r2 r2)), r = x - c in float32 (upstream's float expression), summed in double,
 *     model' points = the float32 transform of the header under the candidate pose
 *   rotated model normal: (R n) / |R n| with R the float32 rounding of the level-start pose in cloud coordinates, products
 *     and sums in double (left to right), rounded to float32 like upstream's convertTo; zero if |R n| <= FLT_EPSILON     */

#define NREFX 30 /* the 29 above + sum |n_p - n_q|^2 */

/* cvRound: upstream rounds through cvtsd2si / lrint under the default rounding mode - half to even */
static long cv_round(double v) { return lrint(v); }

/* sampling step of pyramid level `level` for a model of n_model rows; 0: the level has no samples (n / 2^level rounds to 0;
 * upstream would divide by zero there) */
uint32_t oracle_upstream_step(size_t n_model, int level) {
  const double div = pow(2.0, (double)level);
  const long num = cv_round((double)n_model / div);
  if (num < 1) return 0;
  const long step = cv_round((double)n_model / (double)num);
  return step < 1 ? 1u : (uint32_t)step;
}

int oracle_upstream_max_iterations(int iterations, int level) { return (int)cv_round((double)iterations / (double)(level + 1)); }

/* pose in cloud coordinates of the centred-frame pose Xc: Tr(c) Xc Tr(-c) T0 */
void oracle_from_centred(const double Xc[16], const float c[3], const double T0[16], double out[16]) {
  double M[16];
  memcpy(M, Xc, sizeof(M));
  for (int i = 0; i < 3; ++i) {
    double rc = Xc[4 * i] * (double)c[0] + Xc[4 * i + 1] * (double)c[1] + Xc[4 * i + 2] * (double)c[2];
    M[4 * i + 3] = (Xc[4 * i + 3] + (double)c[i]) - rc;
  }
  mat4_mul(M, T0, out);
}

static double dist_to_centre(const float p[3], const float c[3]) {
  float r0 = p[0] - c[0], r1 = p[1] - c[1], r2 = p[2] - c[2];
  float a = r0 * r0, b = r1 * r1, e = r2 * r2;
  a = a + b;
  a = a + e;
  return (double)sqrtf(a);
}

/* centre (float32) and scale of upstream's pre-normalisation for model src (rows of src_stride floats) under init_pose */
void oracle_upstream_normalisation(const float* src, size_t ns, size_t src_stride, const float* tgt, size_t nt, size_t tgt_stride,
                                   const double init_pose[16], float centre[3], double* scale) {
  double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
  for (size_t i = 0; i < ns; ++i)
    for (int a = 0; a < 3; ++a) ms[a] += (double)src[i * src_stride + a];
  for (size_t j = 0; j < nt; ++j)
    for (int a = 0; a < 3; ++a) mt[a] += (double)tgt[j * tgt_stride + a];
  for (int a = 0; a < 3; ++a) { ms[a] /= (double)ns; mt[a] /= (double)nt; }
  for (int a = 0; a < 3; ++a) {
    double m = ((init_pose[4 * a] * ms[0] + init_pose[4 * a + 1] * ms[1]) + init_pose[4 * a + 2] * ms[2]) + init_pose[4 * a + 3];
    centre[a] = (float)(0.5 * (m + mt[a]));
  }
  float pf[12];
  oracle_pose_to_f32(init_pose, pf);
  double ds = 0.0, dt = 0.0;
  for (size_t i = 0; i < ns; ++i) {
    float p[3];
    xform(pf, src[i * src_stride], src[i * src_stride + 1], src[i * src_stride + 2], &p[0], &p[1], &p[2]);
    ds += dist_to_centre(p, centre);
  }
  for (size_t j = 0; j < nt; ++j) dt += dist_to_centre(&tgt[j * tgt_stride], centre);
  *scale = (double)ns / ((ds + dt) * 0.5);
}

/* One pass on level clouds, both variants.  src: ns rows of src_stride floats (xyz, or xyz + normal when centre != NULL);
 * centre == NULL: exactly oracle_ref_pass, sums[29] = 0.  centre != NULL: plane equations about centre, sums[29] = sum over
 * the selected pairs of |n_p - n_q|^2 (n_p: the model normal under the level-start pose, as defined above). */
size_t oracle_ref_pass_ex(const float* src, size_t ns, size_t src_stride, const float* tgt6, size_t nt, const double pose_start[16],
                          const double pose_cur[16], float rejection_scale, const float* centre, int nthreads, double sums[NREFX],
                          int32_t* out_idx, float* out_d2, uint8_t* out_selected, float* out_threshold) {
  float *tgt_xyz = (float*)malloc(nt * 3 * sizeof(float)), *moved = (float*)malloc(ns * 3 * sizeof(float));
  float* start = (float*)malloc(ns * 3 * sizeof(float));
  float* sxyz = (float*)malloc(ns * 3 * sizeof(float));
  int32_t* idx = (int32_t*)malloc(ns * sizeof(int32_t));
  float* d2 = (float*)malloc(ns * sizeof(float));
  for (size_t j = 0; j < nt; ++j) memcpy(&tgt_xyz[3 * j], &tgt6[6 * j], 3 * sizeof(float));
  for (size_t i = 0; i < ns; ++i) memcpy(&sxyz[3 * i], &src[i * src_stride], 3 * sizeof(float));
  float pf[12], ps[12];
  oracle_pose_to_f32(pose_cur, pf);
  oracle_transform(pf, sxyz, ns, moved);
  oracle_pose_to_f32(pose_start, ps);
  oracle_transform(ps, sxyz, ns, start);
  oracle_kdtree* tree = oracle_kdtree_build(tgt_xyz, nt);
  oracle_kdtree_query(tree, moved, ns, idx, d2, nthreads);
  oracle_kdtree_free(tree);
  float thr = INFINITY;
  if (rejection_scale > 0.0f) thr = oracle_rejection_threshold(d2, ns, rejection_scale);
  int32_t* owner = (int32_t*)malloc(nt * sizeof(int32_t));
  for (size_t j = 0; j < nt; ++j) owner[j] = -1;
  for (size_t i = 0; i < ns; ++i) {
    if (!(d2[i] < thr)) continue;
    int32_t j = idx[i];
    if (owner[j] < 0 || d2[i] < d2[owner[j]]) owner[j] = (int32_t)i;
  }
  for (int k = 0; k < NREFX; ++k) sums[k] = 0.0;
  const double C[3] = {centre ? (double)centre[0] : 0.0, centre ? (double)centre[1] : 0.0, centre ? (double)centre[2] : 0.0};
  size_t nsel = 0;
  for (size_t i = 0; i < ns; ++i) {
    int sel = (d2[i] < thr) && owner[idx[i]] == (int32_t)i;
    if (out_selected) out_selected[i] = (uint8_t)sel;
    if (!sel) continue;
    ++nsel;
    const float* p = &start[3 * i];
    const float* q = &tgt6[6 * (size_t)idx[i]];
    double P[3] = {p[0], p[1], p[2]}, Q[3] = {q[0], q[1], q[2]}, Nn[3] = {q[3], q[4], q[5]};
    double Pc[3] = {P[0] - C[0], P[1] - C[1], P[2] - C[2]};
    double a[6] = {Pc[1] * Nn[2] - Pc[2] * Nn[1], Pc[2] * Nn[0] - Pc[0] * Nn[2], Pc[0] * Nn[1] - Pc[1] * Nn[0], Nn[0], Nn[1], Nn[2]};
    double dx = Q[0] - P[0], dy = Q[1] - P[1], dz = Q[2] - P[2];
    double b = dx * Nn[0] + dy * Nn[1] + dz * Nn[2];
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) sums[k++] += a[r] * a[c];
    for (int r = 0; r < 6; ++r) sums[21 + r] += a[r] * b;
    sums[27] += dx * dx + dy * dy + dz * dz;
    sums[28] += 1.0;
    if (centre) {
      const float* sn = &src[i * src_stride + 3];
      double n2[3], np32[3];
      for (int r = 0; r < 3; ++r)
        n2[r] = ((double)ps[4 * r] * (double)sn[0] + (double)ps[4 * r + 1] * (double)sn[1]) + (double)ps[4 * r + 2] * (double)sn[2];
      const double nn = sqrt((n2[0] * n2[0] + n2[1] * n2[1]) + n2[2] * n2[2]);
      for (int r = 0; r < 3; ++r) np32[r] = nn > 1.192092896e-07 ? (double)(float)(n2[r] / nn) : 0.0;
      const double ex = np32[0] - Nn[0], ey = np32[1] - Nn[1], ez = np32[2] - Nn[2];
      sums[29] += (ex * ex + ey * ey) + ez * ez;
    }
  }
  if (out_idx) memcpy(out_idx, idx, ns * sizeof(int32_t));
  if (out_d2) memcpy(out_d2, d2, ns * sizeof(float));
  if (out_threshold) *out_threshold = thr;
  free(tgt_xyz); free(moved); free(start); free(sxyz); free(idx); free(d2); free(owner);
  return nsel;
}

/* src6 / tgt6: N x 6 rows (xyz + normal).  out_centre (3 floats), out_scale, out_level_passes (num_levels ints, passes run on
 * each level, index = level) may be NULL. */
int oracle_icp_reference_upstream(const float* src6, size_t ns, const float* tgt6, size_t nt, const double init_pose[16],
                                  const oracle_ref_params* rp, int nthreads, double out_pose[16], double* out_residual,
                                  int* out_passes, float* out_centre, double* out_scale, int* out_level_passes) {
  if (!rp || rp->iterations < 1 || rp->num_levels < 1 || !ns || !nt) return OR_INVALID;
  float centre[3];
  double scale;
  oracle_upstream_normalisation(src6, ns, 6, tgt6, nt, 6, init_pose, centre, &scale);
  if (out_centre) memcpy(out_centre, centre, sizeof(centre));
  if (out_scale) *out_scale = scale;
  double pose_c[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double residual = 0.0;
  int passes = 0;
  for (int level = rp->num_levels - 1; level >= 0; --level) {
    if (out_level_passes) out_level_passes[level] = 0;
    const size_t step = oracle_upstream_step(ns, level);
    if (!step) continue;
    const size_t ns_l = ns / step, nt_l = nt / step;
    if (ns_l < 6 || nt_l < 1) continue;
    float* s_l = (float*)malloc(ns_l * 6 * sizeof(float));
    float* t_l = (float*)malloc(nt_l * 6 * sizeof(float));
    for (size_t i = 0; i < ns_l; ++i) memcpy(&s_l[6 * i], &src6[6 * i * step], 6 * sizeof(float));
    for (size_t j = 0; j < nt_l; ++j) memcpy(&t_l[6 * j], &tgt6[6 * j * step], 6 * sizeof(float));
    const double tolp = (double)rp->tolerance * (double)(level + 1) * (double)(level + 1);
    const int maxit = oracle_upstream_max_iterations(rp->iterations, level);
    double posex[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double fval_old = 9999999999.0, fval_perc = 0.0, fval_min = 9999999999.0;
    int i = 0;
    while (!(fval_perc < 1.0 + tolp && fval_perc > 1.0 - tolp) && i < maxit) {
      double cur_c[16], start_o[16], cur_o[16], sums[NREFX], x[6];
      mat4_mul(posex, pose_c, cur_c);
      oracle_from_centred(pose_c, centre, init_pose, start_o);
      oracle_from_centred(cur_c, centre, init_pose, cur_o);
      size_t nsel = oracle_ref_pass_ex(s_l, ns_l, 6, t_l, nt_l, start_o, cur_o, rp->rejection_scale, centre, nthreads, sums, NULL, NULL, NULL, NULL);
      ++passes;
      if (out_level_passes) out_level_passes[level] += 1;
      if (nsel < 6) break;
      if (oracle_solve_point_to_plane(sums, x) != OR_OK) break;
      oracle_euler_pose(x, posex);
      double fval = sqrt(scale * scale * sums[27] + sums[29]) / (double)ns_l;
      fval_perc = fval / fval_old;
      fval_old = fval;
      if (fval < fval_min) fval_min = fval;
      ++i;
    }
    mat4_mul(posex, pose_c, pose_c);
    residual = fval_min;
    free(s_l); free(t_l);
  }
  oracle_from_centred(pose_c, centre, init_pose, out_pose);
  if (out_residual) *out_residual = residual;
  if (out_passes) *out_passes = passes;
  return OR_OK;
}

/* ======================================================================== */
/* Scene preparation filters, restated from the reference's OWN code
 * (pose_estimation/src/pose_estimation.cpp):
 *   kind 0  pcl::removeNaNFromPointCloud, :248  - keep points with finite x, y, z
 *   kind 1/2 filter_points, :347-372            - float dx = point[0] - x ...; float d = std::sqrt(dx*dx + dy*dy + dz*dz);
 *                                                 d <= filter_radius marks the point; ExtractIndices keeps the marked
 *                                                 points (kind 1) or the others (kind 2, setNegative(true))
 *   kind 3  remove_planes, :310-333             - float d = (x*a + y*b + z*c + dd) / std::sqrt(x*x + y*y + z*z);
 *                                                 std::abs(d) <= 0.005 (double literal) marks the point; marked points
 *                                                 are removed
 * Output order: ascending original index (the reference's parallel loops are serialised by `omp critical` and push
 * indices in arrival order; one thread gives this order).  The reference has no test for these lines either.      */
typedef struct oracle_scene_filter {
  int32_t kind;
  float p[5];
} oracle_scene_filter;

size_t oracle_scene_filter_apply(const float* xyz, size_t n, const oracle_scene_filter* f, int nf, float* out) {
  size_t kept = 0;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    int keep = 1;
    for (int k = 0; k < nf && keep; ++k) {
      if (f[k].kind == 0) {
        keep = isfinite(x) && isfinite(y) && isfinite(z);
      } else if (f[k].kind == 1 || f[k].kind == 2) {
        float dx = f[k].p[0] - x, dy = f[k].p[1] - y, dz = f[k].p[2] - z;
        float d = sqrtf(dx * dx + dy * dy + dz * dz);
        int inside = d <= f[k].p[3];
        keep = f[k].kind == 1 ? inside : !inside;
      } else {
        float d = (x * f[k].p[0] + y * f[k].p[1] + z * f[k].p[2] + f[k].p[3]) / sqrtf(x * x + y * y + z * z);
        keep = !((double)fabsf(d) <= (double)f[k].p[4]);
      }
    }
    if (keep) {
      out[3 * kept] = x; out[3 * kept + 1] = y; out[3 * kept + 2] = z;
      ++kept;
    }
  }
  return kept;
}

/* ======================================================================== */
/* Scene normals: the step right before the ICP in the reference,
 *   cv::ppf_match_3d::computeNormalsPC3d(pc_scene, pc_scene_normals, 20, true, viewpoint)
 *   (pose_estimation/src/opencv_surface_match.cpp:57-59; third-party, source absent: SURVEY.md 8a R4).
 * Restated from upstream's published structure as recalled (PARITY UNPINNED): for every point the k nearest
 * points of the same cloud (the point itself included; exact float32 d2 as in dist2; ordered by (d2, index)),
 * covariance of those k points accumulated as upstream's meanCovLocalPCInd does - float32 products cloud[j]*cloud[k]
 * added into doubles, divided by k, minus mean*mean^T - the eigenvector of the smallest eigenvalue (cyclic Jacobi
 * here; upstream calls cv::eigen), cast to float32, and flipped so that (viewpoint - point) . normal >= 0
 * (flipNormalViewpoint).  k is clamped to the number of points.                                                    */

typedef struct knn_heap { int k, n; float* d; int32_t* i; } knn_heap;

static inline void knn_offer(knn_heap* h, float d, int32_t j) {
  /* list sorted ascending by (d, index) */
  if (h->n == h->k) {
    float wd = h->d[h->k - 1];
    int32_t wi = h->i[h->k - 1];
    if (!(d < wd || (d == wd && j < wi))) return;
  }
  int pos = h->n < h->k ? h->n++ : h->k - 1;
  while (pos > 0 && (d < h->d[pos - 1] || (d == h->d[pos - 1] && j < h->i[pos - 1]))) {
    h->d[pos] = h->d[pos - 1];
    h->i[pos] = h->i[pos - 1];
    --pos;
  }
  h->d[pos] = d;
  h->i[pos] = j;
}

static void kd_knn_one(const oracle_kdtree* t, float qx, float qy, float qz, knn_heap* h) {
  int32_t stack[128];
  int sp = 0;
  h->n = 0;
  if (t->n) stack[sp++] = 0;
  while (sp) {
    const kd_node* nd = &t->nodes[stack[--sp]];
    if (h->n == h->k && box_lb(nd, qx, qy, qz) > h->d[h->k - 1]) continue;
    if (nd->left < 0) {
      for (int32_t i = nd->begin; i < nd->end; ++i)
        knn_offer(h, dist2(qx, qy, qz, t->pts[3 * (size_t)i], t->pts[3 * (size_t)i + 1], t->pts[3 * (size_t)i + 2]), t->perm[i]);
    } else {
      float ll = box_lb(&t->nodes[nd->left], qx, qy, qz), lr = box_lb(&t->nodes[nd->right], qx, qy, qz);
      if (ll <= lr) { stack[sp++] = nd->right; stack[sp++] = nd->left; }
      else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
    }
  }
}

/* exact k-NN of every point of xyz within xyz: idx/d2 are n x k (k already clamped by the caller), (d2, index) order */
void oracle_knn_self(const float* xyz, size_t n, int k, int32_t* idx, float* d2, int use_kdtree, int nthreads) {
  oracle_kdtree* tree = use_kdtree ? oracle_kdtree_build(xyz, n) : NULL;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
#endif
  for (long long q = 0; q < (long long)n; ++q) {
    knn_heap h = {k, 0, &d2[(size_t)q * k], &idx[(size_t)q * k]};
    float qx = xyz[3 * q], qy = xyz[3 * q + 1], qz = xyz[3 * q + 2];
    if (tree) kd_knn_one(tree, qx, qy, qz, &h);
    else
      for (size_t j = 0; j < n; ++j) knn_offer(&h, dist2(qx, qy, qz, xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]), (int32_t)j);
  }
  (void)nthreads;
  oracle_kdtree_free(tree);
}

/* smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi, fixed sweep order); v normalised */
void oracle_smallest_eigvec3(const double C[9], double v[3]) {
  double A[3][3], V[3][3], w[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = C[3 * i + j];
  jacobi_eig3(A, V, w);
  int m = 0;
  if (w[1] < w[m]) m = 1;
  if (w[2] < w[m]) m = 2;
  double nrm = sqrt(V[0][m] * V[0][m] + V[1][m] * V[1][m] + V[2][m] * V[2][m]);
  for (int i = 0; i < 3; ++i) v[i] = V[i][m] / nrm;
}

void oracle_normals(const float* xyz, size_t n, int k, int flip, const float viewpoint[3], float* out_normals,
                    int32_t* out_idx /* n x k or NULL */, int use_kdtree, int nthreads) {
  if (!n) return;
  if ((size_t)k > n) k = (int)n;
  int32_t* idx = out_idx ? out_idx : (int32_t*)malloc(n * (size_t)k * sizeof(int32_t));
  float* d2 = (float*)malloc(n * (size_t)k * sizeof(float));
  oracle_knn_self(xyz, n, k, idx, d2, use_kdtree, nthreads);
  for (size_t q = 0; q < n; ++q) {
    double cov[9] = {0}, mean[3] = {0};
    for (int m = 0; m < k; ++m) {
      const float* c = &xyz[3 * (size_t)idx[q * k + m]];
      for (int j = 0; j < 3; ++j) {
        for (int l = 0; l < 3; ++l) cov[3 * j + l] += (double)(c[j] * c[l]); /* float32 product, as upstream */
        mean[j] += (double)c[j];
      }
    }
    const double inv = 1.0 / (double)k;
    for (int j = 0; j < 3; ++j) mean[j] *= inv;
    for (int j = 0; j < 9; ++j) cov[j] *= inv;
    for (int j = 0; j < 3; ++j)
      for (int l = 0; l < 3; ++l) cov[3 * j + l] -= mean[j] * mean[l];
    double v[3];
    oracle_smallest_eigvec3(cov, v);
    float nr[3] = {(float)v[0], (float)v[1], (float)v[2]};
    if (flip) {
      float vx = viewpoint[0] - xyz[3 * q], vy = viewpoint[1] - xyz[3 * q + 1], vz = viewpoint[2] - xyz[3 * q + 2];
      float c = vx * nr[0] + vy * nr[1] + vz * nr[2];
      if (c < 0) { nr[0] = -nr[0]; nr[1] = -nr[1]; nr[2] = -nr[2]; }
    }
    out_normals[3 * q] = nr[0]; out_normals[3 * q + 1] = nr[1]; out_normals[3 * q + 2] = nr[2];
  }
  if (!out_idx) free(idx);
  free(d2);
}

/* ======================================================================== */
/* RANSAC plane fit of the scene: what the reference asks PCL for at pose_estimation/src/pose_estimation.cpp:285-297
 *   (pcl::SACSegmentation: SACMODEL_PLANE, SAC_RANSAC, setOptimizeCoefficients(true), setDistanceThreshold(0.0001),
 *   setMaxIterations(100)); third-party, source absent from /root/reference, and its random sample sequence is not
 *   reproducible.  PARITY UNPINNED: this restates PCL's published structure (RandomSampleConsensus::computeModel's
 *   adaptive loop, the plane model's three-point fit and covariance refinement) with every implementation-defined
 *   choice pinned here - the generator (the counter-based one of the synthetic clouds), float32 hypothesis and distance
 *   arithmetic with every product and sum rounded separately, double moments - and is what
 *   mi355_icp_scene_fit_plane (include/mi355_icp.h) is checked against. */
static uint64_t pf_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static uint32_t pf_draw(uint64_t seed, uint64_t ctr, uint32_t n) {
  uint64_t x = pf_mix64(seed + ctr * 0x9E3779B97F4A7C15ULL);
  double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
  uint32_t i = (uint32_t)(u * (double)n);
  return i < n ? i : n - 1;
}
static float pf_distance(const float P[4], const float* p) {
  float d = P[0] * p[0];
  float t = P[1] * p[1], u = P[2] * p[2];
  d = d + t;
  d = d + u;
  d = d + P[3];
  return d;
}
static size_t pf_support(const float* xyz, size_t n, const float P[4], float thr) {
  size_t c = 0;
  for (size_t i = 0; i < n; ++i)
    if (fabsf(pf_distance(P, &xyz[3 * i])) <= thr) ++c; /* NaN: false */
  return c;
}

/* out_planes (max_iterations x 4) and out_counts (max_iterations) may be NULL.  Returns 0, or 1 if no plane was found. */
int oracle_plane_fit(const float* xyz, size_t n, float thr, int max_iterations, double probability, uint64_t seed,
                     int optimize, float out4[4], size_t* out_inliers, int* out_iterations, float* out_planes,
                     uint32_t* out_counts) {
  int best = -1, it = 0;
  size_t best_count = 0;
  double k = 1.0;
  const double log_probability = log(1.0 - probability), eps = 2.220446049250313e-16;
  float bestP[4] = {0, 0, 0, 0};
  for (int h = 0; h < max_iterations; ++h) {
    float P[4] = {0.f, 0.f, 0.f, NAN};
    for (uint32_t attempt = 0; attempt < 16; ++attempt) {
      const uint64_t base = ((uint64_t)h * 16u + attempt) * 3u;
      const uint32_t i0 = pf_draw(seed, base, (uint32_t)n), i1 = pf_draw(seed, base + 1, (uint32_t)n), i2 = pf_draw(seed, base + 2, (uint32_t)n);
      if (i0 == i1 || i0 == i2 || i1 == i2) continue;
      const float* p0 = &xyz[3 * (size_t)i0];
      const float ax = xyz[3 * (size_t)i1] - p0[0], ay = xyz[3 * (size_t)i1 + 1] - p0[1], az = xyz[3 * (size_t)i1 + 2] - p0[2];
      const float bx = xyz[3 * (size_t)i2] - p0[0], by = xyz[3 * (size_t)i2 + 1] - p0[1], bz = xyz[3 * (size_t)i2 + 2] - p0[2];
      float t, u;
      t = ay * bz; u = az * by; const float nx = t - u;
      t = az * bx; u = ax * bz; const float ny = t - u;
      t = ax * by; u = ay * bx; const float nz = t - u;
      float l2 = nx * nx;
      t = ny * ny; l2 = l2 + t;
      t = nz * nz; l2 = l2 + t;
      const float len = sqrtf(l2);
      if (!(len > 0.0f) || !isfinite(len)) continue;
      const float a = nx / len, b = ny / len, c = nz / len;
      float d = a * p0[0];
      t = b * p0[1]; d = d + t;
      t = c * p0[2]; d = d + t;
      P[0] = a; P[1] = b; P[2] = c; P[3] = -d;
      break;
    }
    const int valid = !(P[3] != P[3]);
    const size_t cnt = valid ? pf_support(xyz, n, P, thr) : 0;
    if (out_planes) memcpy(out_planes + 4 * (size_t)h, P, sizeof(P));
    if (out_counts) out_counts[h] = (uint32_t)cnt;
    /* the adaptive loop, in hypothesis order (all hypotheses are evaluated for the record; the choice only looks at a
     * hypothesis while iterations < k, exactly as a sequential run would) */
    if (!((double)it < k)) continue;
    if (!valid) continue;
    if (best < 0 || cnt > best_count) {
      best = h;
      best_count = cnt;
      memcpy(bestP, P, sizeof(P));
      const double w = (double)best_count / (double)n;
      double p_no = 1.0 - w * w * w;
      p_no = fmax(eps, p_no);
      p_no = fmin(1.0 - eps, p_no);
      k = log_probability / log(p_no);
    }
    ++it;
  }
  if (out_iterations) *out_iterations = it;
  if (best < 0) return 1;
  size_t inliers = best_count;
  if (optimize) {
    double m[10] = {0};
    for (size_t i = 0; i < n; ++i) {
      const float* p = &xyz[3 * i];
      if (fabsf(pf_distance(bestP, p)) <= thr) {
        const double X = p[0], Y = p[1], Z = p[2];
        m[0] += 1.0; m[1] += X; m[2] += Y; m[3] += Z;
        m[4] += X * X; m[5] += X * Y; m[6] += X * Z; m[7] += Y * Y; m[8] += Y * Z; m[9] += Z * Z;
      }
    }
    if (m[0] >= 3.0) {
      const double inv = 1.0 / m[0];
      const double c[3] = {m[1] * inv, m[2] * inv, m[3] * inv};
      double A[3][3] = {{m[4] * inv - c[0] * c[0], m[5] * inv - c[0] * c[1], m[6] * inv - c[0] * c[2]},
                        {0, m[7] * inv - c[1] * c[1], m[8] * inv - c[1] * c[2]},
                        {0, 0, m[9] * inv - c[2] * c[2]}};
      A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
      double V[3][3], w[3];
      jacobi_eig3(A, V, w);
      int e = 0;
      if (w[1] < w[e]) e = 1;
      if (w[2] < w[e]) e = 2;
      double v[3] = {V[0][e], V[1][e], V[2][e]};
      const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (nrm > 0.0) {
        const double sign = (v[0] * bestP[0] + v[1] * bestP[1] + v[2] * bestP[2]) < 0.0 ? -1.0 : 1.0;
        for (int q = 0; q < 3; ++q) v[q] = sign * v[q] / nrm;
        bestP[0] = (float)v[0]; bestP[1] = (float)v[1]; bestP[2] = (float)v[2];
        bestP[3] = (float)(-(v[0] * c[0] + v[1] * c[1] + v[2] * c[2]));
        inliers = pf_support(xyz, n, bestP, thr);
      }
    }
  }
  memcpy(out4, bestP, sizeof(bestP));
  if (out_inliers) *out_inliers = inliers;
  return 0;
}

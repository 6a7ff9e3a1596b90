/*
 * synth.c — deterministic synthetic clouds for bench.py and tests.
 *
 * The reference ships no clouds (no .ply/.pcd, no Stanford bunny: SURVEY.md 0 D4);
 * its models arrive at run time through InitCvSurfaceMatch.srv.  These samples
 * stand in for them.  Only IEEE-exact double operations (+ - * / sqrt) are used,
 * never libm transcendentals, so every host produces the same bits and
 * tests/synth_twin.py (numpy) reproduces them exactly.
 *
 * Surface: direction u uniform on the sphere (rejection in the cube), radius
 *   r(u) = 1 + 0.25 ux uy + 0.15 uz uz ux + 0.10 uy uy uy - 0.08 uz
 * which has no symmetry, so the aligning pose is unique.
 * Counter-based PRNG: splitmix64 finaliser of (seed + ctr * golden), so point i
 * is a pure function of (seed, i): any slab of a cloud can be generated alone
 * (slab-sharded multi-GPU runs generate only their own slab).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mi355_icp.h"

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static inline double u01(uint64_t seed, uint64_t ctr) {
  uint64_t x = mix64(seed + ctr * 0x9E3779B97F4A7C15ULL);
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

static int synth_rows(uint64_t seed, uint64_t noise_seed, double sigma, const double* T, size_t first_index,
                      size_t count, float* out, int with_normals);

int mi355_icp_synth_cloud(uint64_t seed, uint64_t noise_seed, double sigma, const double* T, size_t first_index,
                          size_t count, float* out_xyz) {
  return synth_rows(seed, noise_seed, sigma, T, first_index, count, out_xyz, 0);
}

/* Same sample as rows of 6 floats: xyz, then the unit outward normal of the noise-free surface at that sample
 * (gradient of F(p) = |p| - r(p/|p|)), rotated with the transform.  The layout of the reference's model and
 * scene clouds with normals (N x 6 CV_32F: opencv_surface_match.cpp:22,57-59). */
int mi355_icp_synth_cloud6(uint64_t seed, uint64_t noise_seed, double sigma, const double* T, size_t first_index,
                           size_t count, float* out_xyz_normal) {
  return synth_rows(seed, noise_seed, sigma, T, first_index, count, out_xyz_normal, 1);
}

static int synth_rows(uint64_t seed, uint64_t noise_seed, double sigma, const double* T, size_t first_index,
                      size_t count, float* out_xyz, int with_normals) {
  const size_t row = with_normals ? 6 : 3;
  if (!out_xyz && count) return MI355_ICP_ERR_INVALID_ARGUMENT;
  const double sqrt3 = sqrt(3.0);
  for (size_t n = 0; n < count; ++n) {
    uint64_t i = (uint64_t)(first_index + n);
    double a = 1.0, b = 0.0, c = 0.0, r2 = 1.0;
    for (uint64_t k = 0; k < 64; ++k) {
      uint64_t base = (i * 64 + k) * 4;
      double ta = 2.0 * u01(seed, base + 0) - 1.0;
      double tb = 2.0 * u01(seed, base + 1) - 1.0;
      double tc = 2.0 * u01(seed, base + 2) - 1.0;
      double t2 = (ta * ta + tb * tb) + tc * tc;
      if (t2 >= 0.01 && t2 <= 1.0) { a = ta; b = tb; c = tc; r2 = t2; break; }
    }
    double inv = 1.0 / sqrt(r2);
    double ux = a * inv, uy = b * inv, uz = c * inv;
    double rad = 1.0 + 0.25 * (ux * uy);
    rad = rad + 0.15 * ((uz * uz) * ux);
    rad = rad + 0.10 * ((uy * uy) * uy);
    rad = rad - 0.08 * uz;
    double p[3] = {rad * ux, rad * uy, rad * uz};
    double nrm[3] = {0, 0, 0};
    if (with_normals) {
      const double g0 = 0.25 * uy + 0.15 * (uz * uz), g1 = 0.25 * ux + 0.30 * (uy * uy), g2 = 0.30 * (uz * ux) - 0.08;
      const double gu = (g0 * ux + g1 * uy) + g2 * uz;
      double m0 = ux - (g0 - gu * ux) / rad, m1 = uy - (g1 - gu * uy) / rad, m2 = uz - (g2 - gu * uz) / rad;
      const double ml = sqrt((m0 * m0 + m1 * m1) + m2 * m2);
      nrm[0] = m0 / ml; nrm[1] = m1 / ml; nrm[2] = m2 / ml;
    }
    if (sigma > 0.0) {
      for (int d = 0; d < 3; ++d) {
        uint64_t base = (i * 4 + (uint64_t)d) * 4;
        double s = ((u01(noise_seed, base + 0) + u01(noise_seed, base + 1)) + u01(noise_seed, base + 2)) +
                   u01(noise_seed, base + 3);
        p[d] = p[d] + sigma * ((s - 2.0) * sqrt3);
      }
    }
    if (T) {
      double q0 = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
      double q1 = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
      double q2 = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
      p[0] = q0; p[1] = q1; p[2] = q2;
      if (with_normals) {
        double r0 = (T[0] * nrm[0] + T[1] * nrm[1]) + T[2] * nrm[2];
        double r1 = (T[4] * nrm[0] + T[5] * nrm[1]) + T[6] * nrm[2];
        double r2 = (T[8] * nrm[0] + T[9] * nrm[1]) + T[10] * nrm[2];
        nrm[0] = r0; nrm[1] = r1; nrm[2] = r2;
      }
    }
    out_xyz[row * n + 0] = (float)p[0];
    out_xyz[row * n + 1] = (float)p[1];
    out_xyz[row * n + 2] = (float)p[2];
    if (with_normals) {
      out_xyz[row * n + 3] = (float)nrm[0];
      out_xyz[row * n + 4] = (float)nrm[1];
      out_xyz[row * n + 5] = (float)nrm[2];
    }
  }
  return MI355_ICP_OK;
}

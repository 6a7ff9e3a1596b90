// Times the batched (AVX-512) 6x6 solve against six scalar solves: g++ -O2 -o build/solve_bench tools/solve_bench.cpp -Lpose_estimation_amd/lib -lmi355_icp -Wl,-rpath,$PWD/pose_estimation_amd/lib -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/mi355_icp.h"
int main() {
  // SPD systems: A^T A of random rows
  const int n = 6;
  std::vector<double> sums(29 * n, 0.0);
  srand(1);
  for (int s = 0; s < n; ++s) {
    double N[6][6] = {}, g[6] = {};
    for (int r = 0; r < 200; ++r) {
      double a[6]; for (double& v : a) v = rand() / (double)RAND_MAX - 0.5;
      double b = (rand() / (double)RAND_MAX - 0.5) * 0.01;
      for (int i = 0; i < 6; ++i) { g[i] += a[i] * b; for (int j = 0; j < 6; ++j) N[i][j] += a[i] * a[j]; }
    }
    int k = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) sums[29 * s + k++] = N[i][j];
    for (int i = 0; i < 6; ++i) sums[29 * s + 21 + i] = g[i];
    sums[29 * s + 27] = 1.0; sums[29 * s + 28] = 200.0;
  }
  double x[36]; int st[6];
  const int reps = 200000;
  auto t0 = std::chrono::steady_clock::now();
  double acc = 0;
  for (int r = 0; r < reps; ++r) { mi355_icp_solve_point_to_plane_batch(sums.data(), n, x, st); acc += x[0]; }
  auto t1 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { for (int s = 0; s < n; ++s) mi355_icp_solve_point_to_plane(sums.data() + 29 * s, x + 6 * s); acc += x[0]; }
  auto t2 = std::chrono::steady_clock::now();
  std::printf("batch of %d: %.3f us, %d scalar: %.3f us (%g)\n", n, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, n,
              std::chrono::duration<double, std::micro>(t2 - t1).count() / reps, acc);
}

// Dev tool: time and check k_slam's in-wave 16 x 16 symmetric inversion (inv16) in isolation.
#include "../../drl_graph_exploration_amd/csrc/k_slam.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace kslam;

__global__ void k_inv16(const double *Din, double *Eout, long long *cyc, int np, int reps) {
  __shared__ int bad[2];
  const int lane = threadIdx.x & 63;
  SweepCtx x{0, lane, lane & 15, lane >> 4, np, 16, true, true, bad, nullptr};
  v4d d;
  long long best = 1ll << 60;
  for (int it = 0; it < reps; ++it) {
    for (int r = 0; r < 4; ++r) d[r] = Din[(x.lr + 4 * r) * 16 + x.lc];
    __builtin_amdgcn_s_waitcnt(0);
    long long t0 = clock64();
    inv16(x, 0, d);
    long long t1 = clock64() + (d[0] == 1.2345e300 ? 1 : 0);
    if (t1 - t0 < best) best = t1 - t0;
  }
  for (int r = 0; r < 4; ++r) Eout[(x.lr + 4 * r) * 16 + x.lc] = d[r];
  if (lane == 0) cyc[0] = best;
}

int main() {
  const int n = 16;
  std::vector<double> D(n * n), B(n * n);
  srand(1);
  for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
      D[i * n + j] = s * (1.0 + 50.0 * ((i % 3) == 2)) * (1.0 + 50.0 * ((j % 3) == 2)) + (i == j ? 1.0 : 0.0);
    }
  // host inverse (Gauss-Jordan, doubles)
  std::vector<double> A = D, I(n * n, 0.0);
  for (int i = 0; i < n; ++i) I[i * n + i] = 1;
  for (int k = 0; k < n; ++k) {
    double p = A[k * n + k];
    for (int j = 0; j < n; ++j) { A[k * n + j] /= p; I[k * n + j] /= p; }
    for (int i = 0; i < n; ++i) if (i != k) {
      double f = A[i * n + k];
      for (int j = 0; j < n; ++j) { A[i * n + j] -= f * A[k * n + j]; I[i * n + j] -= f * I[k * n + j]; }
    }
  }
  double *dD, *dE; long long *dc;
  hipMalloc(&dD, n * n * 8); hipMalloc(&dE, n * n * 8); hipMalloc(&dc, 8);
  hipMemcpy(dD, D.data(), n * n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_inv16, dim3(1), dim3(64), 0, 0, dD, dE, dc, 16, 20);
  hipDeviceSynchronize();
  std::vector<double> E(n * n); long long c;
  hipMemcpy(E.data(), dE, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (int i = 0; i < n * n; ++i) { err = fmax(err, fabs(E[i] + I[i])); mx = fmax(mx, fabs(I[i])); }
  printf("inv16: %lld cycles, max |E + D^-1| = %.3e (max |D^-1| = %.3e)\n", c, err, mx);
  return 0;
}

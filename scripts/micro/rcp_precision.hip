// accuracy of v_rcp_f64 / v_rsq_f64 and of one / two Newton steps on top (max relative error over random inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = r0 * (2.0 - v * r0);
  double r2 = r1 * (2.0 - v * r1);
  double e = fma(-v, r0, 1.0);
  double f1 = fma(r0, e, r0);
  double q0 = __builtin_amdgcn_rsq(v);
  double q1 = q0 * (1.5 - 0.5 * v * q0 * q0);
  double q2 = q1 * (1.5 - 0.5 * v * q1 * q1);
  o[8 * i + 0] = r0; o[8 * i + 1] = r1; o[8 * i + 2] = r2; o[8 * i + 3] = f1;
  o[8 * i + 4] = q0; o[8 * i + 5] = q1; o[8 * i + 6] = q2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(8 * n);
  srand(1);
  for (int i = 0; i < n; ++i) x[i] = std::exp((rand() / (double)RAND_MAX - 0.5) * 40.0) * (1.0 + rand() / (double)RAND_MAX);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 64);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, n * 64, hipMemcpyDeviceToHost);
  double m[7] = {0};
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / x[i], s = 1.0L / sqrtl((long double)x[i]);
    for (int j = 0; j < 4; ++j) m[j] = std::fmax(m[j], (double)fabsl((o[8 * i + j] - t) / t));
    for (int j = 4; j < 7; ++j) m[j] = std::fmax(m[j], (double)fabsl((o[8 * i + j] - s) / s));
  }
  printf("rcp raw %.3e  +1 Newton %.3e  +2 Newton %.3e  +1 fma-Newton %.3e\nrsq raw %.3e  +1 Newton %.3e  +2 Newton %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6]);
  return 0;
}

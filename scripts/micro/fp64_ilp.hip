// Micro-benchmark (dev tool): fp64 VALU latency vs throughput (1..8 independent chains), mul/add, rcp, sqrt, and the
// same beside a second wave on the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__device__ long long chains(double &sink, double a, double b, int reps) {
  double x[NCH];
  for (int c = 0; c < NCH; ++c) x[c] = a + c;
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) x[c] = __builtin_fma(x[c], b, a);
  }
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += x[c];
  sink += s * 1e-300;
  long long t1 = clock64() + (s == 1.2345e300 ? 1 : 0);
  return t1 - t0;
}

__global__ void k(long long *out, double a, double b, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a += lane * 1e-6;
  double sink = 0;
  long long r[10];
  r[0] = chains<1>(sink, a, b, reps);
  r[1] = chains<2>(sink, a, b, reps);
  r[2] = chains<4>(sink, a, b, reps);
  r[3] = chains<8>(sink, a, b, reps);
  // fp32 reference
  float xf = (float)a, bf = (float)b;
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < reps; ++i) xf = __builtin_fmaf(xf, bf, bf);
  sink += xf * 1e-30f;
  r[4] = clock64() + (xf == 1.2345e30f ? 1 : 0) - t0;
  // sqrt / rcp chains
  double y = a + 2;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < reps; ++i) y = __builtin_amdgcn_sqrt(y) + 2.0;
  sink += y * 1e-300;
  r[5] = clock64() + (y == 1.2345e300 ? 1 : 0) - t0;
  // mul + add (non fused) dependent
  double z = a;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < reps; ++i) z = z * b + a;
  sink += z * 1e-300;
  r[6] = clock64() + (z == 1.2345e300 ? 1 : 0) - t0;
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) out[wave * 12 + i] = r[i];
    out[wave * 12 + 11] = (long long)sink;
  }
}

int main() {
  long long *d, h[16 * 12];
  hipMalloc(&d, sizeof(h));
  const int reps = 256;
  for (int nthreads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, 0.999, reps);
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, 0.999, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads=%4d: fma_f64 cycles/instr with 1,2,4,8 chains: %.1f %.1f %.1f %.1f | dep fma_f32 %.1f | dep sqrt+add %.1f | dep mul,add (contracted?) %.1f\n",
           nthreads, h[0] / (double)reps, h[1] / (2.0 * reps), h[2] / (4.0 * reps), h[3] / (8.0 * reps), h[4] / (double)reps,
           h[5] / (double)reps, h[6] / (double)reps);
  }
  return 0;
}

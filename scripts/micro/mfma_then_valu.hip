// Micro-benchmark (dev tool): does fp64 VALU work stall behind in-flight fp64 MFMAs of the same wave?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)

__global__ void k(long long *out, double a, double b, int nm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a += lane * 1e-6;
  v4d c[5];
  for (int i = 0; i < 5; ++i) c[i] = v4d{0, 0, 0, 0};
  long long t[8];
  double w = b;
  SB(); t[0] = clock64(); SB();
#pragma unroll
  for (int i = 0; i < 40; ++i) w = __builtin_fma(w, b, b);
  SB(); t[1] = clock64() + (w == 1.2345e300 ? 1 : 0); SB();
  // nm independent MFMAs, then the same chain
  if (nm >= 1) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[0], 0, 0, 0);
  if (nm >= 2) c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[1], 0, 0, 0);
  if (nm >= 3) c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[2], 0, 0, 0);
  if (nm >= 4) c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[3], 0, 0, 0);
  if (nm >= 5) c[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[4], 0, 0, 0);
  SB(); t[2] = clock64(); SB();
  double v = b;
#pragma unroll
  for (int i = 0; i < 40; ++i) v = __builtin_fma(v, b, b);
  SB(); t[3] = clock64() + (v == 1.2345e300 ? 1 : 0); SB();
  double s = c[0][0] + c[1][0] + c[2][0] + c[3][0] + c[4][0];
  SB(); t[4] = clock64() + (s == 1.2345e300 ? 1 : 0); SB();
  // fp32 chain after MFMAs
  if (nm >= 1) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[0], 0, 0, 0);
  if (nm >= 2) c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[1], 0, 0, 0);
  if (nm >= 3) c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[2], 0, 0, 0);
  if (nm >= 4) c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[3], 0, 0, 0);
  if (nm >= 5) c[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[4], 0, 0, 0);
  SB(); t[5] = clock64(); SB();
  float f = (float)b, bf = (float)b;
#pragma unroll
  for (int i = 0; i < 40; ++i) f = __builtin_fmaf(f, bf, bf);
  SB(); t[6] = clock64() + (f == 1.2345e30f ? 1 : 0); SB();
  s += c[0][1] + c[1][1] + c[2][1] + c[3][1] + c[4][1];
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) out[wave * 8 + i] = t[i];
    out[wave * 8 + 7] = (long long)(s + w + v + f);
  }
}

int main() {
  long long *d, h[16 * 8];
  hipMalloc(&d, sizeof(h));
  for (int nthreads : {64, 512})
    for (int nm : {0, 1, 5}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, 0.999, nm);
      hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, 0.999, nm);
      hipDeviceSynchronize();
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("threads=%3d mfma=%d: 40 dep fma_f64 alone %lld | issue MFMAs %lld | 40 dep fma_f64 after %lld | drain %lld | 40 dep fma_f32 after MFMAs %lld\n",
             nthreads, nm, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[6] - h[5]);
    }
  return 0;
}

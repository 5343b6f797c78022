// Micro-benchmark (dev tool): issue cost / latency of v_mfma_f64_16x16x4_f64, dependent v_fma_f64 chains, v_rcp_f64,
// LDS round trips, and fp64 VALU beside in-flight fp64 MFMAs, on one wave (and on 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k(long long *out, double seed, int reps) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = seed + lane * 1e-3, b = 1.0 + lane * 1e-4;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0;
  long long t[12];
  // 1. dependent MFMA chain
  t[0] = clock64();
  for (int i = 0; i < reps; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  a += c0[0] * 1e-300;
  t[1] = clock64();
  // 2. 5 independent MFMAs per iteration
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
  }
  a += (c0[0] + c1[0] + c2[0] + c3[0] + c4[0]) * 1e-300;
  t[2] = clock64();
  // 3. dependent fma chain
  double x = a;
  for (int i = 0; i < reps * 8; ++i) x = __builtin_fma(x, b, a);
  a += x * 1e-300;
  t[3] = clock64();
  // 4. dependent rcp chain
  double y = b;
  for (int i = 0; i < reps; ++i) y = __builtin_amdgcn_rcp(y) + 1.0;
  a += y * 1e-300;
  t[4] = clock64();
  // 5. LDS write -> read round trips (dependent)
  double z = a;
  for (int i = 0; i < reps; ++i) {
    lds[(lane + wave * 64) ^ 1] = z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    z = lds[lane + wave * 64] + 1.0;
  }
  a += z * 1e-300;
  t[5] = clock64();
  // 6. dependent fma chain right after issuing 5 independent MFMAs
  c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
  c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
  c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
  t[6] = clock64();
  double w = b;
  for (int i = 0; i < 40; ++i) w = __builtin_fma(w, b, b);
  t[7] = clock64() + (w == 1.2345e300 ? 1 : 0);
  a += (c0[1] + c1[1] + c2[1] + c3[1] + c4[1] + w) * 1e-300;
  t[8] = clock64();
  // 7. barrier cost
  for (int i = 0; i < reps; ++i) __syncthreads();
  t[9] = clock64();
  if (lane == 0) {
    for (int i = 0; i < 10; ++i) out[(blockIdx.x * 16 + wave) * 12 + i] = t[i];
    out[(blockIdx.x * 16 + wave) * 12 + 10] = (long long)(a);
  }
}

int main() {
  long long *d, h[16 * 12 * 2];
  hipMalloc(&d, sizeof(h));
  const int reps = 64;
  for (int nthreads : {64, 512}) {
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, reps);
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1.0, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long long *t = h;
    printf("threads=%d (wave 0): dep-MFMA %.1f cyc each | 5 indep MFMA %.1f cyc each | dep fma_f64 %.1f | dep rcp+add %.1f | LDS w->r %.1f | "
           "issue 5 MFMA %lld, then 40 dep fma %lld (%.1f each), drain %lld | barrier %.1f\n",
           nthreads, (double)(t[1] - t[0]) / reps, (double)(t[2] - t[1]) / (5.0 * reps), (double)(t[3] - t[2]) / (8.0 * reps),
           (double)(t[4] - t[3]) / reps, (double)(t[5] - t[4]) / reps, t[6] - t[5], t[7] - t[6], (double)(t[7] - t[6]) / 40.0, t[8] - t[7],
           (double)(t[9] - t[8]) / reps);
  }
  return 0;
}

// Micro-benchmark (dev tool): sustained v_mfma_f32_32x32x2_f32 rate of the whole chip with no memory traffic
// (4 independent accumulators per wave, W waves per SIMD), to separate "clock / pipe ceiling" from GEMM-kernel losses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float *out, int reps) {
  const int lane = threadIdx.x & 63;
  float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
  floatx16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    for (int reps : {2000, 8000}) {
      const int grid = 256 * wgs_per_cu;
      k<<<grid, 256>>>(out, reps);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<grid, 256>>>(out, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * reps * 4 * 4096.0;
      printf("wgs/cu %d reps %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wgs_per_cu, reps, ms,
             flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / (reps * 4.0 * wgs_per_cu));
    }
  }
  return 0;
}

// Dev tool: which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID bits [5:4] = SIMD_ID, [3:0] = WAVE_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  const int wave = threadIdx.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = (int)hw;
}
int main() {
  int *d, h[64];
  hipMalloc(&d, sizeof(h));
  for (int nt : {512, 1024}) {
    hipLaunchKernelGGL(k, dim3(4), dim3(nt), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("threads=%d block %d: ", nt, b);
      for (int w = 0; w < nt / 64; ++w) printf("w%d:simd%d/slot%d ", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15);
      printf("\n");
    }
  }
  return 0;
}

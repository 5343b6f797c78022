// Micro-benchmark (dev tool): the fp32 MFMA GEMM of k_gcn.hip on the shapes the GCN uses, without the rest of the layer.
// build: hipcc -O3 --offload-arch=gfx950 -I include -I drl_graph_exploration_amd/csrc -o scripts/micro/gemm_bench.bin scripts/micro/gemm_bench.hip
#include "../../drl_graph_exploration_amd/csrc/k_gcn.hip"
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
  const int hidden = 1000;
  std::vector<int> Ms = {4340, 17287, 8192, 32768};
  float *A, *B, *C, *bias, *mask;
  const size_t maxM = 32768;
  hipMalloc(&A, maxM * hidden * 4);
  hipMalloc(&B, (size_t)hidden * hidden * 4);
  hipMalloc(&C, maxM * hidden * 4);
  hipMalloc(&bias, hidden * 4);
  hipMalloc(&mask, maxM * hidden * 4);
  hipMemset(A, 0, maxM * hidden * 4);
  hipMemset(B, 0, (size_t)hidden * hidden * 4);
  hipMemset(bias, 0, hidden * 4);
  hipMemset(mask, 0, maxM * hidden * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int M : Ms) {
    for (int which = 0; which < 2; ++which) {
      auto run = [&]() {
        if (which == 0) gemm<false, false, 1>(0, M, hidden, hidden, A, hidden, B, hidden, C, hidden, bias, mask, 1);
        else gemm<false, true, 0>(0, M, hidden, hidden, A, hidden, B, hidden, C, hidden, nullptr, nullptr, 1);
      };
      for (int i = 0; i < 3; ++i) run();
      hipDeviceSynchronize();
      const int reps = 20;
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= reps;
      printf("%s M=%6d N=K=%d: %8.1f us  %6.1f TFLOP/s\n", which == 0 ? "NN+epi" : "NT    ", M, hidden, ms * 1e3,
             2.0 * M * hidden * hidden / ms * 1e-9);
    }
  }
  return 0;
}

// Experiment (dev tool): the three hidden x hidden products of a GCN train step (forward with epilogue, dZ2 W2^T, split-K
// AH1^T dZ2) through the tall-tile kernel k_gemm_wide at every tile height, against the 64x64 direct-to-LDS kernel
// (both in csrc/k_gcn.hip): result check and time.
// build: hipcc -O3 --offload-arch=gfx950 -I include -I drl_graph_exploration_amd/csrc -o scripts/micro/gemm_wide_bench.bin scripts/micro/gemm_wide_bench.hip
#include "../../drl_graph_exploration_amd/csrc/k_gcn.hip"
#include <cstdio>
#include <vector>

namespace {
// rt: 0 = the 64x64 kernel, 6..10 = 8-wave tall tiles (rt x 16 rows x 128 columns), 106..110 = 4-wave ones (x 64 columns)
template <bool TA, bool TB, int EPI>
void run_rt(int rt, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, const float *bias, const float *mask, int kps) {
#define ARGS 0, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps
  switch (rt) {
    case 0: gemm_tile<TA, TB, EPI, 1, 1>(ARGS); break;
    case 6: gemm_wide_launch<TA, TB, EPI, 6, 8>(ARGS); break;
    case 8: gemm_wide_launch<TA, TB, EPI, 8, 8>(ARGS); break;
    case 10: gemm_wide_launch<TA, TB, EPI, 10, 8>(ARGS); break;
    case 106: gemm_wide_launch<TA, TB, EPI, 6, 4>(ARGS); break;
    case 108: gemm_wide_launch<TA, TB, EPI, 8, 4>(ARGS); break;
    case 110: gemm_wide_launch<TA, TB, EPI, 10, 4>(ARGS); break;
    default:
      if constexpr (!TA) {
        if (rt == 7) gemm_wide_launch<TA, TB, EPI, 7, 8>(ARGS);
        if (rt == 9) gemm_wide_launch<TA, TB, EPI, 9, 8>(ARGS);
        if (rt == 107) gemm_wide_launch<TA, TB, EPI, 7, 4>(ARGS);
        if (rt == 109) gemm_wide_launch<TA, TB, EPI, 9, 4>(ARGS);
      }
  }
#undef ARGS
}
}  // namespace

int main() {
  const int hidden = 1000;
  float *A, *B, *C0, *C1, *bias, *mask;
  const size_t maxM = 17288;
  hipMalloc(&A, maxM * hidden * 4); hipMalloc(&B, (size_t)hidden * hidden * 4);
  hipMalloc(&C0, maxM * hidden * 4); hipMalloc(&C1, maxM * hidden * 4);
  hipMalloc(&bias, hidden * 4); hipMalloc(&mask, maxM * hidden * 4);
  std::vector<float> h(maxM * hidden);
  srand(1);
  for (auto &v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(A, h.data(), maxM * hidden * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < (size_t)hidden * hidden; ++i) h[i] = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(B, h.data(), (size_t)hidden * hidden * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < hidden; ++i) h[i] = (rand() % 201 - 100) * 1e-2f;
  hipMemcpy(bias, h.data(), hidden * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < maxM * hidden; ++i) h[i] = (rand() & 1) ? 2.f : 0.f;
  hipMemcpy(mask, h.data(), maxM * hidden * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float *part;
  hipMalloc(&part, (size_t)8 * hidden * hidden * 4);
  for (int M : {1300, 2100, 2880, 4340, 17288}) {
    for (int shape = 0; shape < 3; ++shape) {  // 0: NN + epilogue (forward), 1: NT (T1 = dZ2 W2^T), 2: TN split-K (dW2 = AH1^T dZ2)
      const int gk = shape == 2 ? M : hidden;
      const int splits = shape == 2 ? 4 : 1;
      const int kps = ((gk + splits - 1) / splits + 15) / 16 * 16, S = (gk + kps - 1) / kps;
      const size_t n = (size_t)(shape == 2 ? hidden : M) * hidden;
      std::vector<float> c0(n), c1(n);
      for (int rt : {0, 6, 7, 8, 9, 10, 106, 107, 108, 109, 110}) {
        if (shape == 2 && (rt & 1)) continue;
        float *Cx = rt == 0 ? C0 : C1;
        hipMemset(Cx, 0xff, n * 4);
        auto run = [&]() {
          if (shape == 0) run_rt<false, false, 1>(rt, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, bias, mask, kps);
          else if (shape == 1) run_rt<false, true, 0>(rt, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, nullptr, nullptr, kps);
          else run_rt<true, false, 0>(rt, hidden, hidden, M, A, hidden, mask, hidden, S == 1 ? Cx : part, hidden, nullptr, nullptr, kps);
          if (shape == 2 && S > 1) hipLaunchKernelGGL(k_splitk_reduce, dim3((hidden * hidden + 255) / 256), dim3(256), 0, 0, hidden * hidden, S, part, Cx);
        };
        for (int i = 0; i < 3; ++i) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        double md = 0, mx = 0;
        if (rt == 0) hipMemcpy(c0.data(), C0, n * 4, hipMemcpyDeviceToHost);
        else {
          hipMemcpy(c1.data(), C1, n * 4, hipMemcpyDeviceToHost);
          for (size_t i = 0; i < n; ++i) { md = std::max(md, fabs((double)c0[i] - c1[i])); mx = std::max(mx, fabs((double)c0[i])); }
        }
        printf("shape %d M=%6d tile %3d x %3d (splits %d): %8.1f us  %6.1f TFLOP/s   max |diff to 64x64| %.3e (max |value| %.3e)  %s\n", shape, M,
               rt ? 16 * (rt % 100) : 64, rt == 0 ? 64 : rt > 100 ? 64 : 128, S, ms * 1e3, 2.0 * M * hidden * hidden / ms * 1e-9, md, mx, hipGetErrorString(hipGetLastError()));
      }
    }
  }
  return 0;
}

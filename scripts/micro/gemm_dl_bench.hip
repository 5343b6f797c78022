// Experiment (dev tool): the forward GCN GEMM  C = relu(A B + bias) * mask  (A [M x K] and B [K x N] row-major) with the
// operand tiles loaded global -> LDS directly (global_load_lds_dwordx4, gfx950: no VGPR staging, no ds_write), against
// k_gemm of csrc/k_gcn.hip on the same shapes: result check and time.
// build: hipcc -O3 --offload-arch=gfx950 -I include -I drl_graph_exploration_amd/csrc -o scripts/micro/gemm_dl_bench.bin scripts/micro/gemm_dl_bench.hip
#include "../../drl_graph_exploration_amd/csrc/k_gcn.hip"
#include <cstdio>
#include <vector>

namespace {

constexpr int DL_ST = 4;  // LDS stages: tile t + 3 is in flight while tile t is multiplied

// C[M x N] = op(A) op(B) with the operand tiles loaded global -> LDS directly.  AKC / BKC: the operand's source is
// k-contiguous (A stored [M][K] / B stored [N][K]), else x-contiguous (A stored [K][M] / B stored [K][N]).
//   k-contiguous tile  [64 x][16 k] unpadded, 16-byte quads of a row XOR-swizzled by (x >> 1) & 3 (conflict-free
//                      ds_read_b128 over 8 consecutive rows): wave w loads rows 16w .. 16w+15 (lane = 4 row + quad);
//   x-contiguous tile  [16 k][64 x] unpadded: wave w loads k rows 4w .. 4w+3 (lane = 16 k + x quad).
// One global_load_lds_dwordx4 per wave and operand brings 1 KB.  Contract (host): lda, ldb, the contiguous extents and the
// base addresses are multiples of 4 floats; extents >= 4.
template <bool KC>
__device__ __forceinline__ const float *dl_src(const float *P, int ld, int x0, int X, int wave, int lane) {
  if (KC) {
    const int x = 16 * wave + (lane >> 2);
    return P + (size_t)min(x0 + x, X - 1) * ld + 4 * ((lane & 3) ^ ((x >> 1) & 3));  // (+ k0)
  }
  return P + (size_t)(4 * wave + (lane >> 4)) * ld + min(x0 + 4 * (lane & 15), X - 4);  // (+ k0 * ld)
}
template <bool KC>
__device__ __forceinline__ void dl_frag(float (&f)[8], const float *T, int xb, int lane) {
  const int li = lane & 31, h = lane >> 5, x = xb + li;
  if (KC) {
    const int sw = (x >> 1) & 3;
    const float4 u0 = *reinterpret_cast<const float4 *>(T + x * 16 + 4 * ((2 * h) ^ sw));
    const float4 u1 = *reinterpret_cast<const float4 *>(T + x * 16 + 4 * ((2 * h + 1) ^ sw));
    f[0] = u0.x; f[1] = u0.y; f[2] = u0.z; f[3] = u0.w; f[4] = u1.x; f[5] = u1.y; f[6] = u1.z; f[7] = u1.w;
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = T[(8 * h + s) * 64 + x];
  }
}
// the partial last K-tile goes through registers with zero fill (k >= kend must contribute nothing)
template <bool KC>
__device__ __forceinline__ void dl_tail(float *T, const float *P, int ld, int x0, int X, int k0, int kend, int tid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (KC) {
    const int x = tid >> 2, q = tid & 3;
    if (k0 + 4 * q < kend) v = *reinterpret_cast<const float4 *>(P + (size_t)min(x0 + x, X - 1) * ld + k0 + 4 * q);
    *reinterpret_cast<float4 *>(T + x * 16 + 4 * (q ^ ((x >> 1) & 3))) = v;
  } else {
    const int k = tid >> 4, xq = tid & 15;
    if (k0 + k < kend) v = *reinterpret_cast<const float4 *>(P + (size_t)(k0 + k) * ld + min(x0 + 4 * xq, X - 4));
    *reinterpret_cast<float4 *>(T + k * 64 + 4 * xq) = v;
  }
}

template <bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(256) void k_gemm_dl(int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                 int ldb, float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                                 const float *__restrict__ mask, int k_per_split, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[DL_ST][64 * 16];
  __shared__ __attribute__((aligned(16))) float Bs[DL_ST][16 * 64];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * 64, n0 = tn * 64;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nfull = (kend - kbeg) / 16, tail = (kend - kbeg) - 16 * nfull;
  const float *ga = dl_src<AKC>(A, lda, m0, M, wave, lane) + (AKC ? (size_t)kbeg : (size_t)kbeg * lda);
  const float *gb = dl_src<BKC>(B, ldb, n0, N, wave, lane) + (BKC ? (size_t)kbeg : (size_t)kbeg * ldb);
  const size_t sa = AKC ? 16 : (size_t)16 * lda, sb = BKC ? 16 : (size_t)16 * ldb;  // source step per K-tile
  // (inline assembly: through the builtin the compiler knows that the load writes LDS and drains every load in flight
  //  - s_waitcnt vmcnt(0) - before the next LDS read, which is exactly the overlap this kernel is about)
  auto lds_off = [](const float *p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const void *)p);
  };
  auto dma16 = [](const float *g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");  // (m0 is not otherwise used in this kernel: gfx9 LDS instructions do not read it)
  };
  auto issue = [&](int t) {
    const int st = t & (DL_ST - 1);
    dma16(ga + sa * t, lds_off(&As[st][wave * 256]));
    dma16(gb + sb * t, lds_off(&Bs[st][wave * 256]));
  };
  auto multiply = [&](int st) {
    float fa[8], fb[8];
    dl_frag<AKC>(fa, As[st], wm * 32, lane);
    dl_frag<BKC>(fb, Bs[st], wn * 32, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
  };
  if (nfull > 0) issue(0);
  if (nfull > 1) issue(1);
  if (nfull > 2) issue(2);
  for (int t = 0; t < nfull; ++t) {
    // tile t has landed when at most the loads of tiles t+1 and t+2 (two instructions each) are still in flight
    if (t + 2 < nfull) __builtin_amdgcn_s_waitcnt(0x0F74);       // vmcnt(4)
    else if (t + 1 < nfull) __builtin_amdgcn_s_waitcnt(0x0F72);  // vmcnt(2)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __builtin_amdgcn_s_barrier();  // (no fence: a fence would drain the loads in flight) every wave's part of tile t is in
                                   // LDS; every wave is done with tile t-1, whose buffer is refilled next
    if (t + 3 < nfull) issue(t + 3);
    multiply(t & (DL_ST - 1));
  }
  if (tail > 0) {
    __syncthreads();
    const int st = nfull & (DL_ST - 1), k0 = kbeg + 16 * nfull;
    dl_tail<AKC>(As[st], A, lda, m0, M, k0, kend, tid);
    dl_tail<BKC>(Bs[st], B, ldb, n0, N, k0, kend, tid);
    __syncthreads();
    multiply(st);
  }
  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  // (the clamped loads of an edge tile only disturb rows >= M / columns >= N, which are not stored)
  float *Cz = C + (EPI == 0 ? (size_t)blockIdx.z * M * ldc : 0);
  const int col = n0 + wn * 32 + (lane & 31);
  if (col < N) {
    const float bj = EPI == 1 ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
      if (row >= M) continue;
      float v = acc[r];
      if (EPI == 1) {
        v = fmaxf(v + bj, 0.f);
        if (mask) v *= mask[(size_t)row * ldc + col];
      }
      Cz[(size_t)row * ldc + col] = v;
    }
  }
}

}  // namespace

int main() {
  const int hidden = 1000;
  float *A, *B, *C0, *C1, *bias, *mask;
  const size_t maxM = 17287;
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
  for (int M : {4340, 17287}) {
    for (int shape = 0; shape < 3; ++shape) {  // 0: NN + epilogue (forward), 1: NT (T1 = dZ2 W2^T), 2: TN split-K (dW2 = AH1^T dZ2)
      const int gm = shape == 2 ? hidden : M, gn = hidden, gk = shape == 2 ? M : hidden;
      const int tiles_m = (gm + 63) / 64, tiles_n = (gn + 63) / 64;
      int splits = 1;
      if (shape == 2) splits = std::max(1, std::min({8, (int)((1024 + tiles_m * tiles_n - 1) / (tiles_m * tiles_n)), (gk + 255) / 256}));
      const int kps = ((gk + splits - 1) / splits + 15) / 16 * 16, S = (gk + kps - 1) / kps;
      dim3 grid(tiles_n * ((tiles_m + 7) / 8) * 8, 1, S);
      for (int which = 0; which < 2; ++which) {
        float *Cx = which == 0 ? C0 : C1;
        auto run = [&]() {
          if (which == 0) {
            if (shape == 0) gemm<false, false, 1>(0, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, bias, mask, 1);
            else if (shape == 1) gemm<false, true, 0>(0, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, nullptr, nullptr, 1);
            else gemm<true, false, 0>(0, hidden, hidden, M, A, hidden, mask, hidden, S == 1 ? Cx : part, hidden, nullptr, nullptr, S);
          } else {
            if (shape == 0) hipLaunchKernelGGL((k_gemm_dl<true, false, 1>), grid, dim3(256), 0, 0, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, bias, mask, kps, tiles_m, tiles_n);
            else if (shape == 1) hipLaunchKernelGGL((k_gemm_dl<true, true, 0>), grid, dim3(256), 0, 0, M, hidden, hidden, A, hidden, B, hidden, Cx, hidden, nullptr, nullptr, kps, tiles_m, tiles_n);
            else hipLaunchKernelGGL((k_gemm_dl<false, false, 0>), grid, dim3(256), 0, 0, hidden, hidden, M, A, hidden, mask, hidden, S == 1 ? Cx : part, hidden, nullptr, nullptr, kps, tiles_m, tiles_n);
          }
          if (shape == 2 && S > 1) hipLaunchKernelGGL(k_splitk_reduce, dim3((hidden * hidden + 255) / 256), dim3(256), 0, 0, hidden * hidden, S, part, Cx);
        };
        for (int i = 0; i < 3; ++i) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("shape %d %s M=%6d (splits %d): %8.1f us  %6.1f TFLOP/s\n", shape, which == 0 ? "k_gemm   " : "k_gemm_dl", M, S, ms * 1e3,
               2.0 * M * hidden * hidden / ms * 1e-9);
      }
      const size_t n = (size_t)(shape == 2 ? hidden : M) * hidden;
      std::vector<float> c0(n), c1(n);
      hipMemcpy(c0.data(), C0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c1.data(), C1, n * 4, hipMemcpyDeviceToHost);
      double md = 0, mx = 0;
      for (size_t i = 0; i < n; ++i) { md = std::max(md, fabs((double)c0[i] - c1[i])); mx = std::max(mx, fabs((double)c0[i])); }
      printf("  max |diff| %.3e (max |value| %.3e)\n", md, mx);
    }
  }
  return 0;
}

// GCN policy network on gfx950: forward and backward of
//   H1 = relu(Â X W1 + b1),  H2 = relu(Â H1 W2 + b2) [* dropout mask],  out = H2 Wf^T + bf
// with Â = D^-1/2 (A_w + 2I) D^-1/2, D = rowsum(A_w + 2I)   (PyG 1.x GCNConv(improved=True), SURVEY.md App. B;
// scripts/Networks.py:12-70 GCN / PolicyGCN / ValueGCN trunks; scripts/policy.py:234-253 for the backward).
//
// Layout / kernels
//   * the batch is irregular (one graph per env): edges are turned into two CSRs (by destination for the
//     forward aggregation, by source for the transposed one) with deterministic per-row order;
//   * aggregation  (Â H)  : one workgroup per node row, float4 lanes across the 1000 features, neighbour rows
//     gathered with coalesced 4 KB reads (HBM/L2-bound); self-loop weight 2/deg fused in;
//   * layer 1 (K = 5) is VALU work fused with the aggregation of X:  H1 = relu((Â X) W1 + b1);
//   * the dense 1000x1000 contractions run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
//     157 TFLOP/s peak): 128x128 block tile, 4 waves x (2x2) 32x32 MFMA tiles, LDS double buffering,
//     fused bias+ReLU(+mask) epilogue; the weight-gradient products (K = #nodes) use split-K with a
//     deterministic second-stage reduction.
// fp32 throughout (the reference trains in fp32); (Â X) W1 is used instead of Â (X W1) — same value up to fp32
// rounding (tests: <= 2e-5 relative against the plain-torch reference).
#include <algorithm>

#include "drlgx_dev.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// graph normalisation + CSR
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_f32(float *p, float v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_i32(int *p, int v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// in/out degree counts for the two CSRs (the weighted degree is summed later in edge order: float atomics here would
// make deg - and through the ReLU gates the whole forward/backward - depend on the arrival order)
__global__ void k_degree(int E, const int64_t *ei, int *cnt_dst, int *cnt_src) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int r = (int)ei[e], c = (int)ei[(size_t)E + e];
  if (r == c) return;  // explicit self loops are folded into the self term (none in this application)
  atomicAdd(&cnt_dst[c], 1);
  atomicAdd(&cnt_src[r], 1);
}
// exclusive scan of two count arrays (single block; N is a few 10^4)
__global__ void k_scan2(int n, const int *a, int *pa, const int *b, int *pb) {
  __shared__ int part[2][1024];
  const int t = threadIdx.x, nt = blockDim.x;
  const int chunk = (n + nt - 1) / nt;
  const int i0 = t * chunk, i1 = min(n, i0 + chunk);
  int sa = 0, sb = 0;
  for (int i = i0; i < i1; ++i) {
    sa += a[i];
    sb += b[i];
  }
  part[0][t] = sa;
  part[1][t] = sb;
  __syncthreads();
  if (t == 0) {
    int ra = 0, rb = 0;
    for (int k = 0; k < nt; ++k) {
      int x = part[0][k];
      part[0][k] = ra;
      ra += x;
      x = part[1][k];
      part[1][k] = rb;
      rb += x;
    }
    pa[n] = ra;
    pb[n] = rb;
  }
  __syncthreads();
  sa = part[0][t];
  sb = part[1][t];
  for (int i = i0; i < i1; ++i) {
    pa[i] = sa;
    sa += a[i];
    pb[i] = sb;
    sb += b[i];
  }
}
__global__ void k_csr_fill(int E, const int64_t *ei, const int *ptr_dst, int *cur_dst, int *eid_dst, const int *ptr_src,
                           int *cur_src, int *eid_src) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int r = (int)ei[e], c = (int)ei[(size_t)E + e];
  if (r == c) return;
  eid_dst[ptr_dst[c] + atomicAdd(&cur_dst[c], 1)] = e;
  eid_src[ptr_src[r] + atomicAdd(&cur_src[r], 1)] = e;
}
// sort each CSR row by edge id (rows are short) -> deterministic summation order
__global__ void k_csr_sort(int N, const int *ptr, int *eid) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int a = ptr[n], b = ptr[n + 1];
  for (int i = a + 1; i < b; ++i) {
    int v = eid[i], j = i - 1;
    while (j >= a && eid[j] > v) {
      eid[j + 1] = eid[j];
      --j;
    }
    eid[j + 1] = v;
  }
}
// deg[row] = sum of the row's edge weights in edge order, then the self loop weight 2 appended by
// add_remaining_self_loops(fill_value = 2)  (PyG: scatter_add(edge_weight, row) with row = source)
__global__ void k_degree_sum(int N, const float *ew, const int *ptr_src, const int *eid_src, float *deg) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int i = ptr_src[n]; i < ptr_src[n + 1]; ++i) s += ew[eid_src[i]];
  deg[n] = s + 2.0f;
}
// resolve (neighbour, normalised weight) per CSR slot.  dis = deg^-1/2 (inf -> 0).
__global__ void k_csr_finish(int N, int E, const int64_t *ei, const float *ew, const float *deg, const int *ptr, const int *eid,
                             int *nbr, float *wn, int by_dst) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int i = ptr[n]; i < ptr[n + 1]; ++i) {
    const int e = eid[i];
    const int r = (int)ei[e], c = (int)ei[(size_t)E + e];
    float dr = deg[r] > 0 ? 1.0f / sqrtf(deg[r]) : 0.0f, dc = deg[c] > 0 ? 1.0f / sqrtf(deg[c]) : 0.0f;
    nbr[i] = by_dst ? r : c;
    wn[i] = dr * ew[e] * dc;  // deg^-1/2[row] * w * deg^-1/2[col]
  }
}

// ------------------------------------------------------------------------------------------------
// layer 1: AX = Â X (5 features, padded to 8) and H1 = relu(AX W1 + b1); one workgroup per node
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layer1(int N, int in_dim, int hidden, const float *x, const float *deg, const int *ptr,
                                                const int *nbr, const float *wn, const float *W1, const float *b1, float *AX,
                                                float *H1) {
  const int n = blockIdx.x, t = threadIdx.x;
  __shared__ float ax[8];
  if (t < 8) {
    float s = 0.f;
    if (t < in_dim) {
      s = (2.0f / deg[n]) * x[(size_t)n * in_dim + t];  // self loop: dis * 2 * dis
      for (int i = ptr[n]; i < ptr[n + 1]; ++i) s += wn[i] * x[(size_t)nbr[i] * in_dim + t];
    }
    ax[t] = s;
    AX[(size_t)n * 8 + t] = s;
  }
  __syncthreads();
  for (int c = t; c < hidden; c += 256) {
    float s = b1[c];
    for (int k = 0; k < in_dim; ++k) s += ax[k] * W1[(size_t)k * hidden + c];
    H1[(size_t)n * hidden + c] = fmaxf(s, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// aggregation out[n] = (2/deg[n]) H[n] + sum_i wn[i] H[nbr[i]]  (+ optional ReLU-gate by `gate` > 0)
// one workgroup per node, float4 per lane
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_aggregate(int N, int hidden, const float *H, const float *deg, const int *ptr, const int *nbr,
                                                   const float *wn, const float *gate, float *out) {
  const int n = blockIdx.x;
  const int h4 = hidden >> 2;
  const float self = 2.0f / deg[n];
  const int a = ptr[n], b = ptr[n + 1];
  for (int c = threadIdx.x; c < h4; c += 256) {
    float4 v = reinterpret_cast<const float4 *>(H + (size_t)n * hidden)[c];
    float4 acc = make_float4(self * v.x, self * v.y, self * v.z, self * v.w);
    for (int i = a; i < b; ++i) {
      const float w = wn[i];
      const float4 u = reinterpret_cast<const float4 *>(H + (size_t)nbr[i] * hidden)[c];
      acc.x += w * u.x; acc.y += w * u.y; acc.z += w * u.z; acc.w += w * u.w;
    }
    if (gate) {
      const float4 g = reinterpret_cast<const float4 *>(gate + (size_t)n * hidden)[c];
      acc.x = g.x > 0.f ? acc.x : 0.f; acc.y = g.y > 0.f ? acc.y : 0.f;
      acc.z = g.z > 0.f ? acc.z : 0.f; acc.w = g.w > 0.f ? acc.w : 0.f;
    }
    reinterpret_cast<float4 *>(out + (size_t)n * hidden)[c] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM  C[M x N] = op(A) op(B)   (row-major; TA: A is stored [K x M]; TB: B is stored [N x K])
// block 128x128x16, 256 threads = 2x2 waves, each wave 2x2 tiles of v_mfma_f32_32x32x2_f32
// EPI 0: C = acc (split-K partial when gridDim.z > 1: C += z * M * N)
// EPI 1: C = relu(acc + bias[col]) * (mask ? mask[row][col] : 1)        (forward layer 2)
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 16, LDA_S = BM + 4, LDB_S = BN + 4;

template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                              int ldb, float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                              const float *__restrict__ mask, int k_per_split) {
  __shared__ float As[2][BK][LDA_S];
  __shared__ float Bs[2][BK][LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging registers: each thread moves 8 floats of A and 8 of B per K-tile
  float ra[8], rb[8];
  auto load_tiles = [&](int k0) {
    if (!TA) {  // A[m][k]: 128 rows x 16 k; thread -> (row = tid/4 + 64 r, k4 = (tid%4)*4)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = m0 + (tid >> 2) + 64 * r, k = k0 + (tid & 3) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[r * 4 + c] = (row < M && k + c < kend) ? A[(size_t)row * lda + k + c] : 0.f;
      }
    } else {  // A stored [K][M]: 16 k x 128 m; thread -> (k = tid/32 + 8 r, m4 = (tid%32)*4)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int k = k0 + (tid >> 5) + 8 * r, m = m0 + (tid & 31) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[r * 4 + c] = (k < kend && m + c < M) ? A[(size_t)k * lda + m + c] : 0.f;
      }
    }
    if (!TB) {  // B[k][n]
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int k = k0 + (tid >> 5) + 8 * r, n = n0 + (tid & 31) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[r * 4 + c] = (k < kend && n + c < N) ? B[(size_t)k * ldb + n + c] : 0.f;
      }
    } else {  // B stored [N][K]
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int col = n0 + (tid >> 2) + 64 * r, k = k0 + (tid & 3) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[r * 4 + c] = (col < N && k + c < kend) ? B[(size_t)col * ldb + k + c] : 0.f;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    if (!TA) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) As[buf][(tid & 3) * 4 + c][(tid >> 2) + 64 * r] = ra[r * 4 + c];
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) As[buf][(tid >> 5) + 8 * r][(tid & 31) * 4 + c] = ra[r * 4 + c];
    }
    if (!TB) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[buf][(tid >> 5) + 8 * r][(tid & 31) * 4 + c] = rb[r * 4 + c];
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[buf][(tid & 3) * 4 + c][(tid >> 2) + 64 * r] = rb[r * 4 + c];
    }
  };

  const int ntile = (kend - kbeg + BK - 1) / BK;
  if (ntile > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) load_tiles(kbeg + (t + 1) * BK);  // global loads in flight during the MFMAs
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kr = kk + (lane >> 5), li = lane & 31;
      const float a0 = As[buf][kr][wm * 64 + li], a1 = As[buf][kr][wm * 64 + 32 + li];
      const float b0 = Bs[buf][kr][wn * 64 + li], b1 = Bs[buf][kr][wn * 64 + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (t + 1 < ntile) store_tiles(buf ^ 1);
    __syncthreads();
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *Cz = C + (EPI == 0 ? (size_t)blockIdx.z * M * ldc : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (row < M && col < N) {
          float v = acc[i][j][r];
          if (EPI == 1) {
            v = fmaxf(v + bias[col], 0.f);
            if (mask) v *= mask[(size_t)row * ldc + col];
          }
          Cz[(size_t)row * ldc + col] = v;
        }
      }
}

// deterministic second stage of split-K: out = sum_z part[z]
__global__ void k_splitk_reduce(int n, int S, const float *part, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(size_t)z * n + i];
  out[i] = s;
}

// out[n][o] = sum_c H2m[n][c] Wf[o][c] + bf[o]    (Linear 1000 -> out_dim); one wave per (node, o-chunk)
__global__ __launch_bounds__(256) void k_linear_out(int N, int hidden, int out_dim, const float *H2m, const float *Wf, const float *bf,
                                                    float *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const float4 *h = reinterpret_cast<const float4 *>(H2m + (size_t)n * hidden);
  const int h4 = hidden >> 2;
  for (int o = 0; o < out_dim; ++o) {
    const float4 *w = reinterpret_cast<const float4 *>(Wf + (size_t)o * hidden);
    float s = 0.f;
    for (int c = lane; c < h4; c += 64) {
      const float4 a = h[c], b = w[c];
      s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) out[(size_t)n * out_dim + o] = s + bf[o];
  }
}

// dZ2[n][c] = (sum_o dOut[n][o] Wf[o][c]) * mask[n][c] * (H2m > 0 <=> pre-activation > 0 and mask != 0)
// H2 holds relu(Z2) * mask, so "active" = (H2 != 0) when mask is a dropout mask of {0, 1/(1-p)}; the relu gate
// is recovered from H2 itself: Z2 > 0 and mask > 0  <=>  H2 > 0.
__global__ __launch_bounds__(256) void k_dz2(int N, int hidden, int out_dim, const float *dOut, const float *Wf, const float *mask,
                                             const float *H2, float *dZ2) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < hidden; c += 256) {
    float s = 0.f;
    for (int o = 0; o < out_dim; ++o) s += dOut[(size_t)n * out_dim + o] * Wf[(size_t)o * hidden + c];
    const float h = H2[(size_t)n * hidden + c];
    float g = h > 0.f ? 1.f : 0.f;
    if (mask) g *= mask[(size_t)n * hidden + c];
    dZ2[(size_t)n * hidden + c] = s * g;
  }
}

// column sums (bias gradients): out[c] = sum_n X[n][c]; two deterministic stages
__global__ __launch_bounds__(256) void k_colsum_part(int N, int C, const float *X, float *part, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(N, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += X[(size_t)r * C + c];
  part[(size_t)blockIdx.y * C + c] = s;
}

struct GcnWs {
  float *deg, *wn_dst, *wn_src, *AX, *H1, *AH1, *H2, *T0, *T1, *part;
  int *cnt_dst, *cnt_src, *ptr_dst, *ptr_src, *cur_dst, *cur_src, *eid_dst, *eid_src, *nbr_dst, *nbr_src;
  size_t part_floats;
};

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t carve(GcnWs *w, char *base, int N, int E, int hidden, int out_dim) {
  size_t off = 0;
  auto takef = [&](float **p, size_t n) {
    if (w) *p = reinterpret_cast<float *>(base + off);
    off += align256(n * sizeof(float));
  };
  auto takei = [&](int **p, size_t n) {
    if (w) *p = reinterpret_cast<int *>(base + off);
    off += align256(n * sizeof(int));
  };
  float *df = nullptr;
  int *di = nullptr;
  const size_t NH = (size_t)N * hidden;
  const size_t part = std::max<size_t>((size_t)8 * hidden * hidden, (size_t)64 * hidden);
  takef(w ? &w->deg : &df, N);
  takef(w ? &w->wn_dst : &df, E);
  takef(w ? &w->wn_src : &df, E);
  takef(w ? &w->AX : &df, (size_t)N * 8);
  takef(w ? &w->H1 : &df, NH);
  takef(w ? &w->AH1 : &df, NH);
  takef(w ? &w->H2 : &df, NH);
  takef(w ? &w->T0 : &df, NH);
  takef(w ? &w->T1 : &df, NH);
  takef(w ? &w->part : &df, part);
  if (w) w->part_floats = part;
  takei(w ? &w->cnt_dst : &di, N + 1);
  takei(w ? &w->cnt_src : &di, N + 1);
  takei(w ? &w->ptr_dst : &di, N + 1);
  takei(w ? &w->ptr_src : &di, N + 1);
  takei(w ? &w->cur_dst : &di, N + 1);
  takei(w ? &w->cur_src : &di, N + 1);
  takei(w ? &w->eid_dst : &di, E);
  takei(w ? &w->eid_src : &di, E);
  takei(w ? &w->nbr_dst : &di, E);
  takei(w ? &w->nbr_src : &di, E);
  (void)out_dim;
  return off;
}

template <bool TA, bool TB, int EPI>
void gemm(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, const float *bias,
          const float *mask, int splits) {
  const int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, (K + kps - 1) / kps);
  hipLaunchKernelGGL((k_gemm<TA, TB, EPI>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps);
}

// weight-gradient GEMM  C[M x N] = A^T B with K = #nodes: split-K + deterministic reduce
void gemm_tn_splitk(hipStream_t st, const GcnWs &w, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C) {
  int splits = (int)std::min<size_t>(8, w.part_floats / ((size_t)M * N));
  splits = std::max(1, std::min(splits, (K + 2047) / 2048));
  const int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  const int S = (K + kps - 1) / kps;
  if (S == 1) {
    gemm<true, false, 0>(st, M, N, K, A, lda, B, ldb, C, N, nullptr, nullptr, 1);
    return;
  }
  gemm<true, false, 0>(st, M, N, K, A, lda, B, ldb, w.part, N, nullptr, nullptr, S);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((M * N + 255) / 256), dim3(256), 0, st, M * N, S, w.part, C);
}

void colsum(hipStream_t st, const GcnWs &w, int N, int C, const float *X, float *out) {
  const int nb = 64, rpb = (N + nb - 1) / nb;
  hipLaunchKernelGGL(k_colsum_part, dim3((C + 255) / 256, nb), dim3(256), 0, st, N, C, X, w.part, rpb);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((C + 255) / 256), dim3(256), 0, st, C, nb, w.part, out);
}

void build_graph(hipStream_t st, const GcnWs &w, int N, int E, const int64_t *ei, const float *ew) {
  hipMemsetAsync(w.cnt_dst, 0, (size_t)(N + 1) * 4, st);
  hipMemsetAsync(w.cnt_src, 0, (size_t)(N + 1) * 4, st);
  hipMemsetAsync(w.cur_dst, 0, (size_t)(N + 1) * 4, st);
  hipMemsetAsync(w.cur_src, 0, (size_t)(N + 1) * 4, st);
  if (E > 0) hipLaunchKernelGGL(k_degree, dim3((E + 255) / 256), dim3(256), 0, st, E, ei, w.cnt_dst, w.cnt_src);
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(1024), 0, st, N, w.cnt_dst, w.ptr_dst, w.cnt_src, w.ptr_src);
  if (E > 0) {
    hipLaunchKernelGGL(k_csr_fill, dim3((E + 255) / 256), dim3(256), 0, st, E, ei, w.ptr_dst, w.cur_dst, w.eid_dst, w.ptr_src,
                       w.cur_src, w.eid_src);
  }
  const dim3 gn((N + 127) / 128), bn(128);
  hipLaunchKernelGGL(k_csr_sort, gn, bn, 0, st, N, w.ptr_dst, w.eid_dst);
  hipLaunchKernelGGL(k_csr_sort, gn, bn, 0, st, N, w.ptr_src, w.eid_src);
  hipLaunchKernelGGL(k_degree_sum, gn, bn, 0, st, N, ew, w.ptr_src, w.eid_src, w.deg);
  hipLaunchKernelGGL(k_csr_finish, gn, bn, 0, st, N, E, ei, ew, w.deg, w.ptr_dst, w.eid_dst, w.nbr_dst, w.wn_dst, 1);
  hipLaunchKernelGGL(k_csr_finish, gn, bn, 0, st, N, E, ei, ew, w.deg, w.ptr_src, w.eid_src, w.nbr_src, w.wn_src, 0);
}

}  // namespace

extern "C" {

size_t drlgx_gcn_workspace_bytes(int n_nodes, int n_edges, int hidden, int out_dim) {
  if (n_nodes <= 0 || n_edges < 0 || hidden <= 0 || out_dim <= 0) return 0;
  return carve(nullptr, nullptr, n_nodes, std::max(n_edges, 1), hidden, out_dim) + 256;
}

int drlgx_gcn_forward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                      const int64_t *edge_index, const float *edge_attr, const float *W1, const float *b1, const float *W2,
                      const float *b2, const float *Wf, const float *bf, const float *dropout_mask, float *out, void *ws_dev) {
  if (n_nodes <= 0 || n_edges < 0 || in_dim <= 0 || in_dim > 8 || hidden <= 0 || (hidden & 3) || out_dim <= 0 || !x || !W1 || !b1 ||
      !W2 || !b2 || !Wf || !bf || !out || !ws_dev || (n_edges > 0 && (!edge_index || !edge_attr)))
    return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  GcnWs w;
  carve(&w, reinterpret_cast<char *>(ws_dev), n_nodes, std::max(n_edges, 1), hidden, out_dim);
  build_graph(st, w, n_nodes, n_edges, edge_index, edge_attr);
  hipLaunchKernelGGL(k_layer1, dim3(n_nodes), dim3(256), 0, st, n_nodes, in_dim, hidden, x, w.deg, w.ptr_dst, w.nbr_dst, w.wn_dst, W1,
                     b1, w.AX, w.H1);
  hipLaunchKernelGGL(k_aggregate, dim3(n_nodes), dim3(256), 0, st, n_nodes, hidden, w.H1, w.deg, w.ptr_dst, w.nbr_dst, w.wn_dst,
                     (const float *)nullptr, w.AH1);
  // H2 = relu(AH1 W2 + b2) * mask   (fp32 MFMA, fused epilogue)
  gemm<false, false, 1>(st, n_nodes, hidden, hidden, w.AH1, hidden, W2, hidden, w.H2, hidden, b2, dropout_mask, 1);
  hipLaunchKernelGGL(k_linear_out, dim3((n_nodes + 3) / 4), dim3(256), 0, st, n_nodes, hidden, out_dim, w.H2, Wf, bf, out);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_gcn_backward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                       const int64_t *edge_index, const float *edge_attr, const float *W1, const float *W2, const float *Wf,
                       const float *dropout_mask, const float *d_out, float *dW1, float *db1, float *dW2, float *db2, float *dWf,
                       float *dbf, void *ws_dev) {
  if (n_nodes <= 0 || in_dim <= 0 || in_dim > 8 || hidden <= 0 || (hidden & 3) || out_dim <= 0 || !d_out || !dW1 || !db1 || !dW2 ||
      !db2 || !dWf || !dbf || !ws_dev || !W2 || !Wf)
    return DRLGX_E_INVALID;
  (void)x; (void)edge_index; (void)edge_attr; (void)W1; (void)n_edges;  // the forward left AX/H1/AH1/H2 and both CSRs in ws
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  GcnWs w;
  carve(&w, reinterpret_cast<char *>(ws_dev), n_nodes, std::max(n_edges, 1), hidden, out_dim);
  // output layer
  gemm_tn_splitk(st, w, out_dim, hidden, n_nodes, d_out, out_dim, w.H2, hidden, dWf);  // dWf = dOut^T H2m
  colsum(st, w, n_nodes, out_dim, d_out, dbf);
  hipLaunchKernelGGL(k_dz2, dim3(n_nodes), dim3(256), 0, st, n_nodes, hidden, out_dim, d_out, Wf, dropout_mask, w.H2, w.T0);  // T0 = dZ2
  // layer 2
  gemm_tn_splitk(st, w, hidden, hidden, n_nodes, w.AH1, hidden, w.T0, hidden, dW2);  // dW2 = AH1^T dZ2
  colsum(st, w, n_nodes, hidden, w.T0, db2);
  gemm<false, true, 0>(st, n_nodes, hidden, hidden, w.T0, hidden, W2, hidden, w.T1, hidden, nullptr, nullptr, 1);  // T1 = dZ2 W2^T
  // dZ1 = (Â^T dAH1) * (H1 > 0)   -> T0
  hipLaunchKernelGGL(k_aggregate, dim3(n_nodes), dim3(256), 0, st, n_nodes, hidden, w.T1, w.deg, w.ptr_src, w.nbr_src, w.wn_src, w.H1,
                     w.T0);
  // layer 1
  gemm_tn_splitk(st, w, 8, hidden, n_nodes, w.AX, 8, w.T0, hidden, w.T1);  // [8 x hidden], rows >= in_dim are zero
  hipMemcpyAsync(dW1, w.T1, (size_t)in_dim * hidden * sizeof(float), hipMemcpyDeviceToDevice, st);
  colsum(st, w, n_nodes, hidden, w.T0, db1);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

}  // extern "C"

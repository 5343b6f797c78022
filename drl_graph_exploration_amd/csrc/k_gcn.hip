// GCN policy network on gfx950: forward and backward of
//   H1 = relu(Â X W1 + b1),  H2 = relu(Â H1 W2 + b2) [* dropout mask],  out = H2 Wf^T + bf
// with Â = D^-1/2 (A_w + 2I) D^-1/2, D = rowsum(A_w + 2I)   (PyG 1.x GCNConv(improved=True), SURVEY.md App. B;
// scripts/Networks.py:12-70 GCN / PolicyGCN / ValueGCN trunks; scripts/policy.py:234-253 for the backward).
//
// Layout / kernels
//   * the batch is irregular (one graph per env): edges are turned into two CSRs (by destination for the
//     forward aggregation, by source for the transposed one) with deterministic per-row order - per graph in ONE launch
//     when the caller knows the batch's graph boundaries (k_csr_graphs: the graph's edges sorted in LDS), else by the
//     generic count / scan / fill / sort / finish sequence;
//   * layer 1 (K = 5) is never materialised: AX = Â X is 8 floats per node (k_ax), and the aggregation of layer 2
//     recomputes the rows of H1 = relu(AX W1 + b1) it gathers (k_aggregate_l1; the backward pass recomputes the ReLU gate);
//   * aggregation (Â H): float4 lanes across the 1000 features, neighbour rows gathered with coalesced 4 KB reads;
//   * the dense 1000x1000 contractions run on the fp32 matrix cores (exact fp32, 157 TFLOP/s peak) with a fused
//     bias+ReLU(+mask) epilogue: batches large enough to fill the chip with (96..160) x 128 tiles on k_gemm_wide (8 waves,
//     v_mfma_f32_16x16x4_f32, tile height picked per launch), smaller ones on the 64x64 kernels (4 waves of one 32x32 MFMA
//     tile); operand tiles global -> LDS directly in both; the weight-gradient products (K = #nodes) use split-K with a
//     deterministic second-stage reduction.
// fp32 throughout (the reference trains in fp32); (Â X) W1 is used instead of Â (X W1) — same value up to fp32
// rounding (tests: <= 2e-5 relative against the plain-torch reference).
#include <algorithm>
#include <cstdlib>

#include "drlgx_dev.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// graph normalisation + CSR
// ------------------------------------------------------------------------------------------------
// in/out degree counts for the two CSRs (the weighted degree is summed later in edge order: float atomics here would
// make deg - and through the ReLU gates the whole forward/backward - depend on the arrival order)
// An explicit self loop keeps its weight as the node's self term (PyG add_remaining_self_loops: only the REMAINING self
// loops get the fill value 2): selfw[n] is preset to 2 and overwritten here.  Edges with an endpoint outside [0, N) are
// ignored (the C ABI has no status word for the GCN calls).
__global__ void k_degree(int N, int E, const int64_t *ei, const float *ew, int *cnt_dst, int *cnt_src, float *selfw) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t r64 = ei[e], c64 = ei[(size_t)E + e];
  if (r64 < 0 || r64 >= N || c64 < 0 || c64 >= N) return;
  const int r = (int)r64, c = (int)c64;
  if (r == c) {
    selfw[r] = ew[e];
    return;
  }
  atomicAdd(&cnt_dst[c], 1);
  atomicAdd(&cnt_src[r], 1);
}
// exclusive scan of two count arrays (single block of 1024 threads; N is a few 10^4): per-thread chunk sums, a
// shuffle scan inside each wave, a shuffle scan of the 16 wave totals, then the chunks are rewritten
__global__ __launch_bounds__(1024) void k_scan2(int n, const int *a, int *pa, const int *b, int *pb) {
  __shared__ int wtot[2][16];
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6;
  const int chunk = (n + nt - 1) / nt;
  const int i0 = t * chunk, i1 = min(n, i0 + chunk);
  int sa = 0, sb = 0;
  for (int i = i0; i < i1; ++i) {
    sa += a[i];
    sb += b[i];
  }
  int xa = sa, xb = sb;  // inclusive scan over the wave
  for (int off = 1; off < 64; off <<= 1) {
    const int ya = __shfl_up(xa, off), yb = __shfl_up(xb, off);
    if (lane >= off) {
      xa += ya;
      xb += yb;
    }
  }
  if (lane == 63) {
    wtot[0][wave] = xa;
    wtot[1][wave] = xb;
  }
  __syncthreads();
  if (wave == 0) {
    int va = lane < 16 ? wtot[0][lane] : 0, vb = lane < 16 ? wtot[1][lane] : 0;
    const int ia = va, ib = vb;
    for (int off = 1; off < 16; off <<= 1) {
      const int ya = __shfl_up(va, off), yb = __shfl_up(vb, off);
      if (lane >= off) {
        va += ya;
        vb += yb;
      }
    }
    if (lane < 16) {
      wtot[0][lane] = va - ia;  // exclusive
      wtot[1][lane] = vb - ib;
    }
    if (lane == 15) {
      pa[n] = va;
      pb[n] = vb;
    }
  }
  __syncthreads();
  sa = wtot[0][wave] + xa - sa;  // exclusive prefix of this thread's chunk
  sb = wtot[1][wave] + xb - sb;
  for (int i = i0; i < i1; ++i) {
    pa[i] = sa;
    sa += a[i];
    pb[i] = sb;
    sb += b[i];
  }
}
__global__ void k_csr_fill(int N, int E, const int64_t *ei, const int *ptr_dst, int *cur_dst, int *eid_dst, const int *ptr_src,
                           int *cur_src, int *eid_src) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t r64 = ei[e], c64 = ei[(size_t)E + e];
  if (r64 < 0 || r64 >= N || c64 < 0 || c64 >= N || r64 == c64) return;
  const int r = (int)r64, c = (int)c64;
  eid_dst[ptr_dst[c] + atomicAdd(&cur_dst[c], 1)] = e;
  eid_src[ptr_src[r] + atomicAdd(&cur_src[r], 1)] = e;
}
// sort each CSR row by edge id (rows are short) -> deterministic summation order; both CSRs in one launch
__global__ void k_csr_sort(int N, const int *ptr0, int *eid0, const int *ptr1, int *eid1) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= 2 * N) return;
  const int *ptr = n < N ? ptr0 : ptr1;
  int *eid = n < N ? eid0 : eid1;
  if (n >= N) n -= N;
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
// deg[row] = sum of the row's edge weights in edge order, then the self loop weight (2 from
// add_remaining_self_loops(fill_value = 2), or the explicit self loop's own)  (PyG: scatter_add(edge_weight, row), row = source)
__global__ void k_degree_sum(int N, const float *ew, const int *ptr_src, const int *eid_src, const float *selfw, float *deg) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int i = ptr_src[n]; i < ptr_src[n + 1]; ++i) s += ew[eid_src[i]];
  deg[n] = s + selfw[n];
}
// resolve (neighbour, normalised weight) per CSR slot, for the by-destination CSR (threads < N) and the by-source one.
// dis = deg^-1/2 (inf -> 0).
__global__ void k_csr_finish(int N, int E, const int64_t *ei, const float *ew, const float *deg, const int *ptr_dst,
                             const int *eid_dst, int *nbr_dst, float *wn_dst, const int *ptr_src, const int *eid_src, int *nbr_src,
                             float *wn_src, int *end_dst, int *end_src) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= 2 * N) return;
  const bool by_dst = n < N;
  const int *ptr = by_dst ? ptr_dst : ptr_src, *eid = by_dst ? eid_dst : eid_src;
  int *nbr = by_dst ? nbr_dst : nbr_src;
  float *wn = by_dst ? wn_dst : wn_src;
  if (!by_dst) n -= N;
  (by_dst ? end_dst : end_src)[n] = ptr[n + 1];
  for (int i = ptr[n]; i < ptr[n + 1]; ++i) {
    const int e = eid[i];
    const int r = (int)ei[e], c = (int)ei[(size_t)E + e];
    float dr = deg[r] > 0 ? 1.0f / sqrtf(deg[r]) : 0.0f, dc = deg[c] > 0 ? 1.0f / sqrtf(deg[c]) : 0.0f;
    nbr[i] = by_dst ? r : c;
    wn[i] = dr * ew[e] * dc;  // deg^-1/2[row] * w * deg^-1/2[col]
  }
}

// ------------------------------------------------------------------------------------------------
// Both CSRs, the degrees and the normalised weights of a BATCH of small graphs in one launch: one workgroup per graph
// (a PyG batch / drlgx_graph export: graph g owns nodes [node_off[g], node_off[g+1]) and edges [edge_off[g],
// edge_off[g+1]), every edge connects two of its nodes).  The graph's edges are sorted in LDS by (source node, edge id)
// and by (destination node, edge id) - one 32-bit key each, two bitonic sorts run stage by stage together -, which IS
// the CSR order of the generic build (rows by node, entries in edge order): the same rows, bit for bit, without its
// eight launches (k_degree .. k_csr_finish are dominated by launch latency at these sizes).
// Row r of graph g starts at edge_off[g] + (position of its first key): rows are contiguous inside a graph; self loops
// and ignored edges sort to the end and leave unused slots there, hence explicit row ends.
// ------------------------------------------------------------------------------------------------
constexpr int kCsrKeyShift = 14;               // key = local node << 14 | local edge id
constexpr int kCsrMaxEdges = 1 << kCsrKeyShift;  // per graph (16 384; 2 x 64 KB of keys in LDS at that size)
constexpr uint32_t kCsrNoKey = 0xffffffffu;

__global__ __launch_bounds__(256) void k_csr_graphs(int N, int E, int P2, int extra, const int64_t *ei, const float *ew, const int *node_off,
                                                    const int *edge_off, float *deg, float *selfw_out, int *ptr_dst, int *end_dst,
                                                    int *nbr_dst, float *wn_dst, int *ptr_src, int *end_src, int *nbr_src, float *wn_src,
                                                    const float *x, int in_dim, float *AX, int local) {
  // local (the replay pool's per-graph cache, drlgx_replay_cache_csr): row starts / ends and neighbour ids are stored relative
  // to the graph's first edge / node, so that a later collation only adds the graph's offsets in the mini-batch
  extern __shared__ uint32_t s_keys[];  // [2][P2]: by source, by destination; then (extra) the weights and packed endpoints
  uint32_t *ks = s_keys, *kd = s_keys + P2;
  float *s_w = reinterpret_cast<float *>(s_keys + 2 * (size_t)P2);
  uint32_t *s_pk = s_keys + 3 * (size_t)P2;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n0 = node_off[g], n1 = node_off[g + 1], e0 = edge_off[g];
  const int ng = n1 - n0, eg = min(edge_off[g + 1] - e0, P2);  // (the caller promised eg <= P2)
  const int eb = local ? 0 : e0, nb = local ? 0 : n0;  // what stored positions / ids are relative to
  const bool ext = extra && ng <= 65535;  // the edges' weights and local endpoints stay in LDS: the later passes read no edge from memory
  for (int m = tid; m < ng; m += 256) selfw_out[n0 + m] = 2.0f;  // add_remaining_self_loops(fill_value = 2)
  __syncthreads();
  for (int j = tid; j < P2; j += 256) {
    uint32_t a = kCsrNoKey, b = kCsrNoKey;
    if (j < eg) {
      const int64_t r = ei[e0 + j], c = ei[(size_t)E + e0 + j];
      const float wj = ew[e0 + j];
      if (ext) s_w[j] = wj;
      if (r >= n0 && r < n1 && c >= n0 && c < n1) {  // an edge with an endpoint outside the graph is ignored
        if (ext) s_pk[j] = (uint32_t)(r - n0) | ((uint32_t)(c - n0) << 16);
        if (r == c) {
          selfw_out[r] = wj;  // an explicit self loop keeps its weight as the node's self term
        } else {
          a = ((uint32_t)(r - n0) << kCsrKeyShift) | (uint32_t)j;
          b = ((uint32_t)(c - n0) << kCsrKeyShift) | (uint32_t)j;
        }
      }
    }
    ks[j] = a;
    kd[j] = b;
  }
  __syncthreads();
  // bitonic sort, ascending, both key arrays in the same stages.  Wave w owns the contiguous segment of P2 / 4 keys
  // [w P2/4, (w+1) P2/4): a compare-exchange at distance j < P2/4 stays inside the segment, and a wave's LDS operations
  // execute in order, so only the stages that cross segments (three of the 55 at P2 = 1024) take workgroup barriers
  {
    const int seg = P2 >> 2, wave = tid >> 6, lane = tid & 63;
    auto exchange = [&](int t, int k, int j) {
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), q = i | j;
      const bool up = (i & k) == 0;
      const uint32_t a0 = ks[i], a1 = ks[q], b0 = kd[i], b1 = kd[q];
      const uint32_t alo = min(a0, a1), ahi = max(a0, a1), blo = min(b0, b1), bhi = max(b0, b1);
      ks[i] = up ? alo : ahi;  // (unconditional stores: no divergent branches in the 55 stages)
      ks[q] = up ? ahi : alo;
      kd[i] = up ? blo : bhi;
      kd[q] = up ? bhi : blo;
    };
    for (int k = 2; k <= P2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        if (seg >= 64 && j < seg) {  // inside the segments: this wave's seg / 2 pairs, wave-level ordering only
          for (int u = lane; u < (seg >> 1); u += 64) exchange(wave * (seg >> 1) + u, k, j);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {                      // across segments (or a tiny sort): other waves' results in, other waves' operands out
          __syncthreads();
          for (int t = tid; t < (P2 >> 1); t += 256) exchange(t, k, j);
          __syncthreads();
        }
      }
    __syncthreads();
  }
  // rows = key ranges; weighted degree = the by-source row summed in edge order, then the self term
  auto lower = [&](const uint32_t *keys, uint32_t v) {
    int lo = 0, hi = P2;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  for (int m = tid; m < ng; m += 256) {
    const uint32_t v0 = (uint32_t)m << kCsrKeyShift, v1 = (uint32_t)(m + 1) << kCsrKeyShift;
    const int s0 = lower(ks, v0), s1 = lower(ks, v1), d0 = lower(kd, v0), d1 = lower(kd, v1);
    const int n = n0 + m;
    ptr_src[n] = eb + s0;
    end_src[n] = eb + s1;
    ptr_dst[n] = eb + d0;
    end_dst[n] = eb + d1;
    float dsum = 0.f;
    for (int i = s0; i < s1; ++i) {
      const int j = (int)(ks[i] & (kCsrMaxEdges - 1));
      dsum += ext ? s_w[j] : ew[e0 + j];
    }
    deg[n] = dsum + selfw_out[n];
  }
  __threadfence_block();  // deg[] of the whole graph is read below
  __syncthreads();
  // entries: (neighbour, deg^-1/2[row] w deg^-1/2[col]), one thread per sorted position
  for (int i = tid; i < eg; i += 256) {
    const uint32_t a = ks[i], b = kd[i];
    if (a != kCsrNoKey) {
      const int j = (int)(a & (kCsrMaxEdges - 1));
      const int r = n0 + (int)(a >> kCsrKeyShift), c = ext ? n0 + (int)(s_pk[j] >> 16) : (int)ei[(size_t)E + e0 + j];
      const float dr = deg[r] > 0 ? 1.0f / sqrtf(deg[r]) : 0.0f, dc = deg[c] > 0 ? 1.0f / sqrtf(deg[c]) : 0.0f;
      nbr_src[e0 + i] = c - n0 + nb;
      wn_src[e0 + i] = dr * (ext ? s_w[j] : ew[e0 + j]) * dc;
    }
    if (b != kCsrNoKey) {
      const int j = (int)(b & (kCsrMaxEdges - 1));
      const int c = n0 + (int)(b >> kCsrKeyShift), r = ext ? n0 + (int)(s_pk[j] & 0xffffu) : (int)ei[e0 + j];
      const float dr = deg[r] > 0 ? 1.0f / sqrtf(deg[r]) : 0.0f, dc = deg[c] > 0 ? 1.0f / sqrtf(deg[c]) : 0.0f;
      nbr_dst[e0 + i] = r - n0 + nb;
      wn_dst[e0 + i] = dr * (ext ? s_w[j] : ew[e0 + j]) * dc;
    }
  }
  if (!AX) return;
  // AX = Â X of the graph's own nodes (k_ax's expression and order), from the rows this workgroup has just written
  __threadfence_block();
  __syncthreads();
  for (int e = tid; e < ng * 8; e += 256) {
    const int n = n0 + (e >> 3), t = e & 7;
    float s = 0.f;
    if (t < in_dim) {
      s = (selfw_out[n] / deg[n]) * x[(size_t)n * in_dim + t];
      for (int i = ptr_dst[n] + (e0 - eb); i < end_dst[n] + (e0 - eb); ++i) s += wn_dst[i] * x[(size_t)(nbr_dst[i] + (n0 - nb)) * in_dim + t];
    }
    AX[(size_t)n * 8 + t] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Mini-batch collation of replay graphs whose normalisation, CSRs and ÂX were cached per graph when their export entered the
// pool (k_csr_graphs in `local` mode): graph g of the mini-batch (desc int64 [5][G] = node_start, node_cnt, edge_start,
// edge_cnt, loc, as k_replay_collate) is copied into the GCN workspace's arrays at its cumulative node / edge offsets, row
// starts / ends shifted by the edge offset, neighbour ids by the node offset - what build_graph_batched + the ÂX pass would
// have produced for the collated batch, bit for bit (the per-graph sort order does not depend on where the graph sits).
// Blocks [G, 2G): the second list's cached per-node value only (the target read-out over the next states), as
// k_replay_collate's pair form.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_csr_collate(int G, const int64_t *desc, drlgx_csr_cache c, float *deg, float *selfw, float *AX,
                                                     int *ptr_dst, int *end_dst, int *ptr_src, int *end_src, int *nbr_dst, int *nbr_src,
                                                     float *wn_dst, float *wn_src, int *node_off_out, int *edge_off_out,
                                                     const int64_t *desc2, const float *pool_q, float *q2_out) {
  __shared__ long long red[2][4];
  const int tid = threadIdx.x;
  int g = blockIdx.x;
  const bool second = g >= G;
  if (second) {
    g -= G;
    desc = desc2;
  }
  long long sn = 0, se = 0;
  for (int j = tid; j < g; j += 256) {
    sn += desc[(size_t)G + j];
    se += desc[3 * (size_t)G + j];
  }
  for (int o = 32; o > 0; o >>= 1) {
    sn += __shfl_down(sn, o);
    se += __shfl_down(se, o);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = sn;
    red[1][tid >> 6] = se;
  }
  __syncthreads();
  const long long node_off = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const long long edge_off = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const long long n0 = desc[g], nn = desc[(size_t)G + g], e0 = desc[2 * (size_t)G + g], ne = desc[3 * (size_t)G + g];
  if (second) {
    for (long long i = tid; i < nn; i += 256) q2_out[node_off + i] = pool_q[n0 + i];
    return;
  }
  if (tid == 0) {
    node_off_out[g] = (int)node_off;
    edge_off_out[g] = (int)edge_off;
    if (g == G - 1) {
      node_off_out[G] = (int)(node_off + nn);
      edge_off_out[G] = (int)(edge_off + ne);
    }
  }
  const int eo = (int)edge_off, no = (int)node_off;
  for (long long i = tid; i < nn; i += 256) {
    deg[node_off + i] = c.deg[n0 + i];
    selfw[node_off + i] = c.selfw[n0 + i];
    ptr_dst[node_off + i] = c.ptr_dst[n0 + i] + eo;
    end_dst[node_off + i] = c.end_dst[n0 + i] + eo;
    ptr_src[node_off + i] = c.ptr_src[n0 + i] + eo;
    end_src[node_off + i] = c.end_src[n0 + i] + eo;
  }
  {
    const float4 *s4 = reinterpret_cast<const float4 *>(c.ax + n0 * 8);
    float4 *d4 = reinterpret_cast<float4 *>(AX + node_off * 8);
    for (long long i = tid; i < nn * 2; i += 256) d4[i] = s4[i];
  }
  for (long long j = tid; j < ne; j += 256) {
    nbr_dst[edge_off + j] = c.nbr_dst[e0 + j] + no;
    nbr_src[edge_off + j] = c.nbr_src[e0 + j] + no;
    wn_dst[edge_off + j] = c.wn_dst[e0 + j];
    wn_src[edge_off + j] = c.wn_src[e0 + j];
  }
}

// ------------------------------------------------------------------------------------------------
// layer 1 and the aggregation of layer 2, without materialising H1:
//   AX = Â X (in_dim <= 8 features, rows padded to 8)                                     k_ax, one thread per (node, k)
//   AH1[n] = sum_i w_i relu(AX[m_i] W1 + b1)  over n itself (self weight) and its neighbours   k_aggregate_l1
// A row of H1 = relu(AX[m] W1 + b1) costs in_dim FMAs per element from 32 bytes of AX, against 4 KB of HBM / L2 traffic to
// write it once and gather it ~8 times: it is recomputed where it is needed (also as the ReLU gate of the backward pass),
// in the same operation order as a stored H1 would have had.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ax(int N, int in_dim, const float *x, const float *deg, const float *selfw, const int *ptr,
                                            const int *pend, const int *nbr, const float *wn, float *AX) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int n = e >> 3, t = e & 7;
  if (n >= N) return;
  float s = 0.f;
  if (t < in_dim) {
    s = (selfw[n] / deg[n]) * x[(size_t)n * in_dim + t];  // self loop: dis * w_self * dis
    for (int i = ptr[n]; i < pend[n]; ++i) s += wn[i] * x[(size_t)nbr[i] * in_dim + t];
  }
  AX[(size_t)n * 8 + t] = s;
}

// H1[m][4c .. 4c+3] from AX[m] (8 floats), this thread's four W1 columns (w[k]) and biases: one FMA chain over k per column,
// written on two-float vectors so that it compiles to v_pk_fma_f32 (two columns per instruction, a[k] broadcast).  Left to
// itself the compiler packs along k instead - v_pk_mul_f32 + two v_add_f32 per pair of products, 28 VALU instructions per
// (row, four columns) where 10 + 4 (ReLU) + 2 (weighted sum) do.
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <int IN>
__device__ __forceinline__ void h1_core(floatx2 &lo, floatx2 &hi, const float (&a)[8], int in_dim, const float4 (&w)[8], const float4 &bias) {
  lo = floatx2{bias.x, bias.y};
  hi = floatx2{bias.z, bias.w};
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (IN > 0 ? k < IN : k < in_dim) {
      const floatx2 ak = {a[k], a[k]};
      lo = __builtin_elementwise_fma(ak, floatx2{w[k].x, w[k].y}, lo);
      hi = __builtin_elementwise_fma(ak, floatx2{w[k].z, w[k].w}, hi);
    }
}
template <int IN = 0>  // IN > 0: the number of input features at compile time (the reference's 5): straight-line code
__device__ __forceinline__ float4 h1_row(const float *AX, int m, int in_dim, const float4 (&w)[8], const float4 &bias) {
  const float4 a0 = reinterpret_cast<const float4 *>(AX + (size_t)m * 8)[0], a1 = reinterpret_cast<const float4 *>(AX + (size_t)m * 8)[1];
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  floatx2 lo, hi;
  h1_core<IN>(lo, hi, a, in_dim, w, bias);
  return make_float4(lo.x, lo.y, hi.x, hi.y);  // pre-activation
}

// H1 row from 8 staged AX values
template <int IN = 0>
__device__ __forceinline__ float4 h1_row_lds(const float *ax, int in_dim, const float4 (&w)[8], const float4 &bias) {
  const float4 a0 = reinterpret_cast<const float4 *>(ax)[0], a1 = reinterpret_cast<const float4 *>(ax)[1];
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  floatx2 lo, hi;
  h1_core<IN>(lo, hi, a, in_dim, w, bias);
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
// acc += wi * relu(u)
__device__ __forceinline__ void relu_axpy(floatx2 &alo, floatx2 &ahi, float wi, const float4 &u) {
  const floatx2 w2 = {wi, wi};
  alo = __builtin_elementwise_fma(w2, floatx2{fmaxf(u.x, 0.f), fmaxf(u.y, 0.f)}, alo);
  ahi = __builtin_elementwise_fma(w2, floatx2{fmaxf(u.z, 0.f), fmaxf(u.w, 0.f)}, ahi);
}

constexpr int kAggStage = 64;  // neighbour rows of AX (and their weights) staged in LDS; longer rows read the rest from memory
constexpr int kAggNodes = 4;   // nodes per workgroup: the thread's W1 columns and biases are loaded once for all of them, and
                               // the neighbour lists of all of them are staged together (one round of memory latency)
static_assert(kAggNodes * kAggStage == 256, "one staging thread per (node, neighbour slot)");
// Every memory access of the kernel sits in ONE dependent chain of three loads (row bounds -> neighbour id -> its AX row),
// walked once by every thread for its own (node, slot) with the W1 columns requested in front of it; the multiply loop
// reads LDS only.  (The first version staged 8 elements per thread in a loop - 24 dependent round trips - and loaded W1
// and the row bounds behind one wait each: 57 us for the 17 288-node batch, a fifth of the VALU rate.)
template <int IN>
__global__ __launch_bounds__(256) void k_aggregate_l1(int N, int in_dim, int hidden, const float *AX, const float *W1, const float *b1,
                                                      const float *deg, const float *selfw, const int *ptr, const int *pend, const int *nbr,
                                                      const float *wn, float *out, float *b1_keep) {
  __shared__ __attribute__((aligned(16))) float s_ax[kAggNodes][(kAggStage + 1) * 8];  // slot kAggStage: the node's own row
  __shared__ float s_wn[kAggNodes][kAggStage + 1];                                       // slot kAggStage: its self weight
  __shared__ int s_ab[kAggNodes][2];
  const int tid = threadIdx.x;
  const int h4 = hidden >> 2;
  const int nb0 = blockIdx.x * kAggNodes, nn = min(kAggNodes, N - nb0);
  auto load_w = [&](float4 (&w)[8], float4 &bias, int c) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      w[k] = (IN > 0 ? k < IN : k < in_dim) ? reinterpret_cast<const float4 *>(W1 + (size_t)k * hidden)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    bias = reinterpret_cast<const float4 *>(b1)[c];
  };
  float4 w[8], bias;
  load_w(w, bias, min(tid, h4 - 1));  // (unconditional: under a branch the compiler waits for the loads before leaving it)
  {
    // every load unconditional (clamped to a valid element), only the LDS stores are predicated: under branches the loads
    // of the own row / self weight / neighbour row would each be waited for in turn
    const int q = tid >> 6, j = tid & (kAggStage - 1), n = min(nb0 + q, N - 1);
    const float4 *AX4 = reinterpret_cast<const float4 *>(AX);
    const int a = ptr[n], b = pend[n];
    const float4 own = AX4[(size_t)n * 2 + (j & 1)];
    const float sw = selfw[n], dg = deg[n];
    const int e = max(min(a + j, b - 1), 0);
    const int m = a + j < b ? nbr[e] : n;  // (outside the row nbr[e] may be a never-written gap of the batched CSR: not an address)
    const float wv = wn[e];
    const float4 r0 = AX4[(size_t)m * 2], r1 = AX4[(size_t)m * 2 + 1];
    if (q < nn) {
      if (j < 2) reinterpret_cast<float4 *>(s_ax[q] + 8 * kAggStage)[j] = own;
      if (j == 2) {
        s_wn[q][kAggStage] = sw / dg;
        s_ab[q][0] = a;
        s_ab[q][1] = b;
      }
      if (a + j < b) {
        s_wn[q][j] = wv;
        reinterpret_cast<float4 *>(s_ax[q] + 8 * j)[0] = r0;
        reinterpret_cast<float4 *>(s_ax[q] + 8 * j)[1] = r1;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < h4; c += 256) {
    if (c != tid) load_w(w, bias, c);
    if (blockIdx.x == 0) reinterpret_cast<float4 *>(b1_keep)[c] = bias;  // the backward pass's copy of b1 (the ReLU gate of layer 1)
    for (int q = 0; q < nn; ++q) {
      const int n = nb0 + q;
      const int a = s_ab[q][0], b = s_ab[q][1];
      const int ns = min(b - a, kAggStage);
      const float self = s_wn[q][kAggStage];
      const float4 v = h1_row_lds<IN>(s_ax[q] + 8 * kAggStage, in_dim, w, bias);
      floatx2 alo = {self * fmaxf(v.x, 0.f), self * fmaxf(v.y, 0.f)}, ahi = {self * fmaxf(v.z, 0.f), self * fmaxf(v.w, 0.f)};
      for (int j = 0; j < ns; ++j) relu_axpy(alo, ahi, s_wn[q][j], h1_row_lds<IN>(s_ax[q] + 8 * j, in_dim, w, bias));
      for (int i = a + ns; i < b; ++i) relu_axpy(alo, ahi, wn[i], h1_row<IN>(AX, nbr[i], in_dim, w, bias));
      reinterpret_cast<float4 *>(out + (size_t)n * hidden)[c] = make_float4(alo.x, alo.y, ahi.x, ahi.y);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// aggregation out[n] = (selfw[n]/deg[n]) H[n] + sum_i wn[i] H[nbr[i]], optionally gated by H1[n] > 0 with H1 recomputed
// from AX / W1 / b1 (the ReLU of layer 1 in the backward pass);  one workgroup per node, float4 per lane
// ------------------------------------------------------------------------------------------------
template <bool kGate>
__global__ __launch_bounds__(256) void k_aggregate(int N, int hidden, const float *H, const float *deg, const float *selfw, const int *ptr,
                                                   const int *pend, const int *nbr, const float *wn, int in_dim, const float *AX,
                                                   const float *W1, const float *b1, float *out) {
  const int n = blockIdx.x;
  const int h4 = hidden >> 2;
  const float self = selfw[n] / deg[n];
  const int a = ptr[n], b = pend[n];
  for (int c = threadIdx.x; c < h4; c += 256) {
    float4 v = reinterpret_cast<const float4 *>(H + (size_t)n * hidden)[c];
    float4 acc = make_float4(self * v.x, self * v.y, self * v.z, self * v.w);
    for (int i = a; i < b; ++i) {
      const float w = wn[i];
      const float4 u = reinterpret_cast<const float4 *>(H + (size_t)nbr[i] * hidden)[c];
      acc.x += w * u.x; acc.y += w * u.y; acc.z += w * u.z; acc.w += w * u.w;
    }
    if (kGate) {
      float4 wk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) wk[k] = k < in_dim ? reinterpret_cast<const float4 *>(W1 + (size_t)k * hidden)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 g = h1_row(AX, n, in_dim, wk, reinterpret_cast<const float4 *>(b1)[c]);
      acc.x = g.x > 0.f ? acc.x : 0.f; acc.y = g.y > 0.f ? acc.y : 0.f;
      acc.z = g.z > 0.f ? acc.z : 0.f; acc.w = g.w > 0.f ? acc.w : 0.f;
    }
    reinterpret_cast<float4 *>(out + (size_t)n * hidden)[c] = acc;
  }
}

// workgroup id -> (row panel tm, column tile tn) of a launch of tiles_n * roundup8(tiles_m) workgroups (x) per K-slice (z); false: no tile.
// XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs; XCD x gets the row panels x, x + 8, ... and walks a
// panel's column tiles on consecutive slots (the panel of A stays in its L2).  With fewer row panels than XCDs - the read-out layer's
// weight gradient, M = out_dim rows: ONE panel - that order would put the whole launch on tiles_m XCDs (32 CUs each: 357 us for a
// 4.4 GFLOP product); such launches take the plain order, consecutive tiles on consecutive XCDs.
__device__ __forceinline__ bool tile_of_block(int tiles_m, int tiles_n, int &tm, int &tn) {
  if (tiles_m < 8) {
    tm = blockIdx.x % tiles_m;
    tn = blockIdx.x / tiles_m;
    return tn < tiles_n;
  }
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  tn = slot % tiles_n;
  tm = (slot / tiles_n) * 8 + xcd;
  return tm < tiles_m;
}

// EPI 3's factor at element `at` of the [M x ldc] matrices: the ReLU gate recovered from G = H2 (k_dz2) times the dropout mask
__device__ __forceinline__ float gate3(const float *G, const float *mask, size_t at) {
  float g = G[at] > 0.f ? 1.f : 0.f;
  if (mask) g *= mask[at];
  return g;
}

// ------------------------------------------------------------------------------------------------
// The same GEMM with the operand tiles loaded global -> LDS directly (global_load_lds_dwordx4, gfx950): no VGPR staging
// and no ds_write instructions, four LDS stages with three tiles in flight.  Bit-identical to k_gemm (same MFMA order);
// 6-10 % faster on the 17 287-row batches, equal on the 4 340-row ones (scripts/micro/gemm_dl_bench.hip).
// ------------------------------------------------------------------------------------------------
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
  int tm, tn;
  if (!tile_of_block(tiles_m, tiles_n, tm, tn)) return;
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
    const float bj = (EPI == 1 || EPI == 2) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
      if (row >= M) continue;
      float v = acc[r];
      if (EPI == 1) {
        v = fmaxf(v + bj, 0.f);
        if (mask) v *= mask[(size_t)row * ldc + col];
      } else if (EPI == 2) {
        v += bj;
      } else if (EPI == 3) {
        v *= gate3(bias, mask, (size_t)row * ldc + col);
      }
      Cz[(size_t)row * ldc + col] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM  C[M x N] = op(A) op(B)   (row-major; TA: A is stored [K x M]; TB: B is stored [N x K])
// block 128x128x16, 256 threads = 2x2 waves, each wave 2x2 tiles of v_mfma_f32_32x32x2_f32.
//  * global -> registers -> LDS with one 16-byte load/store per quarter tile row (scalar predicated loads only for
//    operands whose contiguous dimension is not a multiple of 4: the [nodes x out_dim] head gradient);
//  * an operand whose source is k-contiguous sits in LDS as [x][k] (stride 20 floats: ds_write_b128 straight from
//    the load, fragment = two conflict-free ds_read_b128); an x-contiguous one as [k][x] (stride 132, ds_read_b32);
//  * MFMA step s of a K-tile multiplies k = s (lanes 0-31) and k = 8 + s (lanes 32-63) - any pairing of the 16 k's is
//    a valid 32x32x2 schedule, and this one makes each lane's 8 fragment values contiguous in the [x][k] layout;
//  * all fragments of the K-tile are read up front, then 32 MFMAs issue back to back while the next tile's global
//    loads are in flight; one barrier per K-tile (double-buffered LDS);
//  * workgroup id -> tile mapping keeps the 8 column tiles of one row panel on one XCD (shared A panel in its L2).
// EPI 0: C = acc (split-K partial when gridDim.z > 1: C += z * M * N)
// EPI 1: C = relu(acc + bias[col]) * (mask ? mask[row][col] : 1)        (forward layer 2)
// EPI 2: C = acc + bias[col]                                            (read-out layer with more than 8 outputs: the critic's 100)
// EPI 3: C = acc * (G[row][col] > 0) * (mask ? mask[row][col] : 1), G = `bias` read as an [M x ldc] matrix
//        (dZ2 of that read-out layer: G = H2, see k_dz2)
// ------------------------------------------------------------------------------------------------
constexpr int BK = 16;
constexpr int LDK = BK + 4;  // [x][k] tile stride (floats)

// LDS footprint of one operand tile of XR * 64 rows/cols (either layout)
template <int XR>
struct TileF {
  static constexpr int ldx = XR * 64 + 4;  // [k][x] tile stride
  static constexpr int value = (XR * 64 * LDK > BK * ldx) ? XR * 64 * LDK : BK * ldx;
};

// quarter-tile loads of one operand tile (XR * 64 x 16): KC = source is k-contiguous (src[x * ld + k]), else
// src[k * ld + x]; 256 threads move XR float4 each
template <bool KC, bool VEC, int XR>
__device__ __forceinline__ void g_load(float4 (&reg)[XR], bool (&okr)[XR], const float *__restrict__ src, int ld, int x0, int X,
                                       int k0, int kend, int tid) {
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    const int x = KC ? x0 + (tid >> 2) + 64 * r : (XR == 2 ? x0 + (tid & 31) * 4 : x0 + (tid & 15) * 4);
    const int k = KC ? k0 + (tid & 3) * 4 : (XR == 2 ? k0 + (tid >> 5) + 8 * r : k0 + (tid >> 4));
    if (VEC) {
      // contract (host): the contiguous dimension, ld and the base address are multiples of 4 floats, so a float4
      // is all inside or all outside; outside ones load a clamped (valid) address unconditionally and are zeroed
      // when they are stored to LDS (so that nothing waits on the load before the MFMAs of the current tile)
      okr[r] = x < X && k < kend;
      const int xc = min(x, X - (KC ? 1 : 4)), kc = min(k, kend - (KC ? 4 : 1));
      const size_t at = KC ? (size_t)xc * ld + kc : (size_t)kc * ld + xc;
      reg[r] = *reinterpret_cast<const float4 *>(src + at);
    } else {
      okr[r] = true;
      const size_t at = KC ? (size_t)x * ld + k : (size_t)k * ld + x;
      float e[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = KC ? (x < X && k + c < kend) : (k < kend && x + c < X);
        e[c] = ok ? src[at + c] : 0.f;
      }
      reg[r] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}

template <bool KC, int XR>
__device__ __forceinline__ void s_store(float *T, const float4 (&reg)[XR], const bool (&okr)[XR], int tid) {
  constexpr int ldx = TileF<XR>::ldx;
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    float *p = KC ? T + ((tid >> 2) + 64 * r) * LDK + (tid & 3) * 4
                  : (XR == 2 ? T + ((tid >> 5) + 8 * r) * ldx + (tid & 31) * 4 : T + (tid >> 4) * ldx + (tid & 15) * 4);
    const float4 v = reg[r];
    const bool ok = okr[r];
    *reinterpret_cast<float4 *>(p) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
}

// the 8 values lane (li, h) feeds to MFMA steps 0..7 for tile rows/cols xb + li: k = 8 h + s
template <bool KC, int XR>
__device__ __forceinline__ void s_frag(float (&f)[8], const float *T, int xb, int lane) {
  constexpr int ldx = TileF<XR>::ldx;
  const int li = lane & 31, h = lane >> 5;
  if (KC) {
    const float4 u0 = *reinterpret_cast<const float4 *>(T + (xb + li) * LDK + 8 * h);
    const float4 u1 = *reinterpret_cast<const float4 *>(T + (xb + li) * LDK + 8 * h + 4);
    f[0] = u0.x; f[1] = u0.y; f[2] = u0.z; f[3] = u0.w;
    f[4] = u1.x; f[5] = u1.y; f[6] = u1.z; f[7] = u1.w;
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = T[(8 * h + s) * ldx + xb + li];
  }
}

// MI x NI = 32x32 sub-tiles per wave; the workgroup tile is (64 MI) x (64 NI)
template <bool TA, bool TB, int EPI, bool AV, bool BV, int MI, int NI>
__global__ __launch_bounds__(256) void k_gemm(
    int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
    const float *__restrict__ bias, const float *__restrict__ mask, int k_per_split, int tiles_m, int tiles_n) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  __shared__ __attribute__((aligned(16))) float As[3][TileF<MI>::value];
  __shared__ __attribute__((aligned(16))) float Bs[3][TileF<NI>::value];
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give XCD x the row panels
  // p = x, x + 8, ... and walk a panel's column tiles on consecutive slots of the same XCD
  int tm, tn;
  if (!tile_of_block(tiles_m, tiles_n, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // software pipeline, per K-tile t (one barrier each):
  //   MFMAs of tile t from fragment registers F[t & 1], and between them
  //     ds_write  staging registers (tile t+2, loaded during t-1)  -> LDS buffer (t+2) % 3
  //     global    loads of tile t+3                                -> staging registers
  //     ds_read   fragments of tile t+1 from LDS buffer (t+1) % 3  -> F[(t+1) & 1]
  // so a wave's LDS and global traffic issues under its own MFMAs (a lone workgroup on a CU - small problems, the
  // tail of large ones - has no co-resident waves to hide it), and the barrier only orders tile t+2's stores
  // before the next step's fragment reads (and this step's reads of buffer (t+1) % 3 before its reuse at t+2).
  float4 ra[MI], rb[NI];
  bool oka[MI], okb[NI];
  float fa[2][MI][8], fb[2][NI][8];
  const int ntile = (kend - kbeg + BK - 1) / BK;
  auto load = [&](int t) {
    g_load<!TA, AV, MI>(ra, oka, A, lda, m0, M, kbeg + t * BK, kend, tid);
    g_load<TB, BV, NI>(rb, okb, B, ldb, n0, N, kbeg + t * BK, kend, tid);
  };
  auto store = [&](int buf) {
    s_store<!TA, MI>(As[buf], ra, oka, tid);
    s_store<TB, NI>(Bs[buf], rb, okb, tid);
  };
  auto frags = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < MI; ++i) s_frag<!TA, MI>(fa[set][i], As[buf], wm * 32 * MI + 32 * i, lane);
#pragma unroll
    for (int j = 0; j < NI; ++j) s_frag<TB, NI>(fb[set][j], Bs[buf], wn * 32 * NI + 32 * j, lane);
  };
  auto mfma_steps = [&](int set, int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][s], fb[set][j][s], acc[i][j], 0, 0, 0);
  };
  // one pipeline step; `set` is a compile-time constant at both call sites (loop unrolled by two)
  auto step = [&](int set, int t, int b0) {  // b0 = t % 3
    const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
    mfma_steps(set, 0, 2);
    if (t + 2 < ntile) store(b2);
    mfma_steps(set, 2, 4);
    if (t + 3 < ntile) load(t + 3);
    mfma_steps(set, 4, 6);
    if (t + 1 < ntile) frags(set ^ 1, b1);
    mfma_steps(set, 6, 8);
    __syncthreads();
  };
  if (ntile > 0) {
    load(0);
    store(0);
  }
  if (ntile > 1) {
    load(1);
    store(1);
  }
  if (ntile > 2) load(2);
  __syncthreads();
  if (ntile > 0) frags(0, 0);
  int b = 0;
  for (int t = 0; t < ntile; t += 2) {
    step(0, t, b);
    b = b == 2 ? 0 : b + 1;
    if (t + 1 >= ntile) break;
    step(1, t + 1, b);
    b = b == 2 ? 0 : b + 1;
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *Cz = C + (EPI == 0 ? (size_t)blockIdx.z * M * ldc : 0);
  if (m0 + BM <= M && n0 + BN <= N) {  // interior tile: straight-line loads and stores
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * 32 * NI + j * 32 + (lane & 31);
      const float bj = (EPI == 1 || EPI == 2) ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const size_t at = (size_t)(m0 + wm * 32 * MI + i * 32 + 4 * (lane >> 5)) * ldc + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // rows 8 q + {0, 1, 2, 3} (+ 4 for the upper half wave)
          float mk[4];
          if (EPI == 1 && mask) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[r] = mask[at + (size_t)(r + 8 * q) * ldc];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][4 * q + r];
            if (EPI == 1) {
              v = fmaxf(v + bj, 0.f);
              if (mask) v *= mk[r];
            } else if (EPI == 2) {
              v += bj;
            } else if (EPI == 3) {
              v *= gate3(bias, mask, at + (size_t)(r + 8 * q) * ldc);
            }
            Cz[at + (size_t)(r + 8 * q) * ldc] = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn * 32 * NI + j * 32 + (lane & 31);
    if (col >= N) continue;
    const float bj = (EPI == 1 || EPI == 2) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * MI + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
        if (row >= M) continue;
        float v = acc[i][j][r];
        if (EPI == 1) {
          v = fmaxf(v + bj, 0.f);
          if (mask) v *= mask[(size_t)row * ldc + col];
        } else if (EPI == 2) {
          v += bj;
        } else if (EPI == 3) {
          v *= gate3(bias, mask, (size_t)row * ldc + col);
        }
        Cz[(size_t)row * ldc + col] = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Tall-tile GEMM for the hidden x hidden products of a batch (M or K = the batch's node count).  The 64x64 kernels above
// ask the CU's vector L1 for 16 B per clock and workgroup - 68-80 B/clk at the 4-5 workgroups a CU holds, against the
// 64 B/clk it delivers: 64-75 % of the fp32 MFMA rate is their ceiling.  Here a workgroup of 8 waves owns (16 RT) x 128
// of C (RT = 6 .. 10 row sub-tiles, picked per launch so that the tile count fills whole rounds of 256 CUs):
//  * wave w owns the 16 columns 16 w .. 16 w + 15 of the tile and all its rows: RT accumulators of v_mfma_f32_16x16x4_f32,
//    issued with the operands swapped (D = B^T A^T) so that a lane holds four consecutive columns of one row of C and the
//    epilogue is one 16-byte store (bias / mask one 16-byte load) per sub-tile;
//  * operand tiles global -> LDS directly as in k_gemm_dl (four stages, 16 k per stage), (16 RT + 128) * 64 B per stage:
//    7-8 B per clock and workgroup from the L1;
//  * MFMA step s multiplies k = 4 q + s in lane group q = lane >> 4, so that a k-contiguous operand's fragment is one
//    ds_read_b128 ([x][16 k] rows, quads XOR-swizzled by (x >> 1) & 3: the four 16-lane groups of the read are
//    conflict-free) and an x-contiguous one's four ds_read_b32 ([16 k][W x], 16-float groups of row k XOR-swizzled by
//    (k >> 2) & 1: lane groups q and q + 1 read different bank halves);
//  * a DMA instruction brings 1 KB = 16 rows of a k-contiguous tile (or 256 / W rows of an x-contiguous one); the RT + 8
//    instructions of a stage are dealt round-robin to the 8 waves, and a wave without a real one in a round issues it
//    into a scratch KB (every wave's vmcnt then counts the same number per stage).
// Contract (host): vec_ok() operands and C, x-contiguous A only with 16 RT % 32 == 0.
// ------------------------------------------------------------------------------------------------
constexpr int WD_ST = 4;
template <int RT, int NW>  // RT 16-row sub-tiles x NW waves of 16 columns each
struct WideTile {
  static constexpr int AI = (RT + NW - 1) / NW;                  // A instructions per wave and stage
  static constexpr int stage_floats = (RT + NW) * 256;           // A tile, then B tile
  static constexpr int lds_floats = WD_ST * stage_floats + 256;  // + the scratch KB
};

// source address of DMA instruction j (1 KB = quads 64 j .. 64 j + 63 of the tile) for this lane
template <bool KC>
__device__ __forceinline__ const float *wd_src(const float *P, int ld, int x0, int X, int W, int j, int lane) {
  const int q = 64 * j + lane;
  if (KC) {
    const int x = q >> 2;
    return P + (size_t)min(x0 + x, X - 1) * ld + 4 * ((q & 3) ^ ((x >> 1) & 3));  // (+ k0)
  }
  const int wq = W >> 2, k = q / wq, xq = (q - k * wq) ^ (4 * ((k >> 2) & 1));
  return P + (size_t)k * ld + min(x0 + 4 * xq, X - 4);  // (+ k0 * ld)
}
// the four values lane (i = lane & 15, q = lane >> 4) feeds to MFMA steps 0..3 for tile row / column xb + i: k = 4 q + s
template <bool KC>
__device__ __forceinline__ float4 wd_frag(const float *T, int W, int xb, int lane) {
  const int x = xb + (lane & 15), q = lane >> 4;
  if (KC) return *reinterpret_cast<const float4 *>(T + x * 16 + 4 * (q ^ ((x >> 1) & 3)));
  const float *p = T + (4 * q) * W + (x ^ (16 * (q & 1)));
  return make_float4(p[0], p[W], p[2 * W], p[3 * W]);
}
// the partial last K-tile goes through registers with zero fill
template <bool KC, int NT>
__device__ __forceinline__ void wd_tail(float *T, const float *P, int ld, int x0, int X, int W, int k0, int kend, int tid) {
  for (int q = tid; q < 4 * W; q += NT) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int x = q >> 2, c = q & 3;
      if (k0 + 4 * c < kend) v = *reinterpret_cast<const float4 *>(P + (size_t)min(x0 + x, X - 1) * ld + k0 + 4 * c);
      *reinterpret_cast<float4 *>(T + x * 16 + 4 * (c ^ ((x >> 1) & 3))) = v;
    } else {
      const int wq = W >> 2, k = q / wq, xq = q - k * wq;
      if (k0 + k < kend) v = *reinterpret_cast<const float4 *>(P + (size_t)(k0 + k) * ld + min(x0 + 4 * xq, X - 4));
      *reinterpret_cast<float4 *>(T + k * W + 4 * (xq ^ (4 * ((k >> 2) & 1)))) = v;
    }
  }
}

template <bool AKC, bool BKC, int EPI, int RT, int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_wide(int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                   int ldb, float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                                   const float *__restrict__ mask, int k_per_split, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float wd_smem[];
  using WT = WideTile<RT, NW>;
  constexpr int TM = 16 * RT, WD_N = 16 * NW;
  int tm, tn;
  if (!tile_of_block(tiles_m, tiles_n, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = tm * TM, n0 = tn * WD_N;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  floatx4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nfull = (kend - kbeg) / 16, tail = (kend - kbeg) - 16 * nfull;
  const unsigned lds0 =
      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const void *)wd_smem);
  constexpr unsigned stage_bytes = WT::stage_floats * sizeof(float);
  auto dma16 = [](const float *g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");
  };
  // this wave's instructions of a stage: A instruction j = wave + NW i (the scratch KB when j >= RT), B instruction j = wave
  const float *ga[WT::AI];
  unsigned la[WT::AI], la_step[WT::AI];
#pragma unroll
  for (int i = 0; i < WT::AI; ++i) {
    const int j = wave + NW * i;
    const bool real = j < RT;
    ga[i] = wd_src<AKC>(A, lda, m0, M, TM, real ? j : 0, lane) + (AKC ? (size_t)kbeg : (size_t)kbeg * lda);
    la[i] = real ? lds0 + j * 1024u : lds0 + WD_ST * stage_bytes;
    la_step[i] = real ? stage_bytes : 0u;
  }
  const float *gb = wd_src<BKC>(B, ldb, n0, N, WD_N, wave, lane) + (BKC ? (size_t)kbeg : (size_t)kbeg * ldb);
  const unsigned lb = lds0 + (RT + wave) * 1024u;
  const size_t sa = AKC ? 16 : (size_t)16 * lda, sb = BKC ? 16 : (size_t)16 * ldb;  // source step per K-tile
  constexpr int PER = WT::AI + 1;  // DMA instructions of this wave per stage
  const unsigned scratch = lds0 + WD_ST * stage_bytes;
  // DMA instruction n (0 .. PER - 1) of K-tile t; past the last full tile it re-reads that tile into the scratch KB, so that
  // every step issues PER instructions and one vmcnt value is right throughout
  auto dma = [&](int n, int t) {
    const bool live = t < nfull;
    const int ts = live ? t : nfull - 1;
    const unsigned st = (unsigned)(t & (WD_ST - 1));
    if (n < WT::AI) dma16(ga[n] + sa * ts, live ? la[n] + st * la_step[n] : scratch);
    else dma16(gb + sb * ts, live ? lb + st * stage_bytes : scratch);
  };
  // K-tile t is multiplied from registers; between its MFMAs (one filler behind each of the first few, in the shadow of the
  // MFMA pipe) the wave issues its DMA instructions of tile t + 4 into the stage tile t just left and reads the fragments
  // of tile t + 1.  The barrier at the top (tile t + 1 complete in LDS, every wave has tile t in registers) is followed by
  // MFMAs that wait for nothing.
  float4 fa0[RT], fa1[RT], fb0, fb1;
  auto step = [&](int t, const float4 (&fa)[RT], const float4 &fb, float4 (&na)[RT], float4 &nb) {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * PER));  // tiles t + 2 and t + 3 may still be in flight
    __builtin_amdgcn_s_barrier();
    const float *As = wd_smem + ((t + 1) & (WD_ST - 1)) * WT::stage_floats, *Bs = As + RT * 256;
    const float bs[4] = {fb.x, fb.y, fb.z, fb.w};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const float as[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[s], as[s], acc[i], 0, 0, 0);
        const int n = s * RT + i;
        __builtin_amdgcn_sched_barrier(0);
        if (n < PER) dma(n, t + 4);
        else if (n == PER) nb = wd_frag<BKC>(Bs, WD_N, 16 * wave, lane);
        else if (n <= PER + RT) na[n - PER - 1] = wd_frag<AKC>(As, TM, 16 * (n - PER - 1), lane);
        if (n <= PER + RT) __builtin_amdgcn_sched_barrier(0);
      }
  };
  if (nfull > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int n = 0; n < PER; ++n) dma(n, t);
    __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * PER));
    __builtin_amdgcn_s_barrier();
    fb0 = wd_frag<BKC>(wd_smem + RT * 256, WD_N, 16 * wave, lane);
#pragma unroll
    for (int i = 0; i < RT; ++i) fa0[i] = wd_frag<AKC>(wd_smem, TM, 16 * i, lane);
  }
  for (int t = 0; t < nfull; t += 2) {
    step(t, fa0, fb0, fa1, fb1);
    if (t + 1 >= nfull) break;
    step(t + 1, fa1, fb1, fa0, fb0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // the scratch re-reads of the last steps (LDS must not be written after the workgroup ends)
  auto read = [&](float4 (&fa)[RT], float4 &fb, int st) {
    const float *As = wd_smem + st * WT::stage_floats, *Bs = As + RT * 256;
    fb = wd_frag<BKC>(Bs, WD_N, 16 * wave, lane);
#pragma unroll
    for (int i = 0; i < RT; ++i) fa[i] = wd_frag<AKC>(As, TM, 16 * i, lane);
  };
  auto multiply = [&](const float4 (&fa)[RT], const float4 &fb) {
    const float bs[4] = {fb.x, fb.y, fb.z, fb.w};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const float as[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[s], as[s], acc[i], 0, 0, 0);
      }
  };
  if (tail > 0) {
    __syncthreads();
    const int st = nfull & (WD_ST - 1), k0 = kbeg + 16 * nfull;
    float *As = wd_smem + st * WT::stage_floats;
    wd_tail<AKC, 64 * NW>(As, A, lda, m0, M, TM, k0, kend, tid);
    wd_tail<BKC, 64 * NW>(As + RT * 256, B, ldb, n0, N, WD_N, k0, kend, tid);
    __syncthreads();
    read(fa0, fb0, st);
    multiply(fa0, fb0);
  }
  // D = (B^T A^T) sub-tile: lane holds C[m0 + 16 i + (lane & 15)][n .. n + 3], n = n0 + 16 wave + 4 (lane >> 4)
  float *Cz = C + (EPI == 0 ? (size_t)blockIdx.z * M * ldc : 0);
  const int n = n0 + 16 * wave + 4 * (lane >> 4);
  if (n < N) {
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 1 || EPI == 2) bj = *reinterpret_cast<const float4 *>(bias + n);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int m = m0 + 16 * i + (lane & 15);
      if (m >= M) continue;
      const size_t at = (size_t)m * ldc + n;
      float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      if (EPI == 1) {
        v = make_float4(fmaxf(v.x + bj.x, 0.f), fmaxf(v.y + bj.y, 0.f), fmaxf(v.z + bj.z, 0.f), fmaxf(v.w + bj.w, 0.f));
        if (mask) {
          const float4 mk = *reinterpret_cast<const float4 *>(mask + at);
          v = make_float4(v.x * mk.x, v.y * mk.y, v.z * mk.z, v.w * mk.w);
        }
      } else if (EPI == 2) {
        v = make_float4(v.x + bj.x, v.y + bj.y, v.z + bj.z, v.w + bj.w);
      } else if (EPI == 3) {
        const float4 h = *reinterpret_cast<const float4 *>(bias + at);
        float4 g = make_float4(h.x > 0.f ? 1.f : 0.f, h.y > 0.f ? 1.f : 0.f, h.z > 0.f ? 1.f : 0.f, h.w > 0.f ? 1.f : 0.f);
        if (mask) {
          const float4 mk = *reinterpret_cast<const float4 *>(mask + at);
          g = make_float4(g.x * mk.x, g.y * mk.y, g.z * mk.z, g.w * mk.w);
        }
        v = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
      }
      *reinterpret_cast<float4 *>(Cz + at) = v;
    }
  }
}

// thin-M products  out[m][n] = sum_k A[k][m] B[k][n]  (m < M <= 8; A stored [K x lda]) plus, as row M, the column sums
// of B: one pass over B (HBM-bound).  A workgroup owns 256 adjacent columns (a lane 4 of them; N % 4 == 0) of the K-slice
// blockIdx.y; its four waves take every fourth row of the slice and are summed in a fixed order through LDS (one wave per
// slice walked 34 rows at 4 340 nodes with half the chip idle: 14.5 us per call, three calls per train step);
// partials -> part[y][M + 1][N]
__global__ __launch_bounds__(256) void k_thin_tn_part(int K, int N, int M, const float *__restrict__ A, int lda,
                                                      const float *__restrict__ B, int ldb, float *__restrict__ part,
                                                      int rows_per_block) {
  __shared__ float4 red[3][9][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const bool col_ok = n < N;
  const int k0 = blockIdx.y * rows_per_block, k1 = min(K, k0 + rows_per_block);
  float4 acc[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok)
    for (int k = k0 + wave; k < k1; k += 4) {
      const float4 b = *reinterpret_cast<const float4 *>(B + (size_t)k * ldb + n);
      const float *a = A + (size_t)k * lda;  // wave-uniform: scalar loads
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (m < M) {
          const float am = a[m];
          acc[m].x += am * b.x; acc[m].y += am * b.y; acc[m].z += am * b.z; acc[m].w += am * b.w;
        }
      acc[8].x += b.x; acc[8].y += b.y; acc[8].z += b.z; acc[8].w += b.w;
    }
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < 9; ++m)
      if (m < M || m == 8) red[wave - 1][m][lane] = acc[m];
  }
  __syncthreads();
  if (wave > 0 || !col_ok) return;
#pragma unroll
  for (int m = 0; m < 9; ++m)
    if (m < M || m == 8) {
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const float4 o = red[w][m][lane];
        acc[m].x += o.x; acc[m].y += o.y; acc[m].z += o.z; acc[m].w += o.w;
      }
    }
  float *o = part + (size_t)blockIdx.y * (M + 1) * N + n;
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < M) *reinterpret_cast<float4 *>(o + (size_t)m * N) = acc[m];
  *reinterpret_cast<float4 *>(o + (size_t)M * N) = acc[8];
}

// second stage (deterministic): rows < rows_w of the [M x N] product -> outW, the column-sum row -> outB (either may
// be null). 64 outputs per workgroup, the S partials of each summed by 16 threads in a fixed order.
// With outA, one more workgroup (the last) writes the column sums of A itself, outA[m] = sum_k A[k][m] (m < M; the bias
// gradient of the read-out layer, A = dOut), in a fixed order too.
__global__ __launch_bounds__(1024) void k_thin_tn_reduce(int N, int M, int S, const float *part, float *outW, int rows_w,
                                                         float *outB, const float *A, int lda, int K, float *outA) {
  __shared__ float red[16][64];
  if (outA && blockIdx.x == gridDim.x - 1) {
    float *r1 = &red[0][0];
    for (int m = 0; m < M; ++m) {
      float s = 0.f;
      for (int k = threadIdx.x; k < K; k += 1024) s += A[(size_t)k * lda + m];
      r1[threadIdx.x] = s;
      __syncthreads();
      for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) r1[threadIdx.x] += r1[threadIdx.x + h];
        __syncthreads();
      }
      if (threadIdx.x == 0) outA[m] = r1[0];
      __syncthreads();
    }
    return;
  }
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const int total = (M + 1) * N;
  float s = 0.f;
  if (i < total)
    for (int z = q; z < S; z += 16) s += part[(size_t)z * total + i];
  red[q][c] = s;
  __syncthreads();
  if (q != 0 || i >= total) return;
  s = 0.f;
#pragma unroll
  for (int z = 0; z < 16; ++z) s += red[z][c];
  const int m = i / N;
  if (m < M) {
    if (outW && m < rows_w) outW[i] = s;
  } else if (outB) {
    outB[i - M * N] = s;
  }
}

// The read-out layer's backward in one pass over H2 (out_dim <= 8): dZ2 as k_dz2 writes it, and - from the same registers -
// the partials of dWf = dOut^T H2m (rows m < M of the thin product: H2 already holds relu(Z2) * mask) and of db2 = the
// column sums of dZ2 (row M), in k_thin_tn_part's layout and summation order: k_thin_tn_reduce finishes both.  Replaces
// thin product + k_dz2 + column sums (five launches) by two.
__global__ __launch_bounds__(256) void k_dz2_sums(int K, int N, int M, const float *__restrict__ dOut, const float *__restrict__ Wf,
                                                  const float *__restrict__ mask, const float *__restrict__ H2, float *__restrict__ dZ2,
                                                  float *__restrict__ part, int rows_per_block) {
  __shared__ float4 red[3][9][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const bool col_ok = n < N;
  const int k0 = blockIdx.y * rows_per_block, k1 = min(K, k0 + rows_per_block);
  float4 acc[9], wf[8];
#pragma unroll
  for (int m = 0; m < 9; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 8; ++m) wf[m] = (m < M && col_ok) ? *reinterpret_cast<const float4 *>(Wf + (size_t)m * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok)
    for (int k = k0 + wave; k < k1; k += 4) {
      const float4 h = *reinterpret_cast<const float4 *>(H2 + (size_t)k * N + n);
      const float *a = dOut + (size_t)k * M;  // wave-uniform: scalar loads
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (m < M) {
          const float am = a[m];
          s.x += am * wf[m].x; s.y += am * wf[m].y; s.z += am * wf[m].z; s.w += am * wf[m].w;
          acc[m].x += am * h.x; acc[m].y += am * h.y; acc[m].z += am * h.z; acc[m].w += am * h.w;
        }
      float4 g = make_float4(h.x > 0.f ? 1.f : 0.f, h.y > 0.f ? 1.f : 0.f, h.z > 0.f ? 1.f : 0.f, h.w > 0.f ? 1.f : 0.f);
      if (mask) {
        const float4 mk = *reinterpret_cast<const float4 *>(mask + (size_t)k * N + n);
        g.x *= mk.x; g.y *= mk.y; g.z *= mk.z; g.w *= mk.w;
      }
      const float4 dz = make_float4(s.x * g.x, s.y * g.y, s.z * g.z, s.w * g.w);
      *reinterpret_cast<float4 *>(dZ2 + (size_t)k * N + n) = dz;
      acc[8].x += dz.x; acc[8].y += dz.y; acc[8].z += dz.z; acc[8].w += dz.w;
    }
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < 9; ++m)
      if (m < M || m == 8) red[wave - 1][m][lane] = acc[m];
  }
  __syncthreads();
  if (wave > 0 || !col_ok) return;
#pragma unroll
  for (int m = 0; m < 9; ++m)
    if (m < M || m == 8) {
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const float4 o = red[w][m][lane];
        acc[m].x += o.x; acc[m].y += o.y; acc[m].z += o.z; acc[m].w += o.w;
      }
    }
  float *o = part + (size_t)blockIdx.y * (M + 1) * N + n;
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < M) *reinterpret_cast<float4 *>(o + (size_t)m * N) = acc[m];
  *reinterpret_cast<float4 *>(o + (size_t)M * N) = acc[8];
}

// deterministic second stage of split-K: out = sum_z part[z]
__global__ void k_splitk_reduce(int n, int S, const float *part, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(size_t)z * n + i];
  out[i] = s;
}

// read-out layers of up to this many outputs are served by the one-pass kernels below (k_linear_out, k_dz2_sums / k_dz2: H2 is
// read once, HBM-bound); wider ones (the critic: 100) by the matrix-core products with the epilogues EPI 2 / 3
#ifndef DRLGX_THIN_OUT
#define DRLGX_THIN_OUT 8  // (-DDRLGX_THIN_OUT=128: the one-pass kernels for every width, A/B runs)
#endif
constexpr int kThinOut = DRLGX_THIN_OUT;
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
  const uintptr_t al = reinterpret_cast<uintptr_t>(Wf) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(H2) | reinterpret_cast<uintptr_t>(dZ2);
  if ((hidden & 3) == 0 && (al & 15) == 0) {  // four columns per lane, 16-byte accesses
    for (int c4 = threadIdx.x; c4 < (hidden >> 2); c4 += 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o < out_dim; ++o) {
        const float d = dOut[(size_t)n * out_dim + o];
        const float4 w = reinterpret_cast<const float4 *>(Wf + (size_t)o * hidden)[c4];
        s.x += d * w.x; s.y += d * w.y; s.z += d * w.z; s.w += d * w.w;
      }
      const float4 h = reinterpret_cast<const float4 *>(H2 + (size_t)n * hidden)[c4];
      float4 g = make_float4(h.x > 0.f ? 1.f : 0.f, h.y > 0.f ? 1.f : 0.f, h.z > 0.f ? 1.f : 0.f, h.w > 0.f ? 1.f : 0.f);
      if (mask) {
        const float4 m = reinterpret_cast<const float4 *>(mask + (size_t)n * hidden)[c4];
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
      }
      reinterpret_cast<float4 *>(dZ2 + (size_t)n * hidden)[c4] = make_float4(s.x * g.x, s.y * g.y, s.z * g.z, s.w * g.w);
    }
    return;
  }
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
  float *deg, *selfw, *wn_dst, *wn_src, *AX, *b1s, *AH1, *H2, *T0, *T1, *part;
  int *cnt_dst, *cnt_src, *ptr_dst, *ptr_src, *cur_dst, *cur_src, *eid_dst, *eid_src, *nbr_dst, *nbr_src, *end_dst, *end_src;
  size_t part_floats, counters_bytes;
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
  takef(w ? &w->selfw : &df, N);
  takef(w ? &w->wn_dst : &df, E);
  takef(w ? &w->wn_src : &df, E);
  takef(w ? &w->AX : &df, (size_t)N * 8);
  takef(w ? &w->b1s : &df, hidden);  // the forward's b1, for the backward's layer-1 ReLU gate (H1 is recomputed, not stored)
  takef(w ? &w->AH1 : &df, NH);
  takef(w ? &w->H2 : &df, NH);
  takef(w ? &w->T0 : &df, NH);
  takef(w ? &w->T1 : &df, NH);
  takef(w ? &w->part : &df, part);
  if (w) w->part_floats = part;
  takei(w ? &w->cnt_dst : &di, N + 1);  // the four counters are contiguous: one memset (see build_graph)
  takei(w ? &w->cnt_src : &di, N + 1);
  takei(w ? &w->cur_dst : &di, N + 1);
  takei(w ? &w->cur_src : &di, N + 1);
  if (w) w->counters_bytes = off - ((char *)w->cnt_dst - base);
  takei(w ? &w->ptr_dst : &di, N + 1);
  takei(w ? &w->ptr_src : &di, N + 1);
  takei(w ? &w->eid_dst : &di, E);
  takei(w ? &w->eid_src : &di, E);
  takei(w ? &w->nbr_dst : &di, E);
  takei(w ? &w->nbr_src : &di, E);
  takei(w ? &w->end_dst : &di, N + 1);
  takei(w ? &w->end_src : &di, N + 1);
  (void)out_dim;
  return off;
}

bool vec_ok(const float *p, int ld, int contiguous_dim) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (contiguous_dim & 3) == 0;
}

// workgroup tile of the small-product kernels: 64x64 (four waves of 32x32), best or equal among the register-staged tilings
// on every GCN shape; DRLGX_GEMM_TILE=1 selects the 64x128 variant for experiments.  (Large products: gemm_wide below.)
int pick_tile() {
  static const int v = [] {
    const char *e = getenv("DRLGX_GEMM_TILE");
    return e ? atoi(e) : 2;
  }();
  return v;
}

// DRLGX_GEMM_DL=0 keeps the register-staged kernel (A/B runs)
bool gemm_direct_to_lds() {
  static const bool v = [] {
    const char *e = std::getenv("DRLGX_GEMM_DL");
    return !(e && e[0] == '0');
  }();
  return v;
}

template <bool TA, bool TB, int EPI, int MI, int NI>
void gemm_tile(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
               const float *bias, const float *mask, int kps) {
  const int tiles_m = (M + 64 * MI - 1) / (64 * MI), tiles_n = (N + 64 * NI - 1) / (64 * NI);
  dim3 grid(tiles_n * ((tiles_m + 7) / 8) * 8, 1, (K + kps - 1) / kps);
  const bool avec = vec_ok(A, lda, TA ? M : K), bvec = vec_ok(B, ldb, TB ? K : N);
  if (MI == 1 && NI == 1 && avec && bvec && (TA ? M : K) >= 4 && (TB ? K : N) >= 4 && M >= 1 && N >= 4 && gemm_direct_to_lds()) {
    hipLaunchKernelGGL((k_gemm_dl<!TA, TB, EPI>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps, tiles_m, tiles_n);
    return;
  }
#define DRLGX_GEMM(AV, BV)                                                                                                     \
  hipLaunchKernelGGL((k_gemm<TA, TB, EPI, AV, BV, MI, NI>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, \
                     kps, tiles_m, tiles_n)
  if (avec && bvec) DRLGX_GEMM(true, true);
  else DRLGX_GEMM(false, false);
#undef DRLGX_GEMM
}

// DRLGX_GEMM_WIDE=0 keeps the 64x64 kernels everywhere (A/B runs); DRLGX_GEMM_WIDE=6..10 pins the tile height
int gemm_wide_mode() {
  static const int v = [] {
    const char *e = std::getenv("DRLGX_GEMM_WIDE");
    return e ? atoi(e) : -1;
  }();
  return v;
}

template <bool TA, bool TB, int EPI, int RT, int NW = 8>
void gemm_wide_launch(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                      const float *bias, const float *mask, int kps) {
  const int tiles_m = (M + 16 * RT - 1) / (16 * RT), tiles_n = (N + 16 * NW - 1) / (16 * NW);
  dim3 grid(tiles_n * ((tiles_m + 7) / 8) * 8, 1, (K + kps - 1) / kps);
  constexpr int lds = WideTile<RT, NW>::lds_floats * (int)sizeof(float);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&k_gemm_wide<!TA, TB, EPI, RT, NW>)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, lds);
  hipLaunchKernelGGL((k_gemm_wide<!TA, TB, EPI, RT, NW>), grid, dim3(64 * NW), lds, st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps,
                     tiles_m, tiles_n);
}

// Tile of an M x N product in S K-slices: rt 16-row sub-tiles x nw waves of 16 columns, or rt = 0 for the 64x64 kernels.
//  * enough 128-row x 128-column tiles to occupy half the chip: 8 waves, the height whose tile count wastes least of the last
//    round of 256 CUs (measured order at 4 340 and 17 288 rows: profiles/r04_ab_gemm_tall_tiles.txt);
//  * else, if the 64x64 tiles would not fit one round of 256 CUs: 4 waves x 64 columns, same rule (the 1 000 - 2 000-node
//    mini-batches of the DQN loop: 96 x 64 tiles fill the chip once where 64 x 64 ones need a second, thin round);
//  * else the 64x64 kernels.
struct WidePick {
  int rt, nw;
};
WidePick wide_pick(int M, int N, int S, bool ta) {
  const int mode = gemm_wide_mode();
  if (mode == 0) return {0, 0};
  auto best_rt = [&](int nw) {
    int best = 0;
    long best_cost = 0;
    for (int rt = 6; rt <= (nw == 8 ? 10 : 8); ++rt) {  // (4 waves: one round of 96 / 112 / 128-row tiles covers every size that gets here)
      if (ta && (rt & 1)) continue;  // x-contiguous A: whole 32-float swizzle blocks
      if (mode >= 6 && mode <= 10 && !(ta && (mode & 1)) && rt != mode) continue;
      const long tiles = (long)((M + 16 * rt - 1) / (16 * rt)) * ((N + 16 * nw - 1) / (16 * nw)) * S;
      const long cost = ((tiles + 255) / 256) * rt;
      if (!best || cost <= best_cost) best = rt, best_cost = cost;
    }
    return best;
  };
  if ((long)((M + 127) / 128) * ((N + 127) / 128) * S >= 128) return {best_rt(8), 8};
  if (!ta && (long)((M + 63) / 64) * ((N + 63) / 64) * S > 256 && (long)((M + 95) / 96) * ((N + 63) / 64) * S >= 128) return {best_rt(4), 4};
  return {0, 0};
}

template <bool TA, bool TB, int EPI>
bool gemm_wide(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, const float *bias,
               const float *mask, int kps) {
  if (!vec_ok(A, lda, TA ? M : K) || !vec_ok(B, ldb, TB ? K : N) || !vec_ok(C, ldc, N)) return false;
  if (((EPI == 1 || EPI == 2) && !vec_ok(bias, 4, 4)) || (EPI == 3 && !vec_ok(bias, ldc, N)) ||
      ((EPI == 1 || EPI == 3) && mask && !vec_ok(mask, ldc, N)) || (TA ? M : K) < 4 || (TB ? K : N) < 4 || kps < 16) return false;
  const WidePick pick = wide_pick(M, N, (K + kps - 1) / kps, TA);
#define DRLGX_WIDE(RT, NW)                                                                          \
  case RT:                                                                                          \
    gemm_wide_launch<TA, TB, EPI, RT, NW>(st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps); \
    return true
  if (pick.nw == 8) {
    switch (pick.rt) {
      DRLGX_WIDE(6, 8);
      DRLGX_WIDE(8, 8);
      DRLGX_WIDE(10, 8);
      default: break;
    }
    if constexpr (!TA) {
      switch (pick.rt) {
        DRLGX_WIDE(7, 8);
        DRLGX_WIDE(9, 8);
        default: break;
      }
    }
  }
  if constexpr (!TA) {
    if (pick.nw == 4) {
      switch (pick.rt) {
        DRLGX_WIDE(6, 4);
        DRLGX_WIDE(7, 4);
        DRLGX_WIDE(8, 4);
        default: break;
      }
    }
  }
#undef DRLGX_WIDE
  return false;
}

template <bool TA, bool TB, int EPI>
void gemm(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, const float *bias,
          const float *mask, int splits) {
  const int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  if (gemm_wide<TA, TB, EPI>(st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps)) return;
  if (pick_tile() == 1) gemm_tile<TA, TB, EPI, 1, 2>(st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps);
  else gemm_tile<TA, TB, EPI, 1, 1>(st, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, kps);
}

// weight-gradient GEMM  C[M x N] = A^T B with K = #nodes: split-K (enough splits to fill the chip) + deterministic reduce
// max_splits: 8 for the hidden x hidden gradients (the partials' workspace holds eight of them); the thin read-out gradient
// (M = out_dim rows: 32 tiles) takes more slices to reach every CU
void gemm_tn_splitk(hipStream_t st, const GcnWs &w, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
                    int max_splits = 8) {
  const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
  int splits = (int)std::min<size_t>(max_splits, w.part_floats / ((size_t)M * N));
  splits = std::max(1, std::min({splits, (int)((1024 + tiles - 1) / tiles), (K + 255) / 256}));
  const int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  const int S = (K + kps - 1) / kps;
  if (S == 1) {
    gemm<true, false, 0>(st, M, N, K, A, lda, B, ldb, C, N, nullptr, nullptr, 1);
    return;
  }
#ifndef DRLGX_THIN_TN_WIDE  // (A/B: the tall tile for the thin product too)
  if (max_splits > 8)  // thin M: 64 x 64 tiles
    gemm_tile<true, false, 0, 1, 1>(st, M, N, K, A, lda, B, ldb, w.part, N, nullptr, nullptr, kps);
  else
#endif
    gemm<true, false, 0>(st, M, N, K, A, lda, B, ldb, w.part, N, nullptr, nullptr, S);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((M * N + 255) / 256), dim3(256), 0, st, M * N, S, w.part, C);
}

// outW[rows_w x N] = (A^T B)[:rows_w], outB[N] = column sums of B, for M <= 8 columns of A (M = 0: column sums
// only): one pass over B. N % 4 == 0 and 16-byte aligned B rows (hidden-sized operands).
void thin_tn(hipStream_t st, const GcnWs &w, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *outW,
             int rows_w, float *outB, float *outA = nullptr) {
  int nb = std::min(128, (K + 7) / 8);
  nb = (int)std::max<size_t>(1, std::min<size_t>(nb, w.part_floats / ((size_t)(M + 1) * N)));
  const int rpb = (K + nb - 1) / nb;
  nb = (K + rpb - 1) / rpb;
  hipLaunchKernelGGL(k_thin_tn_part, dim3((N / 4 + 63) / 64, nb), dim3(256), 0, st, K, N, M, A, lda, B, ldb, w.part, rpb);
  hipLaunchKernelGGL(k_thin_tn_reduce, dim3(((M + 1) * N + 63) / 64 + (outA ? 1 : 0)), dim3(1024), 0, st, N, M, nb, w.part, outW, rows_w, outB,
                     A, lda, K, outA);
}

void colsum(hipStream_t st, const GcnWs &w, int N, int C, const float *X, float *out) {
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    thin_tn(st, w, 0, C, N, nullptr, 0, X, C, nullptr, 0, out);
    return;
  }
  const int nb = 64, rpb = (N + nb - 1) / nb;
  hipLaunchKernelGGL(k_colsum_part, dim3((C + 255) / 256, nb), dim3(256), 0, st, N, C, X, w.part, rpb);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((C + 255) / 256), dim3(256), 0, st, C, nb, w.part, out);
}

void build_graph(hipStream_t st, const GcnWs &w, int N, int E, const int64_t *ei, const float *ew) {
  hipMemsetAsync(w.cnt_dst, 0, w.counters_bytes, st);  // cnt_dst, cnt_src, cur_dst, cur_src
  hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.selfw), 0x40000000, (size_t)N, st);  // 2.0f: the improved-GCN fill value
  if (E > 0) hipLaunchKernelGGL(k_degree, dim3((E + 255) / 256), dim3(256), 0, st, N, E, ei, ew, w.cnt_dst, w.cnt_src, w.selfw);
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(1024), 0, st, N, w.cnt_dst, w.ptr_dst, w.cnt_src, w.ptr_src);
  if (E > 0) {
    hipLaunchKernelGGL(k_csr_fill, dim3((E + 255) / 256), dim3(256), 0, st, N, E, ei, w.ptr_dst, w.cur_dst, w.eid_dst, w.ptr_src,
                       w.cur_src, w.eid_src);
  }
  const dim3 gn((N + 127) / 128), g2((2 * N + 127) / 128), bn(128);
  hipLaunchKernelGGL(k_csr_sort, g2, bn, 0, st, N, w.ptr_dst, w.eid_dst, w.ptr_src, w.eid_src);
  hipLaunchKernelGGL(k_degree_sum, gn, bn, 0, st, N, ew, w.ptr_src, w.eid_src, w.selfw, w.deg);
  hipLaunchKernelGGL(k_csr_finish, g2, bn, 0, st, N, E, ei, ew, w.deg, w.ptr_dst, w.eid_dst, w.nbr_dst, w.wn_dst, w.ptr_src,
                     w.eid_src, w.nbr_src, w.wn_src, w.end_dst, w.end_src);
}

// false: a graph of the batch may have more edges than the per-graph kernel sorts in LDS (the caller falls back to build_graph)
// x / in_dim: the node features - AX = Â X comes out of the same launch (k_ax otherwise)
bool build_graph_batched(hipStream_t st, const GcnWs &w, int N, int E, const int64_t *ei, const float *ew, int G, const int *node_off,
                         const int *edge_off, int max_edges_per_graph, const float *x, int in_dim) {
  if (max_edges_per_graph > kCsrMaxEdges) return false;
  int P2 = 64;
  while (P2 < max_edges_per_graph) P2 <<= 1;
  const int extra = P2 <= 4096 ? 1 : 0;  // weights + packed endpoints beside the keys: 16 bytes per edge slot, <= 64 KB
  const size_t lds = (size_t)(extra ? 4 : 2) * P2 * sizeof(uint32_t);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&k_csr_graphs)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, 160 * 1024);
  hipLaunchKernelGGL(k_csr_graphs, dim3(G), dim3(256), lds, st, N, E, P2, extra, ei, ew, node_off, edge_off, w.deg, w.selfw, w.ptr_dst, w.end_dst,
                     w.nbr_dst, w.wn_dst, w.ptr_src, w.end_src, w.nbr_src, w.wn_src, x, in_dim, w.AX, 0);
  return true;
}

}  // namespace

extern "C" {

int drlgx_debug_gemm_tile_rows(int m, int n, int k_slices, int transpose_a) {
  const WidePick p = wide_pick(m, n, k_slices, transpose_a != 0);
  return p.rt ? 1000 * (16 * p.nw) + 16 * p.rt : 64064;
}

size_t drlgx_gcn_workspace_bytes(int n_nodes, int n_edges, int hidden, int out_dim) {
  if (n_nodes <= 0 || n_edges < 0 || hidden <= 0 || out_dim <= 0) return 0;
  return carve(nullptr, nullptr, n_nodes, std::max(n_edges, 1), hidden, out_dim) + 256;
}

constexpr int kPrebuilt = -7;  // gcn_forward_impl's n_graphs: the graph part of the workspace is already built
static int gcn_forward_impl(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                            const int64_t *edge_index, const float *edge_attr, const float *W1, const float *b1, const float *W2,
                            const float *b2, const float *Wf, const float *bf, const float *dropout_mask, float *out, void *ws_dev,
                            int n_graphs, const int32_t *node_off, const int32_t *edge_off, int max_edges_per_graph) {
  const bool prebuilt = n_graphs == kPrebuilt;
  if (n_nodes <= 0 || n_edges < 0 || in_dim <= 0 || in_dim > 8 || hidden <= 0 || (hidden & 3) || out_dim <= 0 || (!x && !prebuilt) || !W1 ||
      !b1 || !W2 || !b2 || !Wf || !bf || !out || !ws_dev || (n_edges > 0 && !prebuilt && (!edge_index || !edge_attr)))
    return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  GcnWs w;
  carve(&w, reinterpret_cast<char *>(ws_dev), n_nodes, std::max(n_edges, 1), hidden, out_dim);
  if (n_graphs == kPrebuilt) {
    // normalisation, both CSRs and AX already are in the workspace (drlgx_gcn_collate_csr)
  } else if (n_graphs <= 0 ||
      !build_graph_batched(st, w, n_nodes, n_edges, edge_index, edge_attr, n_graphs, node_off, edge_off, max_edges_per_graph, x, in_dim)) {
    build_graph(st, w, n_nodes, n_edges, edge_index, edge_attr);
    hipLaunchKernelGGL(k_ax, dim3((n_nodes * 8 + 255) / 256), dim3(256), 0, st, n_nodes, in_dim, x, w.deg, w.selfw, w.ptr_dst, w.end_dst,
                       w.nbr_dst, w.wn_dst, w.AX);
  }
  {
    const dim3 ga((n_nodes + kAggNodes - 1) / kAggNodes), ba(256);
    if (in_dim == 5)  // the reference's feature count: compiled straight-line
      hipLaunchKernelGGL(k_aggregate_l1<5>, ga, ba, 0, st, n_nodes, in_dim, hidden, w.AX, W1, b1, w.deg, w.selfw, w.ptr_dst, w.end_dst, w.nbr_dst,
                         w.wn_dst, w.AH1, w.b1s);
    else
      hipLaunchKernelGGL(k_aggregate_l1<0>, ga, ba, 0, st, n_nodes, in_dim, hidden, w.AX, W1, b1, w.deg, w.selfw, w.ptr_dst, w.end_dst, w.nbr_dst,
                         w.wn_dst, w.AH1, w.b1s);
  }
  // H2 = relu(AH1 W2 + b2) * mask   (fp32 MFMA, fused epilogue)
  gemm<false, false, 1>(st, n_nodes, hidden, hidden, w.AH1, hidden, W2, hidden, w.H2, hidden, b2, dropout_mask, 1);
  if (out_dim <= kThinOut)  // one pass over H2 (HBM-bound)
    hipLaunchKernelGGL(k_linear_out, dim3((n_nodes + 3) / 4), dim3(256), 0, st, n_nodes, hidden, out_dim, w.H2, Wf, bf, out);
  else  // the critic's 100 outputs: a product for the matrix cores (k_linear_out walked H2's row once per output: 0.85 ms at 12.8 k nodes)
    gemm<false, true, 2>(st, n_nodes, out_dim, hidden, w.H2, hidden, Wf, hidden, out, out_dim, bf, nullptr, 1);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_gcn_forward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                      const int64_t *edge_index, const float *edge_attr, const float *W1, const float *b1, const float *W2,
                      const float *b2, const float *Wf, const float *bf, const float *dropout_mask, float *out, void *ws_dev) {
  return gcn_forward_impl(hip_stream, n_nodes, n_edges, in_dim, hidden, out_dim, x, edge_index, edge_attr, W1, b1, W2, b2, Wf, bf,
                          dropout_mask, out, ws_dev, 0, nullptr, nullptr, 0);
}

int drlgx_gcn_forward_batched(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                              const int64_t *edge_index, const float *edge_attr, const float *W1, const float *b1, const float *W2,
                              const float *b2, const float *Wf, const float *bf, const float *dropout_mask, float *out, void *ws_dev,
                              int n_graphs, const int32_t *node_off, const int32_t *edge_off, int max_edges_per_graph) {
  if (n_graphs <= 0 || !node_off || !edge_off || max_edges_per_graph < 0) return DRLGX_E_INVALID;
  return gcn_forward_impl(hip_stream, n_nodes, n_edges, in_dim, hidden, out_dim, x, edge_index, edge_attr, W1, b1, W2, b2, Wf, bf,
                          dropout_mask, out, ws_dev, n_graphs, node_off, edge_off, max_edges_per_graph);
}

int drlgx_gcn_forward_prebuilt(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *W1, const float *b1,
                               const float *W2, const float *b2, const float *Wf, const float *bf, const float *dropout_mask, float *out,
                               void *ws_dev) {
  return gcn_forward_impl(hip_stream, n_nodes, n_edges, in_dim, hidden, out_dim, nullptr, nullptr, nullptr, W1, b1, W2, b2, Wf, bf, dropout_mask,
                          out, ws_dev, kPrebuilt, nullptr, nullptr, 0);
}

static bool cache_ok(const drlgx_csr_cache *c) {
  return c && c->deg && c->selfw && c->ax && c->ptr_dst && c->end_dst && c->ptr_src && c->end_src && c->nbr_dst && c->nbr_src && c->wn_dst &&
         c->wn_src;
}

int drlgx_replay_cache_csr(void *hip_stream, int n_graphs, const int32_t *node_off, const int32_t *edge_off, int max_edges_per_graph,
                           const float *x, int in_dim, const int64_t *edge_index, int64_t edge_row_stride, const float *edge_attr,
                           const drlgx_csr_cache *cache) {
  if (n_graphs <= 0 || !node_off || !edge_off || max_edges_per_graph < 0 || !x || in_dim <= 0 || in_dim > 8 || !edge_index || !edge_attr ||
      edge_row_stride <= 0 || edge_row_stride >= (1ll << 31) || !cache_ok(cache))
    return DRLGX_E_INVALID;
  if (max_edges_per_graph > kCsrMaxEdges) return DRLGX_E_CAPACITY;  // (the caller keeps such an export uncached)
  int P2 = 64;
  while (P2 < max_edges_per_graph) P2 <<= 1;
  const int extra = P2 <= 4096 ? 1 : 0;
  const size_t lds = (size_t)(extra ? 4 : 2) * P2 * sizeof(uint32_t);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&k_csr_graphs)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, 160 * 1024);
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  // (E = the row stride of edge_index: the kernel reads the second row at ei[E + e])
  hipLaunchKernelGGL(k_csr_graphs, dim3(n_graphs), dim3(256), lds, st, 0, (int)edge_row_stride, P2, extra, edge_index, edge_attr, node_off, edge_off,
                     cache->deg, cache->selfw, cache->ptr_dst, cache->end_dst, cache->nbr_dst, cache->wn_dst, cache->ptr_src, cache->end_src,
                     cache->nbr_src, cache->wn_src, x, in_dim, cache->ax, 1);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_gcn_collate_csr(void *hip_stream, int n_graphs, const int64_t *desc_dev, const drlgx_csr_cache *cache, int n_nodes, int n_edges, int hidden,
                          int out_dim, void *ws_dev, int32_t *node_off_out, int32_t *edge_off_out, const int64_t *desc2_dev, const float *pool_q,
                          float *q2_out) {
  if (n_graphs <= 0 || !desc_dev || !cache_ok(cache) || n_nodes <= 0 || n_edges < 0 || hidden <= 0 || out_dim <= 0 || !ws_dev || !node_off_out ||
      !edge_off_out || (desc2_dev && (!pool_q || !q2_out)))
    return DRLGX_E_INVALID;
  GcnWs w;
  carve(&w, reinterpret_cast<char *>(ws_dev), n_nodes, std::max(n_edges, 1), hidden, out_dim);
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_csr_collate, dim3(desc2_dev ? 2 * n_graphs : n_graphs), dim3(256), 0, st, n_graphs, desc_dev, *cache, w.deg, w.selfw, w.AX,
                     w.ptr_dst, w.end_dst, w.ptr_src, w.end_src, w.nbr_dst, w.nbr_src, w.wn_dst, w.wn_src, node_off_out, edge_off_out, desc2_dev,
                     pool_q, q2_out);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_gcn_backward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *x,
                       const int64_t *edge_index, const float *edge_attr, const float *W1, const float *W2, const float *Wf,
                       const float *dropout_mask, const float *d_out, float *dW1, float *db1, float *dW2, float *db2, float *dWf,
                       float *dbf, void *ws_dev) {
  if (n_nodes <= 0 || in_dim <= 0 || in_dim > 8 || hidden <= 0 || (hidden & 3) || out_dim <= 0 || !d_out || !dW1 || !db1 || !dW2 ||
      !db2 || !dWf || !dbf || !ws_dev || !W1 || !W2 || !Wf)
    return DRLGX_E_INVALID;
  (void)x; (void)edge_index; (void)edge_attr; (void)n_edges;  // the forward left AX / b1 / AH1 / H2 and both CSRs in ws
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  GcnWs w;
  carve(&w, reinterpret_cast<char *>(ws_dev), n_nodes, std::max(n_edges, 1), hidden, out_dim);
  // output layer
  const uintptr_t al16 = reinterpret_cast<uintptr_t>(Wf) | reinterpret_cast<uintptr_t>(dropout_mask) | reinterpret_cast<uintptr_t>(w.H2) |
                         reinterpret_cast<uintptr_t>(w.T0);
  if (out_dim <= kThinOut && (al16 & 15) == 0) {
    // dWf = dOut^T H2m, dbf = colsum(dOut), T0 = dZ2, db2 = colsum(dZ2): one pass over H2 and one reduce
    int nb = std::min(128, (n_nodes + 7) / 8);
    nb = (int)std::max<size_t>(1, std::min<size_t>(nb, w.part_floats / ((size_t)(out_dim + 1) * hidden)));
    const int rpb = (n_nodes + nb - 1) / nb;
    nb = (n_nodes + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_dz2_sums, dim3((hidden / 4 + 63) / 64, nb), dim3(256), 0, st, n_nodes, hidden, out_dim, d_out, Wf, dropout_mask, w.H2, w.T0,
                       w.part, rpb);
    hipLaunchKernelGGL(k_thin_tn_reduce, dim3(((out_dim + 1) * hidden + 63) / 64 + 1), dim3(1024), 0, st, hidden, out_dim, nb, w.part, dWf, out_dim,
                       db2, d_out, out_dim, n_nodes, dbf);
  } else {
    if (out_dim <= kThinOut) {
      thin_tn(st, w, out_dim, hidden, n_nodes, d_out, out_dim, w.H2, hidden, dWf, out_dim, nullptr, dbf);  // dWf = dOut^T H2m, dbf = colsum(dOut)
    } else {
      gemm_tn_splitk(st, w, out_dim, hidden, n_nodes, d_out, out_dim, w.H2, hidden, dWf, 32);
      colsum(st, w, n_nodes, out_dim, d_out, dbf);
    }
    // T0 = dZ2 = (dOut Wf) * gate
    if (out_dim <= kThinOut)
      hipLaunchKernelGGL(k_dz2, dim3(n_nodes), dim3(256), 0, st, n_nodes, hidden, out_dim, d_out, Wf, dropout_mask, w.H2, w.T0);
    else
      gemm<false, false, 3>(st, n_nodes, hidden, out_dim, d_out, out_dim, Wf, hidden, w.T0, hidden, w.H2, dropout_mask, 1);
    colsum(st, w, n_nodes, hidden, w.T0, db2);
  }
  // layer 2
  gemm_tn_splitk(st, w, hidden, hidden, n_nodes, w.AH1, hidden, w.T0, hidden, dW2);  // dW2 = AH1^T dZ2
  gemm<false, true, 0>(st, n_nodes, hidden, hidden, w.T0, hidden, W2, hidden, w.T1, hidden, nullptr, nullptr, 1);  // T1 = dZ2 W2^T
  // dZ1 = (Â^T dAH1) * (H1 > 0)   -> T0
  hipLaunchKernelGGL(k_aggregate<true>, dim3(n_nodes), dim3(256), 0, st, n_nodes, hidden, w.T1, w.deg, w.selfw, w.ptr_src, w.end_src, w.nbr_src,
                     w.wn_src, in_dim, w.AX, W1, w.b1s, w.T0);
  // layer 1
  thin_tn(st, w, 8, hidden, n_nodes, w.AX, 8, w.T0, hidden, dW1, in_dim, db1);  // dW1 = AX^T dZ1 (AX rows are 8 wide), db1 = colsum(dZ1)
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

}  // extern "C"

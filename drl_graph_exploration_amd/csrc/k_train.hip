// Trainer-side kernels of the DQN update (scripts/policy.py:137-178 and :234-253): mini-batch collation of replay graphs
// out of the device pool, the TD targets, the squared-error cost with its gradient, and the clamp + Adam step.
// Everything here replaces a handful of tiny framework ops each (and, in the reference, a device -> host -> device round
// trip per update: `readout_j1_batch.cpu()`, numpy target loop, `torch.tensor(y)`), so the whole update is a fixed
// sequence of kernel launches with no host synchronisation.  All of it is HBM-bound byte / elementwise work: one pass
// over the data, coalesced.
#include <cmath>
#include <cstdint>

#include "drlgx_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// torch_geometric DataLoader / Batch.from_data_list over graphs stored in a pool (policy.py:146-153): graph g of the
// mini-batch is rows [node_start, +node_cnt) of pool_x and columns [edge_start, +edge_cnt) of pool_ei / pool_ea, with
// node ids relative to its export (first id = loc).  Output = PyG batch: x, edge_index shifted by the cumulative node
// counts, edge_attr, batch vector.  One workgroup per graph; its output offsets are the sums over the earlier graphs.
// desc: int64 [5][G] = node_start, node_cnt, edge_start, edge_cnt, loc.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_replay_collate(int G, const int64_t *desc, const float *pool_x, int in_dim, const int64_t *pool_ei,
                                                        int64_t pool_edges, const float *pool_ea, float *x_out, int64_t *ei_out,
                                                        int64_t E_total, float *ea_out, int64_t *batch_out, int *node_off_out,
                                                        int *edge_off_out, const float *pool_q, float *q_out, const int64_t *desc2,
                                                        float *q2_out) {
  __shared__ long long red[2][4];
  const int tid = threadIdx.x;
  int g = blockIdx.x;
  if (g >= G) {  // the second list (drlgx_replay_collate_pair): only the per-node value of its graphs is gathered
    g -= G;
    desc = desc2;
    x_out = nullptr;
    node_off_out = edge_off_out = nullptr;
    q_out = q2_out;
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
  const long long n0 = desc[g], nn = desc[(size_t)G + g], e0 = desc[2 * (size_t)G + g], ne = desc[3 * (size_t)G + g],
                  loc = desc[4 * (size_t)G + g];
  if (tid == 0 && node_off_out && edge_off_out) {
    node_off_out[g] = (int)node_off;
    edge_off_out[g] = (int)edge_off;
    if (g == G - 1) {
      node_off_out[G] = (int)(node_off + nn);
      edge_off_out[G] = (int)(edge_off + ne);
    }
  }
  if (x_out) {  // (NULL: only the per-node cache is gathered)
    const float *xs = pool_x + n0 * in_dim;
    float *xd = x_out + node_off * in_dim;
    for (long long i = tid; i < nn * in_dim; i += 256) xd[i] = xs[i];
    for (long long i = tid; i < nn; i += 256) batch_out[node_off + i] = g;
  }
  if (pool_q && q_out)  // a per-node value stored beside the pooled graphs (the target network's cached read-out)
    for (long long i = tid; i < nn; i += 256) q_out[node_off + i] = pool_q[n0 + i];
  const long long shift = node_off - loc;
  for (long long j = tid; x_out && j < ne; j += 256) {
    ei_out[edge_off + j] = pool_ei[e0 + j] + shift;
    ei_out[E_total + edge_off + j] = pool_ei[pool_edges + e0 + j] + shift;
    ea_out[edge_off + j] = pool_ea[e0 + j];
  }
}

// ------------------------------------------------------------------------------------------------
// TD targets (policy.py:154-175).  Sample i reads the target network's read-out over the window [lo_i, hi_i) of q1 (the
// host resolves the reference's slicing rule into these bounds), takes its maximum in float32 like np.max over the
// float32 read-out, and writes  a_batch[pos_i] = 1,  y_batch[pos_i] = r_i (+ gamma max_q unless terminal)  in float64 into
// the zero-initialised vectors over the current-state nodes.  meta: int64 [4][B] = lo, hi, pos, terminal.
// ------------------------------------------------------------------------------------------------
// One workgroup: it zeroes both vectors first (two memset launches less per update), then sample i < B writes its entry.
__global__ __launch_bounds__(1024) void k_dqn_targets(int B, const float *q1, const int64_t *meta, const double *r, double gamma,
                                                      long long n_total, double *a_batch, double *y_batch) {
  for (long long k = threadIdx.x; k < n_total; k += 1024) {
    a_batch[k] = 0.0;
    y_batch[k] = 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += 1024) {
    const long long lo = meta[i], hi = meta[(size_t)B + i], pos = meta[2 * (size_t)B + i];
    const bool term = meta[3 * (size_t)B + i] != 0;
    double t = r[i];
    if (!term) {
      float m = -INFINITY;
      for (long long k = lo; k < hi; ++k) m = fmaxf(m, q1[k]);
      t = r[i] + gamma * (double)m;
    }
    a_batch[pos] = 1.0;
    y_batch[pos] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// DeepQ.cost (policy.py:234-239) and its gradient with respect to the read-out:
//   loss = sum_i (pred_i * a_i - y_i)^2 / batch   in float64 (y / action are float64 in the reference, the float32
//   read-out is promoted);  d_pred_i = float32( 2 (pred_i a_i - y_i) a_i / batch ).
// One workgroup (N is the node count of a mini-batch, a few thousand).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_dqn_loss_grad(int N, const float *pred, const double *action, const double *y, double batch,
                                                        double *loss_out, float *d_pred) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int i = tid; i < N; i += 1024) {
    const double a = action[i];
    const double e = (double)pred[i] * a - y[i];
    acc += e * e;
    d_pred[i] = (float)((2.0 * e) * (1.0 / batch) * a);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w];
    loss_out[0] = s / batch;
  }
}

// ------------------------------------------------------------------------------------------------
// `param.grad.data.clamp_(-c, c)` for every parameter followed by torch.optim.Adam.step() (policy.py:250-253; Adam with
// its defaults, no weight decay, no amsgrad), all tensors of the model in one launch.  The arithmetic follows torch's
// single-tensor Adam in float32:  m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g g;
// p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)   (the scalars are formed on the host in double).
// ------------------------------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 8;
struct AdamTensors {
  float *p[kAdamMaxTensors];
  const float *g[kAdamMaxTensors];
  float *m[kAdamMaxTensors];
  float *v[kAdamMaxTensors];
  long long first_block[kAdamMaxTensors + 1];  // blocks of 1024 elements, prefix over the tensors
  long long n[kAdamMaxTensors];
  int count;
};
__global__ __launch_bounds__(256) void k_adam(AdamTensors T, float gscale, float clamp, float b1w, float b2, float b2w, float neg_step,
                                              float bc2_sqrt, float eps) {
  const long long blk = blockIdx.x;
  int t = 0;
  while (t + 1 < T.count && blk >= T.first_block[t + 1]) ++t;
  const long long base = (blk - T.first_block[t]) * 1024;
  float *p = T.p[t], *m = T.m[t], *v = T.v[t];
  const float *g = T.g[t];
  const long long n = T.n[t];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + u * 256 + threadIdx.x;
    if (i < n) {
      float gi = g[i] * gscale;  // (1 / world size of the summed gradient; 1 otherwise: exact)
      if (clamp > 0.f) gi = fminf(fmaxf(gi, -clamp), clamp);
      const float mi = m[i] + b1w * (gi - m[i]);
      const float vi = v[i] * b2 + b2w * gi * gi;
      m[i] = mi;
      v[i] = vi;
      p[i] = p[i] + neg_step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ExplorationEnv.rewards_all_goals, the normalisation (exploration_env.py:151-161): per environment the look-ahead rewards of
// its frontiers [first_e, first_e + n_e) are mapped by np.interp from [min, max] onto [-1, 0] when the vehicle's nearest
// frontier (the first one) is the (first) arg-max - loop_clo False -, else onto [-1, 1] - loop_clo True.  One wave per env.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_normalise_rewards(const double *raw, const int64_t *first, const int *n_frontier, double *out,
                                                          uint8_t *loop_clo) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const long long f0 = first[e];
  const int nf = n_frontier[e];
  double lo = INFINITY, hi = -INFINITY;
  for (int i = lane; i < nf; i += 64) {
    const double v = raw[f0 + i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
  }
  if (nf <= 0) {
    if (lane == 0) loop_clo[e] = 0;
    return;
  }
  const bool loop = raw[f0] < hi;  // np.nanargmax returns the first maximum
  const double top = loop ? 1.0 : 0.0, span = hi - lo;
  const double slope = (top + 1.0) / (span > 0 ? span : 1.0);
  for (int i = lane; i < nf; i += 64) {
    const double v = raw[f0 + i];
    out[f0 + i] = v >= hi ? top : slope * (v - lo) - 1.0;  // np.interp: slope * (x - xp[0]) + fp[0]; x >= xp[-1] -> fp[-1]
  }
  if (lane == 0) loop_clo[e] = loop ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// PolicyGCN head (scripts/Networks.py:47-50): masked_select of the per-node read-out and of `batch`, then
// torch_geometric.utils.softmax over every graph's selected nodes:  p = exp(q - max) / (sum exp(q - max) + 1e-16).
// One wave per graph (graph boundaries node_off); the selected nodes of all graphs are written in node order, so a
// graph's output offset is the number of selected nodes before its first node.
// Backward:  dq_j = p_j (dp_j - sum_i p_i dp_i)  for the selected nodes, 0 elsewhere.
// ------------------------------------------------------------------------------------------------
// number of non-zero bytes of a word
__device__ __forceinline__ int nz_bytes(unsigned w) {
  w |= w >> 4;
  w |= w >> 2;
  w |= w >> 1;
  return __popc(w & 0x01010101u);
}
// selected nodes in front of node n0: 16 mask bytes per lane and trip (a byte per lane and trip walked the 22 k nodes in front of a
// 256-graph batch's last graphs in 344 dependent trips: 59 us for the kernel)
__device__ __forceinline__ int masked_before(const uint8_t *mask, int n0, int lane) {
  int c = 0;
  const int head = min(n0, (int)((16 - (reinterpret_cast<uintptr_t>(mask) & 15)) & 15));  // (a chunk's mask is a slice: any alignment)
  if (lane < head) c += mask[lane] ? 1 : 0;
  const int nv = (n0 - head) >> 4;
  const uint4 *v = reinterpret_cast<const uint4 *>(mask + head);
  for (int i = lane; i < nv; i += 64) {
    const uint4 w = v[i];
    c += nz_bytes(w.x) + nz_bytes(w.y) + nz_bytes(w.z) + nz_bytes(w.w);
  }
  for (int i = head + 16 * nv + lane; i < n0; i += 64) c += mask[i] ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  return c;
}
__global__ __launch_bounds__(64) void k_segment_softmax(const float *q, const uint8_t *mask, const int *node_off, float *p_out) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n0 = node_off[g], n1 = node_off[g + 1];
  int k = masked_before(mask, n0, lane);
  float m = -INFINITY;
  for (int i = n0 + lane; i < n1; i += 64)
    if (mask[i]) m = fmaxf(m, q[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int i = n0 + lane; i < n1; i += 64)
    if (mask[i]) s += expf(q[i] - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / (s + 1e-16f);
  for (int b = n0; b < n1; b += 64) {
    const int i = b + lane;
    const bool on = i < n1 && mask[i];
    const unsigned long long bal = __ballot(on);
    if (on) p_out[k + __popcll(bal & ((1ull << lane) - 1ull))] = expf(q[i] - m) * inv;
    k += __popcll(bal);
  }
}
__global__ __launch_bounds__(64) void k_segment_softmax_bwd(const float *p, const float *dp, const uint8_t *mask, const int *node_off,
                                                            float *dq) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n0 = node_off[g], n1 = node_off[g + 1];
  const int k0 = masked_before(mask, n0, lane);
  int cnt = 0;
  for (int i = n0 + lane; i < n1; i += 64) cnt += mask[i] ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  float dot = 0.f;
  for (int j = lane; j < cnt; j += 64) dot += p[k0 + j] * dp[k0 + j];
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  int k = k0;
  for (int b = n0; b < n1; b += 64) {
    const int i = b + lane;
    const bool on = i < n1 && mask[i];
    const unsigned long long bal = __ballot(on);
    if (i < n1) {
      float v = 0.f;
      if (on) {
        const int j = k + __popcll(bal & ((1ull << lane) - 1ull));
        v = p[j] * (dp[j] - dot);
      }
      dq[i] = v;
    }
    k += __popcll(bal);
  }
}

// ------------------------------------------------------------------------------------------------
// ValueGCN head (scripts/Networks.py:66-70): global_mean_pool(x, batch).mean(dim=1) over the [N, C] read-out = per graph
// the mean over its nodes of every column, then the mean over the columns.  One workgroup per graph.
// Backward: dh[n][c] = dv[g] / (n_g C).
// ------------------------------------------------------------------------------------------------
// (A column per thread walked the graph's nodes one dependent load after the other, 100 of 256 threads at work: 37-46 us for 256
// graphs.  Now 128 columns x 2 row groups per pass, four rows of a group in flight; the partial sums are combined in a fixed order.)
__global__ __launch_bounds__(256) void k_mean_pool(const float *h, int C, const int *node_off, float *v_out) {
  __shared__ float red[4], half[128];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n0 = node_off[g], n1 = node_off[g + 1];
  const int cl = tid & 127, r = tid >> 7;
  float acc = 0.f;
  for (int c0 = 0; c0 < C; c0 += 128) {
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
      const float *hc = h + c;
      int n = n0 + r;
      for (; n + 6 < n1; n += 8) {
        s0 += hc[(size_t)n * C];
        s1 += hc[(size_t)(n + 2) * C];
        s2 += hc[(size_t)(n + 4) * C];
        s3 += hc[(size_t)(n + 6) * C];
      }
      for (; n < n1; n += 2) s0 += hc[(size_t)n * C];
    }
    const float s = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r == 1) half[cl] = s;
    __syncthreads();
    if (r == 0 && c < C) acc += (s + half[cl]) / (float)max(n1 - n0, 1);  // the column's mean over the graph's nodes
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) v_out[g] = (red[0] + red[1] + red[2] + red[3]) / (float)C;
}
__global__ __launch_bounds__(256) void k_mean_pool_bwd(const float *dv, int C, const int *node_off, float *dh) {
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n0 = node_off[g], n1 = node_off[g + 1];
  const float v = dv[g] / ((float)max(n1 - n0, 1) * (float)C);
  for (long long e = tid; e < (long long)(n1 - n0) * C; e += 256) dh[(size_t)n0 * C + e] = v;
}

}  // namespace

extern "C" {

int drlgx_normalise_rewards(void *hip_stream, int n_envs, const double *raw, const int64_t *cand_first, const int32_t *n_frontier,
                            double *out, uint8_t *loop_clo) {
  if (n_envs <= 0 || !raw || !cand_first || !n_frontier || !out || !loop_clo) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_normalise_rewards, dim3(n_envs), dim3(64), 0, st, raw, cand_first, n_frontier, out, loop_clo);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_segment_softmax(void *hip_stream, int n_graphs, const int32_t *node_off, const float *q, const uint8_t *mask, float *p_out) {
  if (n_graphs <= 0 || !node_off || !q || !mask || !p_out) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_segment_softmax, dim3(n_graphs), dim3(64), 0, st, q, mask, node_off, p_out);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_segment_softmax_backward(void *hip_stream, int n_graphs, const int32_t *node_off, const float *p, const float *d_p,
                                   const uint8_t *mask, float *d_q) {
  if (n_graphs <= 0 || !node_off || !p || !d_p || !mask || !d_q) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_segment_softmax_bwd, dim3(n_graphs), dim3(64), 0, st, p, d_p, mask, node_off, d_q);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_mean_pool(void *hip_stream, int n_graphs, const int32_t *node_off, const float *h, int n_cols, float *v_out) {
  if (n_graphs <= 0 || !node_off || !h || n_cols <= 0 || !v_out) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_mean_pool, dim3(n_graphs), dim3(256), 0, st, h, n_cols, node_off, v_out);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_mean_pool_backward(void *hip_stream, int n_graphs, const int32_t *node_off, const float *d_v, int n_cols, float *d_h) {
  if (n_graphs <= 0 || !node_off || !d_v || n_cols <= 0 || !d_h) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_mean_pool_bwd, dim3(n_graphs), dim3(256), 0, st, d_v, n_cols, node_off, d_h);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}


int drlgx_replay_collate(void *hip_stream, int n_graphs, const int64_t *desc_dev, const float *pool_x, int in_dim, const int64_t *pool_ei,
                         int64_t pool_edges, const float *pool_ea, float *x_out, int64_t *ei_out, int64_t n_edges_total, float *ea_out,
                         int64_t *batch_out, int32_t *node_off_out, int32_t *edge_off_out, const float *pool_q, float *q_out) {
  const bool graphs = x_out != nullptr;
  if (n_graphs <= 0 || !desc_dev || !pool_x || in_dim <= 0 || !pool_ei || !pool_ea || n_edges_total < 0 || pool_edges < 0 ||
      (graphs ? (!ei_out || !ea_out || !batch_out) : (!pool_q || !q_out)))
    return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_replay_collate, dim3(n_graphs), dim3(256), 0, st, n_graphs, desc_dev, pool_x, in_dim, pool_ei, pool_edges, pool_ea,
                     x_out, ei_out, n_edges_total, ea_out, batch_out, node_off_out, edge_off_out, pool_q, q_out, nullptr, nullptr);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_replay_collate_pair(void *hip_stream, int n_graphs, const int64_t *desc_dev, const float *pool_x, int in_dim, const int64_t *pool_ei,
                              int64_t pool_edges, const float *pool_ea, float *x_out, int64_t *ei_out, int64_t n_edges_total, float *ea_out,
                              int64_t *batch_out, int32_t *node_off_out, int32_t *edge_off_out, const int64_t *desc2_dev, const float *pool_q,
                              float *q2_out) {
  if (n_graphs <= 0 || !desc_dev || !pool_x || in_dim <= 0 || !pool_ei || !pool_ea || n_edges_total < 0 || pool_edges < 0 || !x_out ||
      !ei_out || !ea_out || !batch_out || !desc2_dev || !pool_q || !q2_out)
    return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_replay_collate, dim3(2 * n_graphs), dim3(256), 0, st, n_graphs, desc_dev, pool_x, in_dim, pool_ei, pool_edges, pool_ea,
                     x_out, ei_out, n_edges_total, ea_out, batch_out, node_off_out, edge_off_out, pool_q, nullptr, desc2_dev, q2_out);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_dqn_targets(void *hip_stream, int n_samples, const float *q1, const int64_t *meta_dev, const double *r_dev, double gamma,
                      int64_t n_nodes_total, double *a_batch, double *y_batch) {
  if (n_samples <= 0 || !q1 || !meta_dev || !r_dev || n_nodes_total <= 0 || !a_batch || !y_batch) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_dqn_targets, dim3(1), dim3(1024), 0, st, n_samples, q1, meta_dev, r_dev, gamma, (long long)n_nodes_total, a_batch, y_batch);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_dqn_loss_grad(void *hip_stream, int n_nodes, const float *pred, const double *action, const double *y, double batch,
                        double *loss_out, float *d_pred) {
  if (n_nodes <= 0 || !pred || !action || !y || !(batch > 0) || !loss_out || !d_pred) return DRLGX_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_dqn_loss_grad, dim3(1), dim3(1024), 0, st, n_nodes, pred, action, y, batch, loss_out, d_pred);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_adam_step_scaled(void *hip_stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                           float *const *exp_avg_sq, const int64_t *sizes, double lr, double beta1, double beta2, double eps, int64_t step,
                           double grad_clamp, double grad_scale) {
  if (n_tensors <= 0 || n_tensors > kAdamMaxTensors || !params || !grads || !exp_avg || !exp_avg_sq || !sizes || step <= 0)
    return DRLGX_E_INVALID;
  AdamTensors T;
  T.count = n_tensors;
  long long blocks = 0;
  for (int t = 0; t < n_tensors; ++t) {
    if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || sizes[t] <= 0) return DRLGX_E_INVALID;
    T.p[t] = params[t];
    T.g[t] = grads[t];
    T.m[t] = exp_avg[t];
    T.v[t] = exp_avg_sq[t];
    T.n[t] = sizes[t];
    T.first_block[t] = blocks;
    blocks += (sizes[t] + 1023) / 1024;
  }
  T.first_block[n_tensors] = blocks;
  for (int t = n_tensors; t < kAdamMaxTensors; ++t) {
    T.p[t] = nullptr; T.g[t] = nullptr; T.m[t] = nullptr; T.v[t] = nullptr; T.n[t] = 0;
    T.first_block[t + 1] = blocks;
  }
  // torch/optim/adam.py (_single_tensor_adam): python-float scalars, float32 tensor arithmetic
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  const double step_size = lr / bc1, bc2_sqrt = std::sqrt(bc2);
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, st, T, (float)grad_scale, (float)grad_clamp, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)(-step_size), (float)bc2_sqrt, (float)eps);
  return hipGetLastError() == hipSuccess ? DRLGX_OK : DRLGX_E_HIP;
}

int drlgx_adam_step(void *hip_stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                    float *const *exp_avg_sq, const int64_t *sizes, double lr, double beta1, double beta2, double eps, int64_t step,
                    double grad_clamp) {
  return drlgx_adam_step_scaled(hip_stream, n_tensors, params, grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, step, grad_clamp, 1.0);
}

}  // extern "C"

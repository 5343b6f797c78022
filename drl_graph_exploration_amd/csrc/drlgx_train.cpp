// Host side of one DQN update as two C-ABI calls (include/drlgx.h: drlgx_dqn_prepare, drlgx_dqn_forward_backward): the launches of
// scripts/policy.py:139-178 (mini-batch collation, TD targets) and :234-249 (forward, cost, backward) issued back to back from
// C++ on the caller's stream, with every intermediate in ONE caller-owned arena.  Nothing new runs on the device - these are the
// kernels of drlgx_replay_collate_pair / drlgx_dqn_targets / drlgx_gcn_forward_batched / drlgx_dqn_loss_grad /
// drlgx_gcn_backward; what goes away is the per-launch host work of the Python layer (a dozen tensor allocations and five
// foreign calls per update: the 256-env DQN loop was host-bound, 72 ms per vector step against ~49 ms of kernels).
#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "../../include/drlgx.h"

namespace {

size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Arena {
  float *x, *ea, *q1, *out, *d_out;
  int64_t *ei, *bt;
  int32_t *node_off, *edge_off;
  double *a_batch, *y_batch, *loss;
  void *gcn_ws;
  size_t bytes;
};

// the layout depends on the capacities only, so that both calls of an update (and the caller, for read-backs) agree on it
Arena carve(char *base, int k, int64_t cap_n, int64_t cap_e, int64_t cap_n1, int in_dim, int hidden, int out_dim) {
  Arena a{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *p = base ? base + off : nullptr;
    off += up256(bytes);
    return p;
  };
  a.x = reinterpret_cast<float *>(take((size_t)cap_n * in_dim * 4));
  a.ei = reinterpret_cast<int64_t *>(take((size_t)cap_e * 2 * 8));
  a.ea = reinterpret_cast<float *>(take((size_t)cap_e * 4));
  a.bt = reinterpret_cast<int64_t *>(take((size_t)cap_n * 8));
  a.node_off = reinterpret_cast<int32_t *>(take((size_t)(k + 1) * 4));
  a.edge_off = reinterpret_cast<int32_t *>(take((size_t)(k + 1) * 4));
  a.q1 = reinterpret_cast<float *>(take((size_t)std::max<int64_t>(cap_n1, 1) * 4));
  a.a_batch = reinterpret_cast<double *>(take((size_t)cap_n * 8));
  a.y_batch = reinterpret_cast<double *>(take((size_t)cap_n * 8));
  a.out = reinterpret_cast<float *>(take((size_t)cap_n * out_dim * 4));
  a.d_out = reinterpret_cast<float *>(take((size_t)cap_n * out_dim * 4));
  a.loss = reinterpret_cast<double *>(take(8));
  a.gcn_ws = take(drlgx_gcn_workspace_bytes((int)cap_n, (int)cap_e, hidden, out_dim));
  a.bytes = off;
  return a;
}

bool caps_ok(int k, int64_t cap_n, int64_t cap_e, int64_t cap_n1, int in_dim, int hidden, int out_dim) {
  return k > 0 && cap_n > 0 && cap_e >= 0 && cap_n1 >= 0 && cap_n < (1ll << 31) && cap_e < (1ll << 31) && in_dim > 0 && hidden > 0 && out_dim > 0;
}

}  // namespace

extern "C" {

size_t drlgx_dqn_arena_bytes(int n_graphs, int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int in_dim, int hidden, int out_dim) {
  if (!caps_ok(n_graphs, cap_nodes, cap_edges, cap_nodes1, in_dim, hidden, out_dim)) return 0;
  return carve(nullptr, n_graphs, cap_nodes, std::max<int64_t>(cap_edges, 1), cap_nodes1, in_dim, hidden, out_dim).bytes;
}

int drlgx_dqn_arena_views(void *arena_dev, int n_graphs, int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int in_dim, int hidden,
                          int out_dim, void **views /* [12] */) {
  if (!arena_dev || !views || !caps_ok(n_graphs, cap_nodes, cap_edges, cap_nodes1, in_dim, hidden, out_dim)) return DRLGX_E_INVALID;
  const Arena a = carve(static_cast<char *>(arena_dev), n_graphs, cap_nodes, std::max<int64_t>(cap_edges, 1), cap_nodes1, in_dim, hidden, out_dim);
  void *v[12] = {a.x, a.ei, a.ea, a.bt, a.node_off, a.edge_off, a.q1, a.a_batch, a.y_batch, a.out, a.d_out, a.loss};
  for (int i = 0; i < 12; ++i) views[i] = v[i];
  return DRLGX_OK;
}

int drlgx_dqn_prepare(void *hip_stream, int n_graphs, const int64_t *desc_dev, const int64_t *desc1_dev, const float *pool_x, int in_dim,
                      const int64_t *pool_ei, int64_t pool_edges, const float *pool_ea, const float *pool_q, int64_t n_nodes,
                      int64_t n_edges, int64_t n_nodes1, const int64_t *meta_dev, const double *r_dev, double gamma, void *arena_dev,
                      int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int hidden, int out_dim, const drlgx_csr_cache *cache) {
  if (!arena_dev || !desc_dev || !desc1_dev || (!pool_x && !cache) || !pool_q || !meta_dev || !r_dev ||
      !caps_ok(n_graphs, cap_nodes, cap_edges, cap_nodes1, in_dim, hidden, out_dim) || n_nodes <= 0 || n_nodes > cap_nodes || n_edges < 0 ||
      n_edges > cap_edges || n_nodes1 < 0 || n_nodes1 > cap_nodes1)
    return DRLGX_E_INVALID;
  const Arena a = carve(static_cast<char *>(arena_dev), n_graphs, cap_nodes, std::max<int64_t>(cap_edges, 1), cap_nodes1, in_dim, hidden, out_dim);
  // with the pool's per-graph cache the collation fills the graph part of the GCN workspace (the forward then builds nothing
  // and never reads x / edge_index / edge_attr); the workspace is carved for the ACTUAL counts, as the forward / backward do
  int rc = cache ? drlgx_gcn_collate_csr(hip_stream, n_graphs, desc_dev, cache, (int)n_nodes, (int)n_edges, hidden, out_dim, a.gcn_ws, a.node_off,
                                         a.edge_off, desc1_dev, pool_q, a.q1)
                 : drlgx_replay_collate_pair(hip_stream, n_graphs, desc_dev, pool_x, in_dim, pool_ei, pool_edges, pool_ea, a.x, a.ei, n_edges, a.ea,
                                             a.bt, a.node_off, a.edge_off, desc1_dev, pool_q, a.q1);
  if (rc) return rc;
  return drlgx_dqn_targets(hip_stream, n_graphs, a.q1, meta_dev, r_dev, gamma, n_nodes, a.a_batch, a.y_batch);
}

int drlgx_dqn_forward_backward(void *hip_stream, int n_graphs, int64_t n_nodes, int64_t n_edges, int max_edges_per_graph, int in_dim,
                               int hidden, int out_dim, const float *const *params /* W1 b1 W2 b2 Wf bf */, const float *dropout_mask,
                               double batch, float *const *grads /* dW1 db1 dW2 db2 dWf dbf */, void *arena_dev, int64_t cap_nodes,
                               int64_t cap_edges, int64_t cap_nodes1, int graph_prebuilt) {
  if (!arena_dev || !params || !grads || !caps_ok(n_graphs, cap_nodes, cap_edges, cap_nodes1, in_dim, hidden, out_dim) || n_nodes <= 0 ||
      n_nodes > cap_nodes || n_edges < 0 || n_edges > cap_edges || out_dim != 1)
    return DRLGX_E_INVALID;
  for (int i = 0; i < 6; ++i)
    if (!params[i] || !grads[i]) return DRLGX_E_INVALID;
  const Arena a = carve(static_cast<char *>(arena_dev), n_graphs, cap_nodes, std::max<int64_t>(cap_edges, 1), cap_nodes1, in_dim, hidden, out_dim);
  const int N = (int)n_nodes, E = (int)n_edges;
  int rc = graph_prebuilt
               ? drlgx_gcn_forward_prebuilt(hip_stream, N, E, in_dim, hidden, out_dim, params[0], params[1], params[2], params[3], params[4],
                                            params[5], dropout_mask, a.out, a.gcn_ws)
               : drlgx_gcn_forward_batched(hip_stream, N, E, in_dim, hidden, out_dim, a.x, a.ei, a.ea, params[0], params[1], params[2], params[3],
                                           params[4], params[5], dropout_mask, a.out, a.gcn_ws, n_graphs, a.node_off, a.edge_off,
                                           max_edges_per_graph);
  if (rc) return rc;
  rc = drlgx_dqn_loss_grad(hip_stream, N, a.out, a.a_batch, a.y_batch, batch, a.loss, a.d_out);
  if (rc) return rc;
  return drlgx_gcn_backward(hip_stream, N, E, in_dim, hidden, out_dim, a.x, a.ei, a.ea, params[0], params[2], params[4], dropout_mask, a.d_out,
                            grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], a.gcn_ws);
}

}  // extern "C"

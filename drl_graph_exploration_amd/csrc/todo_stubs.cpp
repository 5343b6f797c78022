// TEMPORARY: entry points not implemented yet (graph export, GCN) — replaced by k_graph.hip / k_gcn.hip.
#include "drlgx_dev.h"
extern "C" {
int drlgx_graph_capacity(const drlgx_engine *, int *, int *, int *) { return DRLGX_E_INVALID; }
int drlgx_graph(drlgx_engine *, int32_t *, int32_t *, float *, int64_t *, float *, int32_t *, double *, int32_t *) { return DRLGX_E_INVALID; }
size_t drlgx_gcn_workspace_bytes(int, int, int, int) { return 0; }
int drlgx_gcn_forward(void *, int, int, int, int, int, const float *, const int64_t *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, void *) { return DRLGX_E_INVALID; }
int drlgx_gcn_backward(void *, int, int, int, int, int, const float *, const int64_t *, const float *, const float *, const float *, const float *, const float *, const float *, float *, float *, float *, float *, float *, float *, void *) { return DRLGX_E_INVALID; }
}

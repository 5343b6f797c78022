// TEMPORARY until k_gcn.hip lands
#include "drlgx_dev.h"
extern "C" {
size_t drlgx_gcn_workspace_bytes(int, int, int, int) { return 0; }
int drlgx_gcn_forward(void *, int, int, int, int, int, const float *, const int64_t *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, void *) { return DRLGX_E_INVALID; }
int drlgx_gcn_backward(void *, int, int, int, int, int, const float *, const int64_t *, const float *, const float *, const float *, const float *, const float *, const float *, float *, float *, float *, float *, float *, float *, void *) { return DRLGX_E_INVALID; }
}

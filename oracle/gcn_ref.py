"""Plain-PyTorch fp32 reference of the GCN policy path — TEST INFRASTRUCTURE ONLY.

Restates PyTorch-Geometric 1.x `GCNConv(improved=True)` (third-party, absent) and
`scripts/Networks.py:12-70` (GCN / PolicyGCN / ValueGCN) with `x @ W` + `index_add_`.  Used by the
tests as the floating-point reference for the HIP GCN kernels and by the CSV pin replay.
"""
import torch


def gcn_norm(edge_index, edge_weight, num_nodes, improved=True):
    """add_remaining_self_loops(fill = 2 if improved) + symmetric normalisation (PyG 1.x GCNConv.norm)."""
    fill = 2.0 if improved else 1.0
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_w = torch.full((num_nodes,), fill, dtype=edge_weight.dtype, device=edge_weight.device)
    # existing self loops keep their weight
    inv = ~mask
    if inv.any():
        loop_w[row[inv]] = edge_weight[inv]
    loop_idx = torch.arange(num_nodes, device=edge_index.device)
    row = torch.cat([row[mask], loop_idx])
    col = torch.cat([col[mask], loop_idx])
    w = torch.cat([edge_weight[mask], loop_w])
    deg = torch.zeros(num_nodes, dtype=w.dtype, device=w.device).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return row, col, dis[row] * w * dis[col]


def gcn_conv(x, edge_index, edge_weight, weight, bias, improved=True):
    n = x.shape[0]
    x = x @ weight
    row, col, norm = gcn_norm(edge_index, edge_weight, n, improved)
    out = torch.zeros_like(x).index_add_(0, col, norm.unsqueeze(1) * x[row])
    return out + bias


def gcn_forward(params, x, edge_index, edge_attr, dropout_mask=None):
    """Networks.GCN.forward (Networks.py:19-28).  params: state_dict with the reference's keys.
    dropout_mask: None (p = 0) or a [N,1000] tensor of {0, 1/(1-p)} applied after the 2nd ReLU."""
    h = torch.relu(gcn_conv(x, edge_index, edge_attr, params["conv1.weight"], params["conv1.bias"]))
    h = torch.relu(gcn_conv(h, edge_index, edge_attr, params["conv2.weight"], params["conv2.bias"]))
    if dropout_mask is not None:
        h = h * dropout_mask
    return h @ params["fully_con1.weight"].t() + params["fully_con1.bias"]


def segment_softmax(src, index, num_segments):
    """torch_geometric.utils.softmax (PyG 1.x): max-shifted exp / (segment sum + 1e-16)."""
    mx = torch.full((num_segments,), -float("inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    s = torch.zeros(num_segments, dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (s[index] + 1e-16)


def policy_gcn_forward(params, x, edge_index, edge_attr, mask, batch, num_graphs, dropout_mask=None):
    """Networks.PolicyGCN.forward (Networks.py:38-50)."""
    q = gcn_forward(params, x, edge_index, edge_attr, dropout_mask).view(-1)
    q = q[mask]
    b = batch[mask]
    return segment_softmax(q, b, num_graphs)


def value_gcn_forward(params, x, edge_index, edge_attr, batch, num_graphs, dropout_mask=None):
    """Networks.ValueGCN.forward (Networks.py:60-70): Linear 1000->100, global_mean_pool, mean(dim=1)."""
    h = gcn_forward(params, x, edge_index, edge_attr, dropout_mask)  # [N,100]
    s = torch.zeros(num_graphs, h.shape[1], dtype=h.dtype, device=h.device).index_add_(0, batch, h)
    cnt = torch.zeros(num_graphs, dtype=h.dtype, device=h.device).index_add_(0, batch, torch.ones_like(batch, dtype=h.dtype))
    return (s / cnt.clamp(min=1).unsqueeze(1)).mean(dim=1)

"""Policy networks with the reference's API (scripts/Networks.py:12-70) on the HIP GCN kernels.

`GCN`, `PolicyGCN` and `ValueGCN` keep the reference's constructor signature, `forward(data, prob_or_mask,
batch=None)` and `state_dict` keys/shapes (`conv1.weight [5,1000]` i.e. [in,out] as PyG 1.x stores it,
`conv1.bias`, `conv2.weight [1000,1000]`, `conv2.bias`, `fully_con1.weight`, `fully_con1.bias`), so the shipped
`MyModel.pt` files load unchanged (scripts/test.py:38-45, scripts/run_training.py:22-36).

The trunk  H1 = relu(Â X W1 + b1), H2 = relu(Â H1 W2 + b2) * dropout_mask, out = H2 Wf^T + bf  runs in
`drlgx_gcn_forward / drlgx_gcn_backward` (csrc/k_gcn.hip: CSR aggregation + fp32-MFMA GEMMs); torch supplies only
device memory, the dropout mask (torch RNG) and the tiny heads (segment softmax / mean pool).
`data` is duck-typed like a PyG `Data`/`Batch`: `.x [N,5] f32`, `.edge_index [2,E] i64`, `.edge_attr [E] f32`.
There is no CPU fallback: tensors must live on a HIP device.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _GCNTrunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_index, edge_attr, W1, b1, W2, b2, Wf, bf, mask, segs=None):
        if not x.is_cuda:
            raise _lib.DrlgxError("drlgx GCN kernels need HIP tensors (no CPU fallback)")
        L = _lib.lib()
        x = x.contiguous().float()
        edge_index = edge_index.contiguous().long()
        edge_attr = edge_attr.contiguous().float()
        W1c, b1c, W2c, b2c, Wfc, bfc = (t.contiguous().float() for t in (W1, b1, W2, b2, Wf, bf))
        N, in_dim = x.shape
        E = edge_index.shape[1]
        hidden = W1c.shape[1]
        out_dim = Wfc.shape[0]
        nbytes = L.drlgx_gcn_workspace_bytes(N, E, hidden, out_dim)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty(N, out_dim, dtype=torch.float32, device=x.device)
        if mask is not None:
            mask = mask.contiguous().float()
        stream = _lib.stream_ptr(x.device)
        _lib.check(_forward_call(L, stream, N, E, in_dim, hidden, out_dim, x, edge_index, edge_attr, (W1c, b1c, W2c, b2c, Wfc, bfc), mask, out,
                                 ws, segs))
        ctx.save_for_backward(x, edge_index, edge_attr, W1c, W2c, Wfc, mask if mask is not None else torch.empty(0, device=x.device), ws)
        ctx.has_mask = mask is not None
        ctx.dims = (N, E, in_dim, hidden, out_dim)
        return out

    @staticmethod
    def backward(ctx, d_out):
        L = _lib.lib()
        x, edge_index, edge_attr, W1, W2, Wf, mask, ws = ctx.saved_tensors
        N, E, in_dim, hidden, out_dim = ctx.dims
        d_out = d_out.contiguous().float()
        dev = x.device
        dW1 = torch.empty(in_dim, hidden, dtype=torch.float32, device=dev)
        db1 = torch.empty(hidden, dtype=torch.float32, device=dev)
        dW2 = torch.empty(hidden, hidden, dtype=torch.float32, device=dev)
        db2 = torch.empty(hidden, dtype=torch.float32, device=dev)
        dWf = torch.empty(out_dim, hidden, dtype=torch.float32, device=dev)
        dbf = torch.empty(out_dim, dtype=torch.float32, device=dev)
        stream = _lib.stream_ptr(dev)
        rc = L.drlgx_gcn_backward(C.c_void_p(stream), N, E, in_dim, hidden, out_dim, _p(x), _p(edge_index), _p(edge_attr), _p(W1),
                                  _p(W2), _p(Wf), _p(mask) if ctx.has_mask else None, _p(d_out), _p(dW1), _p(db1), _p(dW2), _p(db2),
                                  _p(dWf), _p(dbf), _p(ws))
        _lib.check(rc)
        return None, None, None, dW1, db1, dW2, db2, dWf, dbf, None, None


def _forward_call(L, stream, N, E, in_dim, hidden, out_dim, x, edge_index, edge_attr, params, mask, out, ws, segs):
    """drlgx_gcn_forward, or drlgx_gcn_forward_batched when the batch's graph boundaries are known
    (`segs` = (n_graphs, node_off int32 [G+1], edge_off int32 [G+1] on the device, host bound on a graph's edge count))."""
    args = [C.c_void_p(stream), N, E, in_dim, hidden, out_dim, _p(x), _p(edge_index), _p(edge_attr)] + [_p(t) for t in params] + \
           [_p(mask), _p(out), _p(ws)]
    if segs is not None:
        return L.drlgx_gcn_forward_batched(*args, int(segs[0]), _p(segs[1]), _p(segs[2]), int(segs[3]))
    return L.drlgx_gcn_forward(*args)


def graph_segments(data):
    """(n_graphs, node_off, edge_off, max edges of a graph) of a batch that carries its graph boundaries (Engine.graph
    exports, pool collations), else None."""
    no, eo, me = getattr(data, "node_off", None), getattr(data, "edge_off", None), getattr(data, "max_graph_edges", None)
    if no is None or eo is None or me is None or not no.is_cuda or no.dtype != torch.int32 or eo.dtype != torch.int32:
        return None
    return (no.numel() - 1, no, eo, int(me))


def gcn_forward_raw(x, edge_index, edge_attr, params, mask=None, segs=None):
    """The trunk without an autograd graph: `params` = (W1, b1, W2, b2, Wf, bf) fp32 HIP tensors.  Returns (out, saved);
    `saved` is what `gcn_backward_raw` needs (the workspace holds AX / H1 / AH1 / H2 and both CSRs)."""
    if not x.is_cuda:
        raise _lib.DrlgxError("drlgx GCN kernels need HIP tensors (no CPU fallback)")
    L = _lib.lib()
    W1, b1, W2, b2, Wf, bf = (t.detach() for t in params)
    N, in_dim = x.shape
    E = edge_index.shape[1]
    hidden, out_dim = W1.shape[1], Wf.shape[0]
    ws = torch.empty(L.drlgx_gcn_workspace_bytes(N, E, hidden, out_dim), dtype=torch.uint8, device=x.device)
    out = torch.empty(N, out_dim, dtype=torch.float32, device=x.device)
    stream = _lib.stream_ptr(x.device)
    _lib.check(_forward_call(L, stream, N, E, in_dim, hidden, out_dim, x, edge_index, edge_attr, (W1, b1, W2, b2, Wf, bf), mask, out, ws, segs))
    return out, (x, edge_index, edge_attr, W1, W2, Wf, mask, ws, (N, E, in_dim, hidden, out_dim))


def gcn_backward_raw(saved, d_out, grads):
    """Gradients of the six parameter tensors written (not accumulated) into `grads` = (dW1, db1, dW2, db2, dWf, dbf)."""
    L = _lib.lib()
    x, edge_index, edge_attr, W1, W2, Wf, mask, ws, (N, E, in_dim, hidden, out_dim) = saved
    stream = _lib.stream_ptr(x.device)
    _lib.check(L.drlgx_gcn_backward(C.c_void_p(stream), N, E, in_dim, hidden, out_dim, _p(x), _p(edge_index), _p(edge_attr), _p(W1), _p(W2),
                                    _p(Wf), _p(mask), _p(d_out), *(_p(g) for g in grads), _p(ws)))


def gcn_trunk(x, edge_index, edge_attr, W1, b1, W2, b2, Wf, bf, mask=None, segs=None):
    return _GCNTrunk.apply(x, edge_index, edge_attr, W1, b1, W2, b2, Wf, bf, mask, segs)


class GCNConvParams(torch.nn.Module):
    """Parameter holder with PyG-1.x GCNConv's layout and init (weight [in, out] glorot, bias zeros)."""

    def __init__(self, in_channels, out_channels, improved=True):
        super().__init__()
        assert improved, "only GCNConv(improved=True) is used by the reference"
        self.weight = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        stdv = math.sqrt(6.0 / (in_channels + out_channels))
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)


_ONES = {}  # device -> a [rows, hidden] tensor of ones (the input of the mask draw below)
_ONES_MAX_ELEMS = 1 << 26  # 256 MB of float32: larger masks take the two-kernel path


def _dropout_mask(n, hidden, p, device):
    """F.dropout(x, p) is the FUNCTIONAL form in the reference => always active.  The mask (0 or
    1 / (1 - p)) is drawn by F.dropout itself on a tensor of ones of the activations' shape: one kernel, and the same
    Philox draw as the reference's F.dropout(x) on its [n, hidden] activations."""
    if p <= 0.0:
        return None
    if p >= 1.0:
        return torch.zeros(n, hidden, device=device)
    if n * hidden <= _ONES_MAX_ELEMS:
        key = (str(device), hidden)
        ones = _ONES.get(key)
        if ones is None or ones.shape[0] < n:
            rows = max(n, 2 * (ones.shape[0] if ones is not None else 0))
            rows = min(rows, _ONES_MAX_ELEMS // hidden)
            ones = _ONES[key] = torch.ones(rows, hidden, device=device)
        return torch.nn.functional.dropout(ones[:n], p, True)
    return torch.empty(n, hidden, device=device).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))


class GCN(torch.nn.Module):
    """scripts/Networks.py:12-28 (DQN head: one Q value per node)."""

    def __init__(self):
        super().__init__()
        self.conv1 = GCNConvParams(5, 1000, improved=True)
        self.conv2 = GCNConvParams(1000, 1000, improved=True)
        self.fully_con1 = torch.nn.Linear(1000, 1)

    def trunk_parameters(self):
        """(W1, b1, W2, b2, Wf, bf) in the order of drlgx_gcn_forward / _backward."""
        return (self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, self.fully_con1.weight, self.fully_con1.bias)

    def forward(self, data, prob, batch=None):
        x, edge_index, edge_weight = data.x, data.edge_index, data.edge_attr
        mask = _dropout_mask(x.shape[0], 1000, float(prob), x.device)
        return gcn_trunk(x, edge_index, edge_weight, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                         self.fully_con1.weight, self.fully_con1.bias, mask, graph_segments(data))


class _SegmentSoftmax(torch.autograd.Function):
    """drlgx_segment_softmax / _backward: masked_select + per-graph softmax of the read-out (PolicyGCN head)."""

    @staticmethod
    def forward(ctx, q, mask, node_off, n_selected):
        L = _lib.lib()
        q = q.contiguous().float()
        mask = mask.contiguous()
        p = torch.empty(n_selected, dtype=torch.float32, device=q.device)
        stream = _lib.stream_ptr(q.device)
        _lib.check(L.drlgx_segment_softmax(C.c_void_p(stream), node_off.numel() - 1, _p(node_off), _p(q), _p(mask), _p(p)))
        ctx.save_for_backward(p, mask, node_off)
        ctx.n = q.numel()
        return p

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        p, mask, node_off = ctx.saved_tensors
        dq = torch.empty(ctx.n, dtype=torch.float32, device=p.device)
        stream = _lib.stream_ptr(p.device)
        _lib.check(L.drlgx_segment_softmax_backward(C.c_void_p(stream), node_off.numel() - 1, _p(node_off), _p(p), _p(dp.contiguous().float()),
                                                    _p(mask), _p(dq)))
        return dq, None, None, None


class _MeanPool(torch.autograd.Function):
    """drlgx_mean_pool / _backward: global_mean_pool(h, batch).mean(dim=1) (ValueGCN head)."""

    @staticmethod
    def forward(ctx, h, node_off):
        L = _lib.lib()
        h = h.contiguous().float()
        G = node_off.numel() - 1
        v = torch.empty(G, dtype=torch.float32, device=h.device)
        stream = _lib.stream_ptr(h.device)
        _lib.check(L.drlgx_mean_pool(C.c_void_p(stream), G, _p(node_off), _p(h), h.shape[1], _p(v)))
        ctx.save_for_backward(node_off)
        ctx.shape = tuple(h.shape)
        return v

    @staticmethod
    def backward(ctx, dv):
        L = _lib.lib()
        (node_off,) = ctx.saved_tensors
        dh = torch.empty(ctx.shape, dtype=torch.float32, device=dv.device)
        stream = _lib.stream_ptr(dv.device)
        _lib.check(L.drlgx_mean_pool_backward(C.c_void_p(stream), node_off.numel() - 1, _p(node_off), _p(dv.contiguous().float()), ctx.shape[1],
                                              _p(dh)))
        return dh, None


def segment_softmax(src, index, num_segments):
    """torch_geometric.utils.softmax (PyG 1.x): exp(src - segment max) / (segment sum + 1e-16)."""
    mx = torch.full((num_segments,), -float("inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    s = torch.zeros(num_segments, dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (s[index] + 1e-16)


class PolicyGCN(torch.nn.Module):
    """scripts/Networks.py:31-50 (A2C actor: softmax over the masked (frontier) nodes of every graph)."""

    def __init__(self):
        super().__init__()
        self.conv1 = GCNConvParams(5, 1000, improved=True)
        self.conv2 = GCNConvParams(1000, 1000, improved=True)
        self.fully_con1 = torch.nn.Linear(1000, 1)

    def forward(self, data, mask, batch=None):
        x, edge_index, edge_weight = data.x, data.edge_index, data.edge_attr
        dmask = _dropout_mask(x.shape[0], 1000, 0.5, x.device)  # F.dropout(x): p = 0.5 even at inference
        q = gcn_trunk(x, edge_index, edge_weight, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                      self.fully_con1.weight, self.fully_con1.bias, dmask, graph_segments(data))
        segs = graph_segments(data)
        if segs is not None and mask.dtype == torch.bool:  # wavefront segment reductions over the batch's graph boundaries
            n_masked = getattr(data, "n_masked", None)  # (a caller that knows the count on the host spares the synchronisation)
            return _SegmentSoftmax.apply(q.view(-1), mask, segs[1], int(mask.sum()) if n_masked is None else int(n_masked))
        q = torch.masked_select(q.view(-1), mask)
        b = torch.masked_select(batch, mask)
        return segment_softmax(q, b, int(batch.max().item()) + 1 if batch.numel() else 0)


class ValueGCN(torch.nn.Module):
    """scripts/Networks.py:53-70 (A2C critic: Linear 1000->100, global mean pool, mean over the 100)."""

    def __init__(self):
        super().__init__()
        self.conv1 = GCNConvParams(5, 1000, improved=True)
        self.conv2 = GCNConvParams(1000, 1000, improved=True)
        self.fully_con1 = torch.nn.Linear(1000, 100)

    def forward(self, data, mask, batch=None):
        x, edge_index, edge_weight = data.x, data.edge_index, data.edge_attr
        dmask = _dropout_mask(x.shape[0], 1000, 0.5, x.device)
        h = gcn_trunk(x, edge_index, edge_weight, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                      self.fully_con1.weight, self.fully_con1.bias, dmask, graph_segments(data))
        segs = graph_segments(data)
        if segs is not None:
            return _MeanPool.apply(h, segs[1])
        g = int(batch.max().item()) + 1
        s = torch.zeros(g, h.shape[1], dtype=h.dtype, device=h.device).index_add_(0, batch, h)
        cnt = torch.zeros(g, dtype=h.dtype, device=h.device).index_add_(0, batch, torch.ones_like(batch, dtype=h.dtype))
        return (s / cnt.clamp(min=1).unsqueeze(1)).mean(dim=1)  # global_mean_pool(x, batch).mean(dim=1)


class GraphData(object):
    """Minimal stand-in for torch_geometric.data.Data / Batch (x, edge_index, edge_attr, batch, .to())."""

    def __init__(self, x, edge_index, edge_attr, batch=None, node_off=None, edge_off=None, max_graph_edges=None):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch
        # optional graph boundaries of a batch (device int32 [n_graphs + 1]) and a host bound on the edges of one graph:
        # the GCN builds its CSRs per graph from them
        self.node_off, self.edge_off, self.max_graph_edges = node_off, edge_off, max_graph_edges

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def to(self, device):
        self.x = self.x.to(device)
        self.edge_index = self.edge_index.to(device)
        self.edge_attr = self.edge_attr.to(device)
        if self.batch is not None:
            self.batch = self.batch.to(device)
        if self.node_off is not None:
            self.node_off, self.edge_off = self.node_off.to(device), self.edge_off.to(device)
        return self

    @staticmethod
    def collate(items):
        """torch_geometric DataLoader/Batch semantics: concatenate, offset edge_index by cumulative node counts."""
        if items and all(isinstance(d, PoolRef) for d in items) and all(d.pool is items[0].pool for d in items):
            return items[0].pool.collate(items)  # one gather per tensor instead of a few torch ops per graph
        xs, eis, eas, counts, ecounts = [], [], [], [], []
        off = 0
        for d in items:
            x = d.x
            n = x.shape[0]
            xs.append(x)
            # a GraphSlice keeps the batch-global node ids of the export it was cut from: one shift does both offsets
            eis.append(d.edge_index_global + (off - d.node0) if isinstance(d, GraphSlice) else d.edge_index + off)
            eas.append(d.edge_attr)
            counts.append(n)
            ecounts.append(int(eas[-1].shape[0]))
            off += n
        dev = xs[0].device
        batch = torch.repeat_interleave(torch.arange(len(items), device=dev), torch.tensor(counts, device=dev), output_size=off)
        offs = torch.from_numpy(np.stack([np.concatenate([[0], np.cumsum(counts)]), np.concatenate([[0], np.cumsum(ecounts)])]).astype(np.int32)).to(dev)
        return GraphData(torch.cat(xs), torch.cat(eis, dim=1), torch.cat(eas), batch, offs[0], offs[1], max(ecounts))


class GraphSlice(object):
    """One env's graph as lazy VIEWS into the engine's batched export (no copies, no kernels and no tensor objects until
    it is used): what a replay transition holds.  Same duck type as GraphData; `edge_index` (local node ids) is
    materialised on access."""

    __slots__ = ("g", "node0", "node1", "edge0", "edge1", "batch")

    def __init__(self, g, n0, n1, e0, e1):
        self.g, self.node0, self.node1, self.edge0, self.edge1, self.batch = g, int(n0), int(n1), int(e0), int(e1), None

    @property
    def num_nodes(self):
        return self.node1 - self.node0

    @property
    def x(self):
        return self.g["x"][self.node0:self.node1]

    @property
    def edge_attr(self):
        return self.g["edge_attr"][self.edge0:self.edge1]

    @property
    def edge_index_global(self):
        return self.g["edge_index"][:, self.edge0:self.edge1]

    @property
    def edge_index(self):
        return self.edge_index_global - self.node0

    def to(self, device):
        x = self.x
        return self if x.device == torch.device(device) else GraphData(x, self.edge_index, self.edge_attr).to(device)


class ReplayPool(object):
    """Device-resident storage of the graphs a replay buffer refers to: every batched export (`Engine.graph`) is copied
    once into one slot of three pooled tensors, a transition holds `PoolRef`s (slot, env) into it, and a mini-batch is
    collated with ONE gather per tensor from host-built index arrays (the per-graph torch ops of a Python-side collate
    cost more than the train step they feed).  Slots are recycled when no transition refers to them any more."""

    def __init__(self, device, n_slots, cap_nodes, cap_edges, in_dim=5, cache_csr=False):
        self.device, self.n_slots, self.cap_nodes, self.cap_edges = device, n_slots, cap_nodes, cap_edges
        # cache_csr: what the GCN derives from a stored graph (degrees, self weights, both CSRs with the normalised weights,
        # AX) is computed once per export, at `put`, and a mini-batch is collated from THAT (include/drlgx.h: drlgx_csr_cache)
        self.csr, self.csr_ok = None, np.zeros(n_slots, dtype=bool)
        if cache_csr and torch.device(device).type == "cuda":
            nodes, edges = n_slots * cap_nodes, n_slots * cap_edges
            nw, ew = sum(_lib.CsrCache.NODE_WORDS.values()), sum(_lib.CsrCache.EDGE_WORDS.values())
            self._csr_buf = torch.empty(nw * nodes + ew * edges + 64, dtype=torch.int32, device=device)
            self.csr = _lib.CsrCache()
            off = (-self._csr_buf.data_ptr() // 4) % 4  # (AX rows are moved as 16-byte pairs)
            self._csr_off = {}
            for name, _ in _lib.CsrCache._fields_:
                words = _lib.CsrCache.NODE_WORDS.get(name, 0) * nodes + _lib.CsrCache.EDGE_WORDS.get(name, 0) * edges
                self._csr_off[name] = off
                setattr(self.csr, name, self._csr_buf.data_ptr() + 4 * off)
                off += (words + 3) & ~3
        self.X = torch.empty(n_slots * cap_nodes, in_dim, dtype=torch.float32, device=device)
        self.EI = torch.empty(2, n_slots * cap_edges, dtype=torch.int64, device=device)
        self.EA = torch.empty(n_slots * cap_edges, dtype=torch.float32, device=device)
        self.node_off = [None] * n_slots  # host int64 [n_graphs + 1] per slot
        self.edge_off = [None] * n_slots
        self.desc = [None] * n_slots      # host int64 [n_graphs, 5] per slot
        self.ref = [0] * n_slots
        self._next = 0
        # one cached float per pooled node (the DQN trainer keeps the target network's read-out here), with the version of
        # the weights it was computed with per slot
        self.Q = torch.empty(n_slots * cap_nodes, dtype=torch.float32, device=device)
        self.q_version = [None] * n_slots

    def put(self, g):
        """Copy one export (dict of `Engine.graph`, with host offsets `node_off_h` / `edge_off_h`) into a free slot."""
        for k in range(self.n_slots):
            slot = (self._next + k) % self.n_slots
            if self.ref[slot] == 0:
                break
        else:
            raise RuntimeError("replay pool exhausted: every slot is still referenced")
        self._next = (slot + 1) % self.n_slots
        N, E = int(g["node_off_h"][-1]), int(g["edge_off_h"][-1])
        if N > self.cap_nodes or E > self.cap_edges:
            raise RuntimeError("export larger than a replay-pool slot")
        self.X[slot * self.cap_nodes:slot * self.cap_nodes + N] = g["x"]
        self.EI[:, slot * self.cap_edges:slot * self.cap_edges + E] = g["edge_index"]
        self.EA[slot * self.cap_edges:slot * self.cap_edges + E] = g["edge_attr"]
        no = self.node_off[slot] = np.asarray(g["node_off_h"], dtype=np.int64)
        eo = self.edge_off[slot] = np.asarray(g["edge_off_h"], dtype=np.int64)
        # drlgx_replay_collate's descriptor of every graph of the export (PoolRef.d5 is a row of this)
        self.desc[slot] = np.stack([slot * self.cap_nodes + no[:-1], np.diff(no), slot * self.cap_edges + eo[:-1], np.diff(eo), no[:-1]], axis=1)
        self.q_version[slot] = None
        if self.csr is not None:
            self._cache_csr(slot, g, no, eo)
        return slot

    def _cache_csr(self, slot, g, no, eo):
        """drlgx_replay_cache_csr over the export just stored in `slot` (one launch, one workgroup per graph)."""
        n_graphs = len(no) - 1
        nod, eod = g.get("node_off"), g.get("edge_off")
        if nod is None or eod is None or nod.dtype != torch.int32 or not nod.is_cuda:
            offs = torch.from_numpy(np.stack([no, eo]).astype(np.int32)).to(self.device)
            nod, eod = offs[0], offs[1]
        at = _lib.CsrCache()
        n0, e0 = slot * self.cap_nodes, slot * self.cap_edges
        for name, _ in _lib.CsrCache._fields_:
            per = _lib.CsrCache.NODE_WORDS.get(name)
            setattr(at, name, getattr(self.csr, name) + 4 * (per * n0 if per else e0))
        rc = _lib.lib().drlgx_replay_cache_csr(C.c_void_p(_lib.stream_ptr(self.device)), n_graphs, _p(nod), _p(eod),
                                               int(np.diff(eo).max()) if n_graphs else 0, C.c_void_p(self.X.data_ptr() + 4 * n0 * self.X.shape[1]),
                                               self.X.shape[1], C.c_void_p(self.EI.data_ptr() + 8 * e0), self.EI.shape[1],
                                               C.c_void_p(self.EA.data_ptr() + 4 * e0), C.byref(at))
        self.csr_ok[slot] = rc == 0
        if rc not in (0, -3):  # (DRLGX_E_CAPACITY: a graph beyond the per-graph sort - that export is collated the generic way)
            _lib.check(rc)

    def export(self, slot):
        """The batched export stored in `slot` as one GraphData (views into the pool, with its graph boundaries)."""
        no, eo = self.node_off[slot], self.edge_off[slot]
        N, E = int(no[-1]), int(eo[-1])
        n0, e0 = slot * self.cap_nodes, slot * self.cap_edges
        offs = torch.from_numpy(np.stack([no, eo]).astype(np.int32)).to(self.device)
        return GraphData(self.X[n0:n0 + N], self.EI[:, e0:e0 + E].contiguous(), self.EA[e0:e0 + E], None, offs[0], offs[1],
                         int(np.diff(eo).max()) if len(eo) > 1 else 0)

    @staticmethod
    def descriptors(refs):
        """Host side of a collation: int64 [5, k] = node start in the pool, node count, edge start in the pool, edge count,
        first node id inside the export (drlgx_replay_collate's `desc`), plus the node / edge totals."""
        d = np.concatenate([r.d5 for r in refs]).reshape(len(refs), 5).T.copy()
        return d, int(d[1].sum()), int(d[3].sum())

    def collate_from(self, desc_dev, k, n_nodes, n_edges, max_graph_edges=None, with_q=False, q_desc=None, q_nodes=0):
        """The PyG batch of `k` pooled graphs from their device descriptors: one kernel, no host synchronisation.
        with_q: also gather the pool's per-node cache (`.q` of the result).  q_desc / q_nodes: gather the cache of a SECOND
        list of `k` graphs (their descriptors, their node total) in the same launch instead - `.q` is then theirs."""
        dev = self.device
        x = torch.empty(n_nodes, self.X.shape[1], dtype=torch.float32, device=dev)
        ei = torch.empty(2, n_edges, dtype=torch.int64, device=dev)
        ea = torch.empty(n_edges, dtype=torch.float32, device=dev)
        bt = torch.empty(n_nodes, dtype=torch.int64, device=dev)
        offs = torch.empty(2, k + 1, dtype=torch.int32, device=dev)
        stream = _lib.stream_ptr(dev)
        if q_desc is not None:
            q = torch.empty(q_nodes, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().drlgx_replay_collate_pair(C.c_void_p(stream), k, _p(desc_dev), _p(self.X), self.X.shape[1], _p(self.EI),
                                                            self.EI.shape[1], _p(self.EA), _p(x), _p(ei), n_edges, _p(ea), _p(bt), _p(offs[0]),
                                                            _p(offs[1]), _p(q_desc), _p(self.Q), _p(q)))
        else:
            q = torch.empty(n_nodes, dtype=torch.float32, device=dev) if with_q else None
            _lib.check(_lib.lib().drlgx_replay_collate(C.c_void_p(stream), k, _p(desc_dev), _p(self.X), self.X.shape[1], _p(self.EI),
                                                       self.EI.shape[1], _p(self.EA), _p(x), _p(ei), n_edges, _p(ea), _p(bt), _p(offs[0]),
                                                       _p(offs[1]), _p(self.Q) if with_q else None, _p(q)))
        d = GraphData(x, ei, ea, bt, offs[0], offs[1], max_graph_edges)
        d.q = q
        return d

    def gather_q(self, desc_dev, k, n_nodes):
        """The per-node cache of `k` pooled graphs, concatenated in collation order."""
        q = torch.empty(n_nodes, dtype=torch.float32, device=self.device)
        stream = _lib.stream_ptr(self.device)
        _lib.check(_lib.lib().drlgx_replay_collate(C.c_void_p(stream), k, _p(desc_dev), _p(self.X), self.X.shape[1], _p(self.EI),
                                                   self.EI.shape[1], _p(self.EA), None, None, 0, None, None, None, None, _p(self.Q), _p(q)))
        return q

    def collate(self, refs):
        d, n_nodes, n_edges = self.descriptors(refs)
        return self.collate_from(torch.from_numpy(d).to(self.device), len(refs), n_nodes, n_edges, int(d[3].max()))


class PoolRef(object):
    """One graph of a `ReplayPool` (what a replay transition holds); same duck type as GraphData.  The graph's ranges in
    the pooled tensors are resolved once, at construction."""

    __slots__ = ("pool", "slot", "env", "batch", "n0", "nn", "e0", "ne", "loc", "d5")

    def __init__(self, pool, slot, env):
        self.pool, self.slot, self.env, self.batch = pool, int(slot), int(env), None
        self.d5 = pool.desc[self.slot][self.env]  # (n0, nn, e0, ne, loc) as the collation kernel reads them
        self.n0, self.nn, self.e0, self.ne, self.loc = self.d5.tolist()

    @classmethod
    def many(cls, pool, slot, n):
        """The references of graphs 0 .. n-1 of one pooled export: the descriptor block is converted once (a constructor call per
        graph cost ~10 us - 2.5 ms per 256-env vector step of the A2C loop, 5 ms of the DQN loop)."""
        slot = int(slot)
        D = pool.desc[slot]
        rows = D[:n].tolist()
        out = []
        for i in range(n):
            r = cls.__new__(cls)
            r.pool, r.slot, r.env, r.batch, r.d5 = pool, slot, i, None, D[i]
            r.n0, r.nn, r.e0, r.ne, r.loc = rows[i]
            out.append(r)
        return out

    @property
    def num_nodes(self):
        return self.nn

    @property
    def x(self):
        return self.pool.X[self.n0:self.n0 + self.nn]

    @property
    def edge_attr(self):
        return self.pool.EA[self.e0:self.e0 + self.ne]

    @property
    def edge_index(self):
        return self.pool.EI[:, self.e0:self.e0 + self.ne] - self.loc

    def to(self, device):
        return GraphData(self.x, self.edge_index, self.edge_attr).to(device)

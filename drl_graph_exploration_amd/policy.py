"""`DeepQ` — the reference's DQN trainer (scripts/policy.py:16-262) over a `VecExplorationEnv`.

Same constructor `DeepQ(case_path, model_name)`, hyper-parameters, `running / data_process / cost / train / test` and
target computation as the reference; the differences are the ones the batched hot path implies:

* `running` steps `n_envs` environments in lock-step (one decision of every env per iteration; `step_t` advances by
  `n_envs`), graphs never leave the device: `data_process` slices the engine's batched export instead of scanning a
  dense adjacency matrix in Python (the edge order it would have produced is what the engine emits).
* replay transitions hold device `GraphData`; mini-batches are collated on the device.
* multi-GPU: one process per GPU, each with its own env shard and replay buffer; `allreduce_gradients` averages the
  flattened policy gradient over ranks (RCCL over xGMI, one collective per train step) before the clamp + Adam step.
"""
import csv
import gc
import math
import os
import random
from collections import deque

import numpy as np
import torch
import torch.distributed as dist

import ctypes as C

from . import _lib
from . import networks as NW
from .networks import GCN, GraphData, GraphSlice, PoolRef, ReplayPool
from .optim import FusedAdam, GradientBucket
from .vecenv import VecExplorationEnv


def _graph_to_host(d):
    """A replay graph (GraphData or a GraphSlice view into a batched export) as compact host tensors: the on-disk form of
    the reference's pickled PyG `Data` objects (scripts/train.py:33-35: the trainer, with its replay buffer, is pickled to
    saved_training.pkl and re-loaded by every run_training.py epoch)."""
    return GraphData(d.x.detach().cpu().clone(), d.edge_index.detach().cpu().clone(), d.edge_attr.detach().cpu().clone())


def _is_rank0():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_state_dict(module, path):
    """torch.save of a state_dict as the reference writes its artefacts (policy.py:194-199), made safe for the
    data-parallel trainer: rank 0 only (every rank holds the same parameters), through a temporary file + os.replace, so
    that a reader such as test.py never sees a truncated or interleaved archive."""
    if not _is_rank0():
        return
    tmp = "%s.tmp.%d" % (path, os.getpid())
    torch.save(module.state_dict(), tmp)
    os.replace(tmp, path)


def allreduce_gradients(model, group=None, optimizer=None):
    """Average the gradients of `model` over all ranks with ONE flat all-reduce (3 MB..4 MB for the GCN: far below
    the xGMI per-link bandwidth-delay product, so a single bucket is optimal).  No-op without a process group.
    With a `FusedAdam` the gradients already are views of one flat tensor: it is reduced in place (no concatenation, no
    copies back).  The trainers' hot path does not call this at all: it issues the exchange asynchronously right after
    the backward pass and folds the 1 / world into the Adam kernel (`DeepQ._train_begin / _train_end`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    ws = dist.get_world_size(group)
    if ws == 1:
        return
    if isinstance(optimizer, FusedAdam):
        optimizer.grads()
        optimizer.bucket.start(group)
        optimizer.bucket.finish(group, apply=True)
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(ws)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_parameters(model, src=0, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


# the trainers' exports plan the line to every frontier in the same call (VecExplorationEnv.graph_matrix(plan=True)); DRLGX_EXPORT_PLAN=0:
# a separate actions_all_goals() with its own synchronisation (A/B runs)
_EXPORT_PLAN = os.environ.get("DRLGX_EXPORT_PLAN", "1") != "0"


class _QuietGc(object):
    """Python's cyclic collector held off while a trainer's loop runs.  A vector step allocates a few thousand short-lived
    containers (one PoolRef / buffer tuple / log row per env); the automatic collections they trigger walk the whole heap of a
    torch process - 1-2 ms per vector step of a 256-env A2C loop, all of it on the critical path between two launches.  Reference
    counting frees the loop's objects anyway; `tick()` (once per update) collects the young generations, and everything every
    64th time, so that cycles cannot pile up over a long training."""

    def __init__(self, full_every=64):
        self.full_every, self.n, self.was = full_every, 0, False

    def __enter__(self):
        self.was = gc.isenabled()
        gc.disable()
        return self

    def tick(self):
        self.n += 1
        gc.collect(1 if self.n % self.full_every else 2)

    def __exit__(self, *exc):
        if self.was:
            gc.enable()
        return False


class ReplayList(list):
    """The replay buffer: a list with the two deque methods the reference's trainer uses (O(1) random access for
    `random.sample`; `popleft` moves 10^4 pointers, microseconds).

    Beside the tuples it keeps a NUMERIC side table of the transitions whose graphs live in one `ReplayPool` (int64
    [rows][14] = the collation descriptor of s_j (5) and of s_j1 (5), action node, next-state frontier count, terminal,
    slot of s_j1; float64 reward), so that sampling n mini-batches is one fancy-indexing gather instead of n x BATCH
    attribute reads (`DeepQ._prepare_updates`).  `append` / `extend` / `popleft` keep it aligned; any other mutation, or a
    transition that is not pool-backed, switches it off (`table()` returns None and the trainer takes the generic path)."""
    _COLS = 14

    def __init__(self, items=()):
        super().__init__()
        self._reset_side()
        for t in items:
            self.append(t)

    def _reset_side(self):
        self._rows, self._rew, self._head, self._n, self._pool, self._ok = None, None, 0, 0, None, True

    def _side_push(self, t):
        a, b = t[0], t[3]
        if not (isinstance(a, PoolRef) and isinstance(b, PoolRef)) or a.pool is not b.pool or (self._pool is not None and a.pool is not self._pool):
            self._ok = False
            self._rows = self._rew = None
            return
        if self._rows is None:
            self._pool = a.pool
            self._rows = np.empty((4096, self._COLS), dtype=np.int64)
            self._rew = np.empty(4096, dtype=np.float64)
            self._head = self._n = 0
        end = self._head + self._n
        if end == self._rows.shape[0]:
            if self._head >= self._rows.shape[0] // 2:  # compact: the live rows move to the front
                self._rows[:self._n] = self._rows[self._head:end]
                self._rew[:self._n] = self._rew[self._head:end]
            else:                                        # grow
                rows, rew = np.empty((2 * self._rows.shape[0], self._COLS), dtype=np.int64), np.empty(2 * self._rows.shape[0])
                rows[:self._n], rew[:self._n] = self._rows[self._head:end], self._rew[self._head:end]
                self._rows, self._rew = rows, rew
            self._head, end = 0, self._n
        r = self._rows[end]
        r[0:5], r[5:10] = a.d5, b.d5
        r[10], r[11], r[12], r[13] = t[1], t[5], 1 if t[4] else 0, b.slot
        self._rew[end] = t[2]
        self._n += 1

    def append(self, t):
        if "_ok" not in self.__dict__:  # (unpickling appends the items before it restores the attributes)
            self._reset_side()
        list.append(self, t)
        if self._ok:
            self._side_push(t)

    def extend(self, items):
        for t in items:
            self.append(t)

    def popleft(self):
        t = list.pop(self, 0)
        if self._ok and self._n:
            self._head += 1
            self._n -= 1
        return t

    # Every other mutation of the list switches the side table off for good (the trainer then takes the generic path, which reads the
    # transitions themselves): a same-length mutation - buffer[i] = t, sort, reverse, random.shuffle - would otherwise leave rows
    # that no longer describe the items, and `table()` only compares the first and the last one.
    def _side_off(self):
        self._ok = False
        self._rows = self._rew = None

    def __setitem__(self, i, v):
        self._side_off()
        list.__setitem__(self, i, v)

    def __delitem__(self, i):
        self._side_off()
        list.__delitem__(self, i)

    def __iadd__(self, items):
        self.extend(items)
        return self

    def __imul__(self, k):
        self._side_off()
        return list.__imul__(self, k)

    def insert(self, i, v):
        self._side_off()
        list.insert(self, i, v)

    def pop(self, i=-1):
        if i == 0:
            return self.popleft()
        self._side_off()
        return list.pop(self, i)

    def remove(self, v):
        self._side_off()
        list.remove(self, v)

    def sort(self, *a, **k):
        self._side_off()
        list.sort(self, *a, **k)

    def reverse(self):
        self._side_off()
        list.reverse(self)

    def clear(self):
        list.clear(self)
        self._reset_side()

    def table(self):
        """(pool, rows int64 [len][14], rewards float64 [len]) aligned with the list, or None."""
        if not self.__dict__.get("_ok") or self._rows is None or self._n != len(self) or self._n == 0:
            return None
        a, z = self[0], self[-1]
        h = self._head
        if self._rows[h, 0] != a[0].n0 or self._rows[h + self._n - 1, 5] != z[3].n0:  # (someone reordered the list behind our back)
            self._ok = False
            return None
        return self._pool, self._rows[h:h + self._n], self._rew[h:h + self._n]

    def __reduce_ex__(self, protocol):  # pickled as its items (the side table is rebuilt, or stays off for host graphs)
        return (ReplayList, (list(self),))


class _Prepared(dict):
    """What `_prepare_updates` hands to one update: scalars + raw device addresses; the tensor views (`desc_j`, `desc_j1`,
    `meta`, `r`) of the generic path are made on first use."""

    def __missing__(self, key):
        I, R, u = self["_I"], self["_R"], self["_u"]
        self["desc_j"], self["desc_j1"], self["meta"], self["r"] = I[u, 0:5], I[u, 5:10], I[u, 10:14], R[u]
        return dict.__getitem__(self, key)


class DeepQ(object):
    def __init__(self, case_path, model_name, data_root="../data"):
        self.case_path = case_path
        self.weights_path = os.path.join(data_root, "torch_weights", self.case_path)
        self.reward_data_path = os.path.join(data_root, "reward_data", self.case_path)
        self.object_path = os.path.join(data_root, "training_object_data", self.case_path)
        for p in (self.weights_path, self.reward_data_path, self.object_path):
            os.makedirs(p, exist_ok=True)
        with open(os.path.join(self.reward_data_path, "reward_data.csv"), "w", newline="") as f:
            csv.writer(f).writerow(["Step", "Reward"])

        # RL parameters (policy.py:33-52)
        self.BATCH = 64
        self.REPLAY_MEMORY = 1e4
        self.GAMMA = 0.99
        self.OBSERVE = 5e3
        self.EXPLORE = 1e6
        self.epoch = 1e4
        self.TARGET_UPDATE = 15000 if model_name == "GCN" else 9000
        self.FINAL_EPSILON = 0
        self.INITIAL_EPSILON = 0.9
        self.max_grad_norm = 0.5
        self.map_size = 40
        self.buffer = ReplayList()
        self.step_t = 0
        self.epsilon = self.INITIAL_EPSILON
        self.temp_loss = 0
        self.total_reward = np.empty([0, 0])
        self.target_window = "reference"  # or "aligned": see td_targets
        # minibatch updates per vector step; None = one per environment step like the reference (n_envs per vector step)
        self.updates_per_vector_step = None
        # pool-backed updates as two host calls over one arena (drlgx_dqn_prepare / _forward_backward); False: the same launches one
        # by one from Python (the A/B of scripts/ab_dqn_update_paths.py)
        self.fused_update = True

    @property
    def temp_loss(self):
        """Loss of the last `train` call (scripts/policy.py:249 `self.temp_loss = loss.item()`), synchronised on access."""
        t = self.__dict__.get("_loss_t")
        return float(t) if t is not None else self.__dict__.get("_loss_v", 0)

    @temp_loss.setter
    def temp_loss(self, v):
        self.__dict__["_loss_v"], self.__dict__["_loss_t"] = v, None

    # ------------------------------------------------------------------ pickling (saved_training.pkl hand-off)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_loss_v"], st["_loss_t"] = self.temp_loss, None
        st["buffer"] = ReplayList((_graph_to_host(t[0]), t[1], t[2], _graph_to_host(t[3]), t[4], t[5]) for t in self.buffer)
        st.pop("_pool", None)  # device storage: the pickled transitions carry their graphs themselves
        st.pop("_arena", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    # ------------------------------------------------------------------ data
    @staticmethod
    def data_process(g, i):
        """Graph of env `i` out of the engine's batched export `g` (Engine.graph) as one `GraphData` with local node
        ids — what policy.py:211-232 builds from the dense (A, X) pair."""
        n0, n1 = int(g["node_off_h"][i]), int(g["node_off_h"][i + 1])
        e0, e1 = int(g["edge_off_h"][i]), int(g["edge_off_h"][i + 1])
        return GraphSlice(g, n0, n1, e0, e1)  # views: the replay buffer keeps the export alive, nothing is copied

    def _repool(self, pool, device):
        """Replace the host GraphData of a re-loaded replay buffer by PoolRefs into `pool` (see running)."""
        todo = [(k, j) for k, t in enumerate(self.buffer) for j in (0, 3) if not isinstance(t[j], PoolRef)]
        if not todo:
            return
        buf = list(self.buffer)
        new = {}
        pos = 0
        while pos < len(todo) and any(r == 0 for r in pool.ref):
            # as many graphs as one slot holds
            n_nodes = n_edges = 0
            end = pos
            while end < len(todo):
                d = buf[todo[end][0]][todo[end][1]]
                nn, ne = int(d.x.shape[0]), int(d.edge_index.shape[1])
                if n_nodes + nn > pool.cap_nodes or n_edges + ne > pool.cap_edges:
                    break
                n_nodes += nn
                n_edges += ne
                end += 1
            if end == pos:
                break  # (a graph larger than a slot: cannot happen for graphs exported with the same capacities)
            ds = [buf[k][j] for k, j in todo[pos:end]]
            node_off = np.concatenate([[0], np.cumsum([int(d.x.shape[0]) for d in ds])]).astype(np.int64)
            edge_off = np.concatenate([[0], np.cumsum([int(d.edge_index.shape[1]) for d in ds])]).astype(np.int64)
            g = {"x": torch.cat([d.x.cpu() for d in ds]).to(device),
                 "edge_index": torch.cat([d.edge_index.cpu() + int(o) for d, o in zip(ds, node_off[:-1])], dim=1).to(device),
                 "edge_attr": torch.cat([d.edge_attr.cpu() for d in ds]).to(device),
                 "node_off_h": node_off, "edge_off_h": edge_off}
            slot = pool.put(g)
            for i, kj in enumerate(todo[pos:end]):
                new[kj] = PoolRef(pool, slot, i)
                pool.ref[slot] += 1
            pos = end
        for k, t in enumerate(buf):
            if (k, 0) in new or (k, 3) in new:
                buf[k] = (new.get((k, 0), t[0]), t[1], t[2], new.get((k, 3), t[3])) + tuple(t[4:])
        for k, j in todo[pos:]:  # no free slot left for these
            buf[k][j].to(device)
        self.buffer = ReplayList(buf)

    @staticmethod
    def _host_offsets(g):
        if "node_off_h" not in g:  # (Engine.graph brings them along; a hand-made export may not)
            g["node_off_h"] = g["node_off"].cpu().numpy()
            g["edge_off_h"] = g["edge_off"].cpu().numpy()
        return g

    def cost(self, pred, target, action):
        pred_flat = pred.view(-1)
        target_flat = target.view(-1)
        readout_action = torch.mul(pred_flat, action)
        return torch.pow(readout_action - target_flat, 2).sum() / self.BATCH

    def train(self, data, action, y, device, model, optimizer):
        model.train()
        data = data.to(device)
        if type(model) is GCN and isinstance(optimizer, FusedAdam) and data.x.is_cuda:
            self._train_end(self._train_begin(data, action, y, device, model, optimizer))
            return
        optimizer.zero_grad()
        out = model(data, 0.5, batch=data.batch)
        # the reference builds y / action as numpy float64 and `torch.tensor(y)` keeps that dtype: the product with
        # the float32 read-out promotes, the loss is float64, the parameter gradients float32 (policy.py:241-248)
        y = torch.as_tensor(y, dtype=torch.float64, device=device)
        action = torch.as_tensor(action, dtype=torch.float64, device=device)
        loss = self.cost(out, y, action)
        self._loss_t = loss.detach()  # read lazily (`temp_loss`): a `.item()` here would stall the host on every update
        loss.backward()
        allreduce_gradients(model, optimizer=optimizer)
        for param in model.parameters():
            param.grad.data.clamp_(-self.max_grad_norm, self.max_grad_norm)
        optimizer.step()

    def _train_begin(self, data, action, y, device, model, optimizer):
        """First half of the fused update: the same arithmetic as `train`'s framework path as a fixed sequence of HIP
        launches without an autograd graph - trunk forward, cost and its gradient (drlgx_dqn_loss_grad), trunk backward
        straight into the optimiser's flat gradient buffer - and, with several ranks, the gradient all-reduce ISSUED (one
        in-place SUM over the flat buffer, on the backend's own stream).  Returns the handle for `_train_end`."""
        x = data.x
        n = x.shape[0]
        mask = NW._dropout_mask(n, 1000, 0.5, x.device)
        out, saved = NW.gcn_forward_raw(x, data.edge_index, data.edge_attr, model.trunk_parameters(), mask, NW.graph_segments(data))
        y = torch.as_tensor(y, dtype=torch.float64, device=device)
        action = torch.as_tensor(action, dtype=torch.float64, device=device)
        loss = torch.empty(1, dtype=torch.float64, device=x.device)
        d_out = torch.empty(n, 1, dtype=torch.float32, device=x.device)
        vp = C.c_void_p
        stream = _lib.stream_ptr(x.device)
        _lib.check(_lib.lib().drlgx_dqn_loss_grad(vp(stream), n, vp(out.data_ptr()), vp(action.data_ptr()), vp(y.data_ptr()),
                                                  float(self.BATCH), vp(loss.data_ptr()), vp(d_out.data_ptr())))
        self._loss_t = loss
        NW.gcn_backward_raw(saved, d_out, optimizer.grads())
        optimizer.bucket.start()
        return model, optimizer

    def _train_end(self, handle):
        """Second half: wait for the exchange (a stream dependency, the host does not block on NCCL/RCCL), then clamp + Adam
        in one kernel with the 1 / world of the averaged gradient folded in (drlgx_adam_step_scaled)."""
        model, optimizer = handle
        scale = optimizer.bucket.finish()
        if optimizer.grad_clamp != self.max_grad_norm:
            optimizer.bucket.flat.mul_(scale).clamp_(-self.max_grad_norm, self.max_grad_norm)
            scale = 1.0
        optimizer.step(grad_scale=scale)

    # ------------------------------------------------------------------ the fused update of pooled mini-batches
    class _Arena(object):
        """ONE device buffer for every intermediate of an update (include/drlgx.h: drlgx_dqn_arena_bytes), grown in steps; the
        views the Python side reads (loss, out, a_batch, y_batch) are slices of it."""

        def __init__(self, device, k, cap_n, cap_e, cap_n1, dims):
            L = _lib.lib()
            self.k, self.cap = k, (int(cap_n), int(cap_e), int(cap_n1))
            self.dims = dims  # (in_dim, hidden, out_dim)
            nbytes = L.drlgx_dqn_arena_bytes(k, *self.cap, *dims)
            if nbytes == 0:
                raise _lib.DrlgxError("drlgx_dqn_arena_bytes: invalid capacities %r" % (self.cap,))
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.ptr = C.c_void_p(self.buf.data_ptr())
            v = (C.c_void_p * 12)()
            _lib.check(L.drlgx_dqn_arena_views(self.ptr, k, *self.cap, *dims, v))
            off = [int(a) - self.buf.data_ptr() for a in v]
            self.off = dict(zip(("x", "ei", "ea", "bt", "node_off", "edge_off", "q1", "a", "y", "out", "d_out", "loss"), off))
            self.loss = self.buf[off[11]:off[11] + 8].view(torch.float64)

        def fits(self, k, n, e, n1, dims):
            return k == self.k and dims == self.dims and n <= self.cap[0] and e <= self.cap[1] and n1 <= self.cap[2]

        def view(self, name, count, dtype):
            o = self.off[name]
            return self.buf[o:o + count * torch.empty(0, dtype=dtype).element_size()].view(dtype)

    def _arena_for(self, device, k, n, e, n1, dims):
        a = self.__dict__.get("_arena")
        if a is None or a.buf.device != torch.device(device) or not a.fits(k, n, e, n1, dims):
            grow = lambda v, old: max(int(v * 1.25) + 64, old)  # noqa: E731
            old = a.cap if a is not None and a.k == k and a.dims == dims else (0, 0, 0)
            # (the previous arena may still be read by launches in flight: stream-ordered free of the caching allocator)
            a = self.__dict__["_arena"] = DeepQ._Arena(device, k, grow(n, old[0]), grow(e, old[1]), grow(n1, old[2]), dims)
        return a

    def _fused_prepare(self, pr, device, dims):
        """Collation of s_j, the cached target read-out of s_j1 and the TD targets of one prepared mini-batch: one host call
        (drlgx_dqn_prepare), results in the arena."""
        pool, B = pr["pool"], self.BATCH
        # (the arena is carved with dims[0] as the feature width - x comes first, every offset behind it follows - and
        # drlgx_dqn_prepare carves it with the pool's: the two must be one number)
        if pool.X.shape[1] != dims[0]:
            raise ValueError("replay pool features are %d wide, the network's first layer takes %d" % (pool.X.shape[1], dims[0]))
        a = self._arena_for(device, B, pr["N"], pr["E"], pr["N1"], dims)
        vp = C.c_void_p
        _lib.check(_lib.lib().drlgx_dqn_prepare(vp(_lib.stream_ptr(device)), B, vp(pr["p_desc_j"]), vp(pr["p_desc_j1"]), vp(pool.X.data_ptr()),
                                                pool.X.shape[1], vp(pool.EI.data_ptr()), pool.EI.shape[1], vp(pool.EA.data_ptr()),
                                                vp(pool.Q.data_ptr()), pr["N"], pr["E"], pr["N1"], vp(pr["p_meta"]), vp(pr["p_r"]),
                                                float(self.GAMMA), a.ptr, *a.cap, dims[1], dims[2], C.byref(pool.csr) if pr["csr"] else None))
        return a

    def _fused_forward_backward(self, pr, arena, device, model, optimizer):
        """Trunk forward, cost + its gradient, trunk backward into the optimiser's flat gradient buffer: one host call
        (drlgx_dqn_forward_backward); then the gradient exchange is issued (several ranks).  Handle for `_train_end`."""
        n = pr["N"]
        in_dim, hidden, out_dim = arena.dims
        mask = NW._dropout_mask(n, hidden, 0.5, arena.buf.device)
        vp = C.c_void_p
        params = (vp * 6)(*[t.data_ptr() for t in model.trunk_parameters()])
        grads = (vp * 6)(*[g.data_ptr() for g in optimizer.grads()])
        _lib.check(_lib.lib().drlgx_dqn_forward_backward(vp(_lib.stream_ptr(device)), self.BATCH, n, pr["E"], pr["ME"], in_dim, hidden, out_dim,
                                                         params, vp(mask.data_ptr()) if mask is not None else None, float(self.BATCH), grads,
                                                         arena.ptr, *arena.cap, 1 if pr["csr"] else 0))
        self._loss_t = arena.loss
        optimizer.bucket.start()
        return model, optimizer

    def test(self, data, prob, device, model):
        model.eval()
        data = data.to(device)
        return model(data, prob)

    # ------------------------------------------------------------------ one mini-batch (policy.py:139-177)
    def td_targets(self, minibatch, q1, device):
        """(a_batch, y_batch) [sum of current-state nodes] float64 for a list of transitions
        (s_j, local action node, reward, s_j1, terminal, fro_size1) and the target network's read-out `q1` over the
        collated next states.

        `self.target_window == "reference"` (default) restates the reference's loop literally (policy.py:154-175): the
        read-out window of sample i is `q1[start_p : start_p + n_i][-fro1_i:]` with `start_p` advanced by the
        CURRENT-state node counts n_i, although q1 is laid out by the next-state counts - so the window drifts away from
        sample i's own next-state frontier nodes whenever graphs grew during the step (quirks are
        restated, not fixed).  `"aligned"` reads the last fro1_i nodes of sample i's own next-state graph."""
        meta, r, n_tot = self._td_meta(minibatch, int(q1.numel()))
        return self._td_apply(q1, torch.from_numpy(meta).to(device), torch.from_numpy(r).to(device), len(minibatch), n_tot,
                              meta if not q1.is_cuda else None, r)

    def _td_meta(self, minibatch, n1_tot, n_j=None):
        """Host part of `td_targets`: int64 [4, B] = lo, hi (the read-out window of every sample in the collated next-state
        read-out), pos (the sample's action node among the collated current-state nodes), terminal; float64 rewards [B];
        the current-state node total.  Raises like numpy would on an empty window."""
        s_j, a_loc, rew, _, term, fro1 = zip(*minibatch)
        n_j = np.array([g.num_nodes for g in s_j], dtype=np.int64) if n_j is None else n_j
        fro1 = np.array(fro1, dtype=np.int64)
        a_loc = np.array(a_loc, dtype=np.int64)
        term = np.array(term, dtype=bool)
        off_j = np.cumsum(n_j) - n_j
        if self.target_window == "reference":
            lo = np.minimum(off_j, n1_tot)               # python slicing clips to the array
            hi = np.minimum(off_j + n_j, n1_tot)
        else:
            n_j1 = np.array([d[3].num_nodes for d in minibatch], dtype=np.int64)
            hi = np.cumsum(n_j1)
            lo = hi - n_j1
        lo = np.maximum(lo, hi - fro1)                   # [-fro1:] of the slice
        if bool(((hi <= lo) & ~term).any()):
            raise ValueError("zero-size array to reduction operation maximum which has no identity")  # as numpy would
        meta = np.stack([lo, hi, off_j + a_loc, term.astype(np.int64)])
        return meta, np.array(rew, dtype=np.float64), int(n_j.sum())

    def _td_apply(self, q1, meta_dev, r_dev, B, n_tot, meta_host=None, r_host=None):
        """(a_batch, y_batch) from the read-out and the windows: drlgx_dqn_targets on the device; plain tensor ops for
        host tensors (the host-logic tests)."""
        if q1.is_cuda:
            a_batch = torch.empty(n_tot, dtype=torch.float64, device=q1.device)
            y_batch = torch.empty(n_tot, dtype=torch.float64, device=q1.device)
            vp = C.c_void_p
            stream = _lib.stream_ptr(q1.device)
            q1 = q1.contiguous()
            _lib.check(_lib.lib().drlgx_dqn_targets(vp(stream), B, vp(q1.data_ptr()), vp(meta_dev.data_ptr()), vp(r_dev.data_ptr()),
                                                    float(self.GAMMA), n_tot, vp(a_batch.data_ptr()), vp(y_batch.data_ptr())))
            return a_batch, y_batch
        a_batch = torch.zeros(n_tot, dtype=torch.float64)
        y_batch = torch.zeros(n_tot, dtype=torch.float64)
        for i in range(B):
            lo, hi, pos, term = (int(v) for v in meta_host[:, i])
            t = float(r_host[i])
            if not term:
                t = t + self.GAMMA * float(q1[lo:hi].max())  # float32 maximum, float64 arithmetic (policy.py:171-172)
            a_batch[pos] = 1.0
            y_batch[pos] = t
        return a_batch, y_batch

    def _collate_minibatch(self, device, target_net, prepared=None, minibatch=None):
        """(s_j batch, a_batch, y_batch) of one mini-batch.  Nothing here reads the POLICY weights (the targets come from the
        target network), so the loop may run it while the previous update's gradient exchange is still in flight."""
        if prepared is None:
            if minibatch is None:  # (a caller that sampled already - _prepare_updates' fallback - hands its sample in)
                minibatch = random.sample(self.buffer, self.BATCH)
            s_j = GraphData.collate([d[0] for d in minibatch])
            s_j1 = GraphData.collate([d[3] for d in minibatch])
            with torch.no_grad():
                q1 = self.test(s_j1, 0.0, device, target_net).view(-1)
            a_batch, y_batch = self.td_targets(minibatch, q1, device)
        else:  # everything the host contributes was uploaded in one piece by _prepare_updates
            pool, B = prepared["pool"], self.BATCH
            # the target network's read-out over the next states (`self.test(s_j1_batch, 0.0, device, target_net)`): a
            # function of the graph and the frozen target weights only, evaluated once per stored export and target refresh
            # (_refresh_target_readout) and gathered here, in the launch that collates the current states
            s_j = pool.collate_from(prepared["desc_j"], B, prepared["N"], prepared["E"], prepared["ME"], q_desc=prepared["desc_j1"],
                                    q_nodes=prepared["N1"])
            q1 = s_j.q
            a_batch, y_batch = self._td_apply(q1, prepared["meta"], prepared["r"], B, prepared["N"])
        return s_j, a_batch, y_batch

    def _train_minibatch(self, device, policy_net, target_net, optimizer, prepared=None, minibatch=None):
        s_j, a_batch, y_batch = self._collate_minibatch(device, target_net, prepared, minibatch)
        self.train(s_j, a_batch, y_batch, device, policy_net, optimizer)

    def _train_minibatches(self, device, policy_net, target_net, optimizer, prepared, batches, n_upd):
        """`n_upd` updates.  On the fused path the next mini-batch is collated (and its targets gathered) between an update's
        backward pass and its Adam step, i.e. while that update's gradient all-reduce travels: the collective is hidden
        behind work that does not depend on it.  Pool-backed mini-batches (`prepared`) take two host calls per update
        (`_fused_prepare`, `_fused_forward_backward`) plus the Adam launch; the generic path issues the same launches one by one."""
        fused = type(policy_net) is GCN and isinstance(optimizer, FusedAdam)
        pending = None
        if fused and prepared is not None and self.target_window in ("reference", "aligned") and self.__dict__.get("fused_update", True):
            if not policy_net.training:
                policy_net.train()
            W1, _, _, _, Wf, _ = policy_net.trunk_parameters()
            dims = (int(W1.shape[0]), int(W1.shape[1]), int(Wf.shape[0]))
            if dims[2] == 1:
                for u in range(n_upd):
                    arena = self._fused_prepare(prepared[u], device, dims)
                    if pending is not None:
                        self._train_end(pending)
                    pending = self._fused_forward_backward(prepared[u], arena, device, policy_net, optimizer)
                if pending is not None:
                    self._train_end(pending)
                return
        for u in range(n_upd):
            s_j, a_batch, y_batch = self._collate_minibatch(device, target_net, None if prepared is None else prepared[u],
                                                            None if prepared is not None else batches[u])
            if pending is not None:
                self._train_end(pending)
                pending = None
            if fused and s_j.x.is_cuda:
                if not policy_net.training:
                    policy_net.train()
                pending = self._train_begin(s_j.to(device), a_batch, y_batch, device, policy_net, optimizer)
            else:
                self.train(s_j, a_batch, y_batch, device, policy_net, optimizer)
        if pending is not None:
            self._train_end(pending)

    def _refresh_target_readout(self, pool, slots, device, target_net):
        """Target-network read-out (dropout off) of every export in `slots` whose cached values are older than the current
        target weights: one forward over the export's whole batch, stored beside the pooled graphs."""
        ver = self.__dict__.setdefault("_target_version", 0)
        for slot in slots:
            if pool.q_version[slot] == ver:
                continue
            g = pool.export(slot)
            with torch.no_grad():
                q = self.test(g, 0.0, device, target_net).view(-1)
            pool.Q[slot * pool.cap_nodes:slot * pool.cap_nodes + q.numel()] = q
            pool.q_version[slot] = ver

    def _prepare_updates(self, n_upd, device, target_net):
        """Sample `n_upd` mini-batches (the buffer does not change between them, so sampling them up front draws the same
        transitions as sampling before every update) and upload everything the host contributes to the updates - the
        collation descriptors of s_j / s_j1, the TD-target windows, the rewards - as ONE int64 and ONE float64 tensor.
        Returns a list of `prepared` dicts for `_train_minibatch`, or None when the buffer is not pool-backed.

        `random.sample(range(len(buffer)), BATCH)` makes the draws of `random.sample(buffer, BATCH)` (the population's length
        is all the sampler looks at), so the indices ARE the reference's sample (policy.py:141); everything per transition
        then comes out of the buffer's numeric side table with a few array operations for all updates at once."""
        B = self.BATCH
        tab = self.buffer.table() if isinstance(self.buffer, ReplayList) else None
        if tab is None:
            return None, [random.sample(self.buffer, B) for _ in range(n_upd)]
        pool, rows, rew = tab
        idx = np.array([random.sample(range(len(self.buffer)), B) for _ in range(n_upd)], dtype=np.int64)  # [n_upd, B]
        T = rows[idx]                                                                                       # [n_upd, B, 14]
        I = np.empty((n_upd, 14, B), dtype=np.int64)
        I[:, 0:10] = T[:, :, 0:10].transpose(0, 2, 1)
        n_j, n_j1 = T[:, :, 1], T[:, :, 6]
        N, E, N1, E1 = n_j.sum(1), T[:, :, 3].sum(1), n_j1.sum(1), T[:, :, 8].sum(1)
        ME, ME1 = T[:, :, 3].max(1), T[:, :, 8].max(1)
        # the TD-target windows (`_td_meta`, for every update at once)
        off_j = np.cumsum(n_j, axis=1) - n_j
        term = T[:, :, 12] != 0
        if self.target_window == "reference":
            lo, hi = np.minimum(off_j, N1[:, None]), np.minimum(off_j + n_j, N1[:, None])
        else:
            hi = np.cumsum(n_j1, axis=1)
            lo = hi - n_j1
        lo = np.maximum(lo, hi - T[:, :, 11])
        if bool(((hi <= lo) & ~term).any()):
            raise ValueError("zero-size array to reduction operation maximum which has no identity")  # as numpy would
        I[:, 10], I[:, 11], I[:, 12], I[:, 13] = lo, hi, off_j + T[:, :, 10], T[:, :, 12]
        R = rew[idx]
        # the pool's per-graph CSR cache serves a mini-batch when every export it draws its current states from is cached
        csr = pool.csr is not None and bool(pool.csr_ok[np.unique(T[:, :, 0] // pool.cap_nodes)].all())
        self._refresh_target_readout(pool, np.unique(T[:, :, 13]).tolist(), device, target_net)
        I_dev, R_dev = torch.from_numpy(I).to(device), torch.from_numpy(R).to(device)
        pI, pR, sI, sR = I_dev.data_ptr(), R_dev.data_ptr(), 14 * B * 8, B * 8
        N, E, N1, E1, ME, ME1 = (v.tolist() for v in (N, E, N1, E1, ME, ME1))
        prepared = [_Prepared(pool=pool, csr=csr, _I=I_dev, _R=R_dev, _u=u, N=N[u], E=E[u], N1=N1[u], E1=E1[u], ME=ME[u], ME1=ME1[u],
                              p_desc_j=pI + u * sI, p_desc_j1=pI + u * sI + 5 * B * 8, p_meta=pI + u * sI + 10 * B * 8, p_r=pR + u * sR)
                    for u in range(n_upd)]
        return prepared, idx

    # ------------------------------------------------------------------ main loop (policy.py:60-208)
    def running(self, model, modelTarget, test=False, n_envs=64, env=None, log_every=0):
        temp_i = 0
        method = "bayesian"
        own_env = env is None
        if env is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            env = VecExplorationEnv(self.map_size, n_envs, env_index=rank * n_envs, test=test,
                                    device=torch.cuda.current_device(), seed=None if not test else rank)
        n_envs = env.n_envs
        device = env.device
        policy_net, target_net = model, modelTarget
        target_net.eval()
        # graphs of new transitions live in a device pool (one slot per batched export, recycled when unreferenced)
        mn, me, _ = env.engine.graph_capacity()
        n_slots = 2 * (int(math.ceil(self.REPLAY_MEMORY / n_envs)) + 2)
        pool = getattr(self, "_pool", None)
        if pool is None or pool.device != device or pool.cap_nodes < mn or pool.cap_edges < me or pool.n_slots < n_slots:
            pool = self._pool = ReplayPool(device, n_slots, mn, me, cache_csr=True)
        # a replay buffer re-loaded from saved_training.pkl holds host graphs: they go into the pool too (packed into
        # slot-sized synthetic exports, a few large copies), so that the updates after a reload take the same one-gather
        # collation and cached target read-out as before it; what does not fit stays a device GraphData (generic path)
        self._repool(pool, device)

        def release(t):
            for d in (t[0], t[3]):
                if isinstance(d, PoolRef):
                    d.pool.ref[d.slot] -= 1
        broadcast_parameters(policy_net)
        broadcast_parameters(target_net)
        self._target_version = self.__dict__.get("_target_version", 0) + 1  # (the caller may hand in another target network)
        # torch.optim.Adam(policy_net.parameters(), lr=1e-5) of the reference with the gradient clamp fused in front
        optimizer = FusedAdam(policy_net.parameters(), lr=1e-5, grad_clamp=self.max_grad_norm)
        temp_reward_data, temp_loss_data, rows = [], [], []
        recent = deque(self.total_reward[-1000:].tolist(), maxlen=1000)  # average reward window (policy.py:201-203)

        g = self._host_offsets(env.graph_matrix(plan=_EXPORT_PLAN))
        slot_t = pool.put(g)
        pool.ref[slot_t] += 1  # the current state's export is held until the next one replaces it
        with _QuietGc() as quiet:  # (the cyclic collector runs at the updates, not between two launches)
            while temp_i < self.epoch:
                if self.epsilon > self.FINAL_EPSILON and self.step_t > self.OBSERVE:
                    self.epsilon -= n_envs * (self.INITIAL_EPSILON - self.FINAL_EPSILON) / self.EXPLORE
                env.actions_all_goals()
                rewards = env.rewards_all_goals()
                cand_env, cand_node, cand_first = env.candidates
                nfr = g["n_frontier"].long()
                batch_data = GraphData(g["x"], g["edge_index"], g["edge_attr"], g["batch"], g["node_off"], g["edge_off"], g["max_graph_edges"])
                with torch.no_grad():
                    prob = self.epsilon if method == "bayesian" else 0.0
                    readout = self.test(batch_data, prob, device, policy_net).view(-1)
                q_c = readout[cand_node]
                e = cand_env.long()
                best = torch.full((n_envs,), -float("inf"), device=device).scatter_reduce(0, e, q_c, reduce="amax")
                idx = torch.arange(q_c.numel(), device=device)
                pick = torch.full((n_envs,), q_c.numel(), dtype=torch.long, device=device).scatter_reduce(
                    0, e[q_c >= best[e]], idx[q_c >= best[e]], reduce="amin")  # np.argmax: first maximum
                if method == "e-greedy":
                    explore = torch.rand(n_envs, device=device) <= self.epsilon
                    rnd = cand_first + (torch.rand(n_envs, device=device) * nfr).long().clamp(max=nfr - 1)
                    pick = torch.where(explore, rnd, pick)
                choice = pick - cand_first
                r_t = rewards[pick]
                key_size = (g["node_off"][1:] - g["node_off"][:-1]).long() - nfr
                _, done, _ = env.step(choice, check=False)
                # (the step's results in one synchronisation, with the status check)
                a_loc, current_done, done_h, r_h, trunc_h = env.engine.fetch(key_size + choice, done | env.loop_clo, done, r_t, env.truncated())

                # next state = the graph after the step, BEFORE a finished env is re-created (policy.py:127-133 store s_t1,
                # then `env = ExplorationEnv(...)` at :185-189); envs that ran out of pose capacity are re-created too, but
                # their transition stays non-terminal (VecExplorationEnv.truncated)
                renew = done_h | trunc_h
                g1 = self._host_offsets(env.graph_matrix(plan=_EXPORT_PLAN))
                nfr1 = g1["n_frontier_h"] if "n_frontier_h" in g1 else g1["n_frontier"].cpu().numpy()
                slot_t1 = pool.put(g1)
                refs_t, refs_t1 = PoolRef.many(pool, slot_t, n_envs), PoolRef.many(pool, slot_t1, n_envs)
                a_l, r_l, d_l, f_l = a_loc.tolist(), r_h.tolist(), current_done.tolist(), nfr1.tolist()
                for i in range(n_envs):
                    self.buffer.append((refs_t[i], int(a_l[i]), float(r_l[i]), refs_t1[i], bool(d_l[i]), int(f_l[i])))
                    pool.ref[slot_t] += 1
                    pool.ref[slot_t1] += 1
                    if len(self.buffer) > self.REPLAY_MEMORY:
                        release(self.buffer.popleft())
                if renew.any():
                    env.reset(np.nonzero(renew)[0])
                    g1 = self._host_offsets(env.graph_matrix(plan=_EXPORT_PLAN))
                    slot_t1 = pool.put(g1)
                pool.ref[slot_t1] += 1
                pool.ref[slot_t] -= 1
                g, slot_t = g1, slot_t1  # the next iteration's s_t
                self.step_t += n_envs
                temp_i += n_envs

                # the reference trains once per environment step (policy.py:137-178): n_envs mini-batches per vector step
                # unless `updates_per_vector_step` says otherwise; the target network is refreshed when the env-step counter
                # passes a multiple of TARGET_UPDATE
                if self.step_t > self.OBSERVE and len(self.buffer) >= self.BATCH:
                    n_upd = n_envs if self.updates_per_vector_step is None else int(self.updates_per_vector_step)
                    if self.step_t // self.TARGET_UPDATE > (self.step_t - n_envs) // self.TARGET_UPDATE:
                        target_net.load_state_dict(policy_net.state_dict())
                        self._target_version = self.__dict__.get("_target_version", 0) + 1  # cached target read-outs are stale
                    prepared, batches = self._prepare_updates(n_upd, device, target_net)
                    self._train_minibatches(device, policy_net, target_net, optimizer, prepared, batches, n_upd)
                    temp_loss_data.append([self.step_t, self.temp_loss])
                quiet.tick()  # (every vector step: n_envs mini-batches each)

                if log_every and (self.step_t // n_envs) % log_every == 0:
                    print("TIMESTEP", self.step_t, "/ EPSILON", self.epsilon, "/ Q_MAX %e" % float(readout.max()),
                          "/ EXPLORED", float(env.status().mean()), "/ REWARD", float(r_h.mean()))
                rows.extend([self.step_t, float(x)] for x in r_h)
                recent.extend(float(x) for x in r_h)
                if self.step_t // 5e4 > (self.step_t - n_envs) // 5e4:  # every 50000 iterations (policy.py:197-199)
                    save_state_dict(policy_net, os.path.join(self.weights_path, "MyModel.pt"))
                if self.step_t > 1000 and self.step_t // 100 > (self.step_t - n_envs) // 100:  # every 100 (policy.py:200-203)
                    temp_reward_data.append([self.step_t, float(np.average(recent))])

        self.total_reward = np.append(self.total_reward, np.array([r[1] for r in rows]))
        if _is_rank0():  # (data-parallel runs: one writer for the shared artefact files; every rank logs its own envs' rewards in memory)
            np.savetxt(os.path.join(self.object_path, "temp_reward.csv"), np.array(temp_reward_data).reshape(-1, 2), delimiter=",")
            np.savetxt(os.path.join(self.object_path, "temp_loss.csv"), np.array(temp_loss_data).reshape(-1, 2), delimiter=",")
            with open(os.path.join(self.reward_data_path, "reward_data.csv"), "a", newline="") as f:
                csv.writer(f).writerows(rows)
        save_state_dict(policy_net, os.path.join(self.object_path, "Model_Policy.pt"))
        save_state_dict(target_net, os.path.join(self.object_path, "Model_Target.pt"))
        save_state_dict(policy_net, os.path.join(self.weights_path, "MyModel.pt"))
        pool.ref[slot_t] -= 1
        if own_env:
            env.close()


def sample_frontiers(p, n_frontier, rng=np.random):
    """`np.random.choice(fro, 1, p=p_env / p_env.sum())[0]` for every env (scripts/policy.py:392-394) at once.  `p` holds the
    envs' frontier probabilities back to back, `n_frontier` their counts.  Legacy RandomState.choice draws ONE uniform per
    call and searches the normalised cumulative sum, so n_envs uniforms from the same stream in env order pick exactly the
    actions the reference's per-env calls would."""
    n_frontier = np.asarray(n_frontier, dtype=np.int64)
    n = len(n_frontier)
    u = rng.random_sample(n)
    width = int(n_frontier.max())
    live = np.arange(width)[None, :] < n_frontier[:, None]
    P2 = np.zeros((n, width))
    P2[live] = np.asarray(p, dtype=np.float64)
    P2 /= P2.sum(axis=1, keepdims=True)                      # p / p.sum()
    cdf = np.cumsum(P2, axis=1)
    cdf /= cdf[np.arange(n), n_frontier - 1][:, None]        # choice(): cdf /= cdf[-1]
    return ((cdf <= u[:, None]) & live).sum(axis=1).astype(np.int64)  # cdf.searchsorted(u, side="right")


class A2C(object):
    """The reference's advantage actor-critic trainer (scripts/policy.py:262-503) over a `VecExplorationEnv`.

    Same constructor `A2C(case_path)`, hyper-parameters (nstep 40, gamma 0.99, entropy 0.01, value 0.25, grad clamp
    0.5, Adam 1e-5 over actor + critic), costs and artefacts.  `running` steps `n_envs` environments in lock-step; every
    env keeps its own n-step trajectory, and one optimiser step per `nstep` vector steps uses all `n_envs * nstep`
    transitions with the reference's per-trajectory normalisation averaged over the envs (n_envs = 1 is exactly the
    reference's update).  Multi-GPU: gradients of both networks are averaged over ranks before the clamp."""

    def __init__(self, case_path, data_root="../data"):
        self.case_path = case_path
        self.weights_path = os.path.join(data_root, "torch_weights", self.case_path)
        self.reward_data_path = os.path.join(data_root, "reward_data", self.case_path)
        self.object_path = os.path.join(data_root, "training_object_data", self.case_path)
        for p in (self.weights_path, self.reward_data_path, self.object_path):
            os.makedirs(p, exist_ok=True)
        with open(os.path.join(self.reward_data_path, "reward_data.csv"), "w", newline="") as f:
            csv.writer(f).writerow(["Step", "Reward"])
        self.GAMMA = 0.99
        self.EXPLORE = 1e6
        self.epoch = 1e4
        self.nstep = 40
        self.ent_coef = 0.01
        self.vf_coef = 0.25
        self.max_grad_norm = 0.5
        self.graphs_per_pass = 256  # forward/backward chunk of the n_envs * nstep graphs (gradients accumulate)
        self.buffer = deque()
        self.map_size = 40
        self.step_t = 0
        self.temp_loss = 0
        self.entro = 0
        self.total_reward = np.empty([0, 0])

    data_process = staticmethod(DeepQ.data_process)
    _host_offsets = staticmethod(DeepQ._host_offsets)

    def __getstate__(self):
        st = dict(self.__dict__)
        st["buffer"] = deque(([_graph_to_host(d) for d in b[0]],) + tuple(b[1:]) for b in self.buffer)
        st.pop("_pool", None)  # device storage: the pickled window carries its graphs itself
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    # ------------------------------------------------------------------ costs (policy.py:452-472)
    def policy_cost(self, prob, advantages, action, mask):
        adv = torch.masked_select(advantages.view(-1), mask)
        act = torch.masked_select(action, mask)
        return torch.mul(-torch.mul(prob.view(-1).log(), adv), act).sum() / self.nstep

    def value_cost(self, pred, target):
        return torch.nn.functional.mse_loss(pred.view(-1), target.view(-1))

    def entropy_loss(self, prob):
        p = prob.view(-1).detach()
        return -torch.mul(p.log(), p).sum() / self.nstep

    def train(self, data, action, mask, dis_reward, y_adv, device, modelA, modelC, optimizer, n_traj=1):
        """One optimiser step on a batch of `n_traj` trajectories of `nstep` graphs (policy.py:474-497 for n_traj = 1).
        `data` is a list of GraphData (one per transition, trajectory-major) or one collated batch."""
        modelA.train()
        modelC.train()
        items = data if isinstance(data, (list, tuple)) else None
        if items is None:
            chunks = [(data, 0, int(data.batch.max().item()) + 1)]
        else:
            # (collated chunk by chunk INSIDE the loop below: the host prepares chunk k + 1 while the device runs chunk k - collated
            # up front, forty collations left the device idle for ~115 us each before the first pass started - and a chunk's tensors
            # are released when its pass is done)
            def lazy_chunks():
                for a in range(0, len(items), self.graphs_per_pass):
                    b = min(len(items), a + self.graphs_per_pass)
                    yield GraphData.collate(items[a:b]), a, b
            chunks = lazy_chunks()
        # a mask that arrives as a host array gives the masked nodes' positions without asking the device: the chunks below then
        # run without a single synchronisation (torch.masked_select and the actor's softmax size would drain the stream per chunk)
        mask_h = None if torch.is_tensor(mask) else np.asarray(mask, dtype=bool)
        mask = torch.as_tensor(mask, dtype=torch.bool, device=device)
        y_adv = torch.as_tensor(y_adv, dtype=torch.float32, device=device)
        dis_reward = torch.as_tensor(dis_reward, dtype=torch.float32, device=device)
        action = torch.as_tensor(action, dtype=torch.float32, device=device)
        sel_all = None if mask_h is None else torch.as_tensor(np.nonzero(mask_h)[0], device=device)
        csum_h = None if mask_h is None else np.concatenate([[0], np.cumsum(mask_h)])
        optimizer.zero_grad()
        n_graphs = len(items) if items is not None else chunks[0][2]
        node0 = 0
        total = torch.zeros((), dtype=torch.float64, device=device)  # (summed like the host's doubles did)
        entro = torch.zeros((), dtype=torch.float64, device=device)
        for cdata, g0, g1 in chunks:
            cdata = cdata.to(device)
            nn_ = cdata.x.shape[0]
            m = mask[node0:node0 + nn_]
            if mask_h is not None:
                cdata.n_masked = int(csum_h[node0 + nn_] - csum_h[node0])
            actor_out = modelA(cdata, m, batch=cdata.batch) + 1e-35
            critic_out = modelC(cdata, m, batch=cdata.batch)
            if mask_h is not None:  # policy_cost with its two masked_select as gathers (same elements, same order)
                sel = sel_all[int(csum_h[node0]):int(csum_h[node0 + nn_])]
                actor_loss = torch.mul(-torch.mul(actor_out.view(-1).log(), y_adv.index_select(0, sel)),
                                       action.index_select(0, sel)).sum() / self.nstep / n_traj
            else:
                actor_loss = self.policy_cost(actor_out, y_adv[node0:node0 + nn_], action[node0:node0 + nn_], m) / n_traj
            # mse over all graphs of the batch: this chunk's share of the mean
            critic_loss = ((critic_out.view(-1) - dis_reward[g0:g1]) ** 2).sum() / n_graphs
            ent = self.entropy_loss(actor_out) / n_traj
            loss = actor_loss - ent * self.ent_coef + critic_loss * self.vf_coef
            loss.backward()
            total += loss.detach().double()
            entro += ent.detach().double()
            node0 += nn_
        self.temp_loss, self.entro = (float(v) for v in torch.stack([total, entro]).cpu())
        allreduce_gradients(modelA)
        allreduce_gradients(modelC)
        for param in list(modelA.parameters()) + list(modelC.parameters()):
            param.grad.data.clamp_(-self.max_grad_norm, self.max_grad_norm)
        optimizer.step()

    def test(self, data, batch, mask, device, model):
        model.eval()
        data = data.to(device)
        mask = torch.as_tensor(mask, dtype=torch.bool, device=device)
        return model(data, mask, batch=batch)

    @staticmethod
    def discounted_returns(rewards, terminal, last_value, gamma):
        """policy.py:364-369 for every env column: ret_t = r_t + gamma * ret_{t+1} * (1 - terminal_t), bootstrapped
        with the critic's value of the state after the last transition. rewards / terminal: [T, n_envs]."""
        ret = np.asarray(last_value, dtype=np.float64)
        out = np.zeros(np.shape(rewards), dtype=np.float64)
        for t in reversed(range(out.shape[0])):
            ret = rewards[t] + gamma * ret * (1.0 - np.asarray(terminal[t], dtype=np.float64))
            out[t] = ret
        return out

    # ------------------------------------------------------------------ main loop (policy.py:297-427)
    def running(self, actor, critic, test=False, n_envs=16, env=None, log_every=0):
        temp_i = 0
        own_env = env is None
        if env is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            env = VecExplorationEnv(self.map_size, n_envs, env_index=rank * n_envs, test=test,
                                    device=torch.cuda.current_device(), seed=None if not test else rank)
        n_envs = env.n_envs
        device = env.device
        policy_net, value_net = actor, critic
        broadcast_parameters(policy_net)
        broadcast_parameters(value_net)
        optimizer = torch.optim.Adam(list(policy_net.parameters()) + list(value_net.parameters()), lr=1e-5)
        temp_reward_data, temp_loss_data, rows = [], [], []
        rng = np.random  # the reference samples actions from numpy's global stream
        envs = torch.arange(n_envs, device=device)

        def frontier_mask(g):
            n = g["x"].shape[0]
            end = g["node_off"][1:].long()[g["batch"]]
            return torch.arange(n, device=device) >= end - g["n_frontier"].long()[g["batch"]]

        # the graphs of the n-step window live in a device pool (one slot per batched export): the update collates its
        # chunks out of it with one kernel each instead of a few tensor ops per graph
        mn, me, _ = env.engine.graph_capacity()
        pool = getattr(self, "_pool", None)
        if pool is None or pool.device != device or pool.cap_nodes < mn or pool.cap_edges < me or pool.n_slots < self.nstep + 2:
            pool = self._pool = ReplayPool(device, self.nstep + 2, mn, me)
        for k in range(pool.n_slots):
            pool.ref[k] = 0
        for b in self.buffer:  # (a window carried over from an earlier call keeps its own graphs)
            for k, d in enumerate(b[0]):
                if isinstance(d, PoolRef) and d.pool is pool:
                    pool.ref[d.slot] = 1
                else:  # re-loaded from saved_training.pkl (host graphs) or pooled elsewhere: onto this device, once
                    b[0][k] = d.to(device)
        g = self._host_offsets(env.graph_matrix(plan=_EXPORT_PLAN))
        slot = pool.put(g)
        pool.ref[slot] = 1
        with _QuietGc() as quiet:  # (the cyclic collector runs at the updates, not between two launches)
            while temp_i < self.epoch:
                s_t = PoolRef.many(pool, slot, n_envs)
                env.actions_all_goals()
                rewards = env.rewards_all_goals()
                cand_env, cand_node, cand_first = env.candidates
                # (a vector step synchronises four times: the export's boundaries, the plans' lengths, the actor's read-out and the
                # step's results - each with everything the host needs at that point in one copy, Engine.fetch)
                nfr_h = g["n_frontier_h"].astype(np.int64)
                batch_data = GraphData(g["x"], g["edge_index"], g["edge_attr"], g["batch"], g["node_off"], g["edge_off"], g["max_graph_edges"])
                batch_data.n_masked = int(nfr_h.sum())
                mask = frontier_mask(g)
                with torch.no_grad():
                    readout = self.test(batch_data, g["batch"], mask, device, policy_net).view(-1)  # [C], env-major
                    val = self.test(batch_data, g["batch"], mask, device, value_net).view(-1)       # [n_envs]
                (p_h,) = env.engine.fetch(readout)
                choice = sample_frontiers(p_h.astype(np.float64), nfr_h, rng)
                _, done, _ = env.step(choice, check=False)  # (first thing after the read-out: the device waits for this launch)
                r_t = rewards[env.last_pick]
                a_loc = np.diff(g["node_off_h"]).astype(np.int64) - nfr_h + choice  # key_size + choice
                # envs out of pose capacity are re-created like finished ones, but stay non-terminal (VecExplorationEnv.truncated)
                done_h, trunc_h, current_done, r_h, val_h = env.engine.fetch(done, env.truncated(), done | env.loop_clo, r_t, val)
                renew = done_h | trunc_h
                if renew.any():
                    env.reset(np.nonzero(renew)[0])
                g1 = self._host_offsets(env.graph_matrix(plan=_EXPORT_PLAN))
                slot1 = pool.put(g1)
                pool.ref[slot1] = 1
                # (a truncated env is re-created, so in the n-step return - and only there - its trajectory ends here: the
                # bootstrap value and the next state would belong to another episode)
                self.buffer.append((s_t, a_loc, r_h, current_done | done_h | trunc_h, nfr_h.copy(), val_h))
                self.step_t += n_envs
                temp_i += n_envs

                if len(self.buffer) == self.nstep:
                    with torch.no_grad():
                        b1 = GraphData(g1["x"], g1["edge_index"], g1["edge_attr"], g1["batch"], g1["node_off"], g1["edge_off"], g1["max_graph_edges"])
                        last_value = self.test(b1, g1["batch"], frontier_mask(g1), device, value_net).view(-1).cpu().numpy()
                    T = self.nstep
                    disc = self.discounted_returns(np.stack([b[2] for b in self.buffer]), np.stack([b[3] for b in self.buffer]),
                                                   last_value, self.GAMMA)
                    # trajectory-major (env by env, then time), as the reference's DataLoader over one env's buffer; the per-node
                    # vectors (one-hot action, frontier mask, advantage at the action node) are built for all graphs at once
                    items = [self.buffer[t][0][i] for i in range(n_envs) for t in range(T)]
                    nn_ = np.array([d.num_nodes for d in items], dtype=np.int64)
                    al_ = np.stack([b[1] for b in self.buffer]).T.reshape(-1)       # [env, t] -> flat
                    fro_ = np.stack([b[4] for b in self.buffer]).T.reshape(-1)
                    v_ = np.stack([b[5] for b in self.buffer]).T.reshape(-1)
                    dr_ = disc.T.reshape(-1)
                    off_ = np.cumsum(nn_) - nn_
                    tot_ = int(nn_.sum())
                    a_all = np.zeros(tot_, dtype=np.float32)
                    a_all[off_ + al_] = 1.0
                    y_all = np.zeros(tot_, dtype=np.float32)
                    y_all[off_ + al_] = (dr_ - v_).astype(np.float32)
                    local = np.arange(tot_) - np.repeat(off_, nn_)
                    m_all = local >= np.repeat(nn_ - fro_, nn_)
                    self.train(items, a_all, m_all, dr_, y_all, device, policy_net, value_net, optimizer, n_traj=n_envs)
                    temp_loss_data.append([self.step_t, self.temp_loss])
                    self.buffer.clear()
                    quiet.tick()
                    for k in range(pool.n_slots):
                        pool.ref[k] = 1 if k == slot1 else 0
                g, slot = g1, slot1

                if log_every and (self.step_t // n_envs) % log_every == 0:
                    print("TIMESTEP", self.step_t, "/ Loss", self.temp_loss, "/ Entropy", self.entro,
                          "/ EXPLORED", float(env.status().mean()), "/ REWARD", float(r_h.mean()))
                rows.extend([self.step_t, float(x)] for x in r_h)
                self.total_reward = np.append(self.total_reward, r_h)
                if self.step_t // 5e4 > (self.step_t - n_envs) // 5e4:
                    save_state_dict(policy_net, os.path.join(self.weights_path, "MyModel.pt"))
                if self.step_t > 1000 and self.step_t // 100 > (self.step_t - n_envs) // 100:
                    temp_reward_data.append([self.step_t, float(np.average(self.total_reward[-1000:]))])

        if _is_rank0():  # (data-parallel runs: one writer for the shared artefact files; every rank logs its own envs' rewards in memory)
            np.savetxt(os.path.join(self.object_path, "temp_reward.csv"), np.array(temp_reward_data).reshape(-1, 2), delimiter=",")
            np.savetxt(os.path.join(self.object_path, "temp_loss.csv"), np.array(temp_loss_data).reshape(-1, 2), delimiter=",")
            with open(os.path.join(self.reward_data_path, "reward_data.csv"), "a", newline="") as f:
                csv.writer(f).writerows(rows)
        save_state_dict(policy_net, os.path.join(self.object_path, "Model_Policy.pt"))
        save_state_dict(value_net, os.path.join(self.object_path, "Model_Value.pt"))
        if own_env:
            env.close()

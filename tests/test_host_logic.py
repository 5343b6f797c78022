"""CPU tests of the host side: the C-ABI library loads and exports every entry point include/drlgx.h declares, the
config / start-pose / value-type mirrors agree with the oracle, the batched reward normalisation and the DQN target
computation equal the reference's per-env numpy formulation, and the multi-process gradient all-reduce (gloo,
world_size 2) averages correctly. No compute entry point is called (no GPU here)."""
import ctypes as C
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import oracle as O  # noqa: E402  (checker only)


def header_symbols():
    text = open(os.path.join(ROOT, "include", "drlgx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drlgx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_entry_point():
    from drl_graph_exploration_amd import _lib
    names = header_symbols()
    assert len(names) >= 30
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), "libdrlgx.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names  # the ctypes binding covers the whole header, nothing more


def test_error_strings_and_create_without_device():
    from drl_graph_exploration_amd import _lib, default_config
    L = _lib.lib()
    for code in (0, -1, -2, -3, -4, -5):
        assert len(L.drlgx_strerror(code)) > 0
    h = C.c_void_p()
    # argument validation comes before the device: capacities beyond what the SLAM kernels' LDS tables hold are refused
    assert L.drlgx_create(C.byref(default_config(40, max_poses=4096)), 4, 0, 0, C.byref(h)) == -1 and not h.value
    assert L.drlgx_create(C.byref(default_config(40, num_landmarks=4000, max_landmarks=4000)), 4, 0, 0, C.byref(h)) == -1
    assert L.drlgx_create(C.byref(default_config(40, max_poses=1)), 4, 0, 0, C.byref(h)) == -1 and not h.value
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    rc = L.drlgx_create(C.byref(default_config(40)), 4, 0, 0, C.byref(h))
    assert rc in (-2, -4) and not h.value  # fails loudly: no CPU fallback
    from drl_graph_exploration_amd.engine import Engine
    with pytest.raises(_lib.DrlgxError):
        Engine(default_config(40), 4)


def test_config_matches_oracle_config():
    from drl_graph_exploration_amd import default_config
    for ms, nl, alg in [(40, None, 0), (60, None, 0), (100, 100, 1), (40, 60, 0)]:
        a = default_config(ms, num_landmarks=nl, algorithm=alg)
        b = O.default_config(ms, nl, alg)
        for name, _ in b._fields_:
            if hasattr(a, name):
                assert getattr(a, name) == getattr(b, name), name


def test_start_pose_matches_oracle():
    from drl_graph_exploration_amd.config import start_pose
    for lo in (0, 1, 7, 49, 50, 123):
        assert start_pose(lo, 40.0) == O.start_pose(lo, 40.0)
    st = np.random.get_state()[1][:4].copy()
    start_pose(3, 40.0)
    assert (np.random.get_state()[1][:4] == st).all()  # the global numpy stream is left untouched


def test_ini_reader_equals_default_config():
    from configparser import ConfigParser
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.pyplanner2d import config_from_ini
    cp = ConfigParser()
    cp.read_dict({
        "Sensor Model": dict(bearing_noise="0.5", range_noise="0.02", min_bearing="-179.9", max_bearing="179.9",
                             min_range="0.1", max_range="6.0"),
        "Control Model": dict(translation_noise="0.1", rotation_noise="0.2"),
        "Environment": dict(min_x="-20", max_x="20", min_y="-20", max_y="20", max_steps="5000", safe_distance="0.0"),
        "Virtual Map": dict(resolution="2.0", sigma0="1.0", num_samples="1"),
        "Simulator": dict(seed="5", lo="0", num="8", sigma_x0="0.05", sigma_y0="0.05", sigma_theta0="0.01"),
        "Planner": dict(seed="0", angle_weight="0.4", distance_weight0="5.0", distance_weight1="2.0", d_weight="0.0",
                        max_edge_length="2.0", max_nodes="0.5", occupancy_threshold="0.4", safe_distance="1.0",
                        algorithm="EM_AOPT", reg_out="false"),
    })
    c, prm = config_from_ini(cp, max_poses=41)
    d = default_config(40, max_poses=41)
    for name, _ in d._fields_:
        assert getattr(c, name) == getattr(d, name), name
    assert prm["map"].min_x == -40.0 and prm["virtual_map"].sigma0 == 1.0


def test_pose2_value_type():
    from drl_graph_exploration_amd.ss2d import Pose2, Measurement
    a, b = Pose2(1.0, 2.0, 0.3), Pose2(0.5, -0.25, 2.9)
    c = a * b
    assert c.x == pytest.approx(1.0 + math.cos(0.3) * 0.5 + math.sin(0.3) * 0.25)
    assert c.y == pytest.approx(2.0 + math.sin(0.3) * 0.5 - math.cos(0.3) * 0.25)
    assert c.theta == pytest.approx(math.atan2(math.sin(3.2), math.cos(3.2)))
    p = Measurement(math.pi / 2, 2.0).transform_from(Pose2(1.0, 1.0, math.pi / 2))
    assert (p.x, p.y) == (pytest.approx(-1.0), pytest.approx(1.0))


def test_reward_normalisation_equals_reference_interp():
    from drl_graph_exploration_amd.vecenv import normalise_rewards
    rng = np.random.RandomState(0)
    nfr = [1, 3, 2, 5, 4, 2]
    raws = [rng.randn(k) * 3 for k in nfr]
    raws[2] = np.array([1.5, 1.5])          # all equal -> nearest is the first arg-max, interp degenerate
    raws[3][0] = raws[3].max() + 1.0        # nearest frontier wins -> [-1, 0]
    raws[4][0] = raws[4].min() - 1.0        # nearest frontier loses -> [-1, 1]
    raw = torch.tensor(np.concatenate(raws))
    cand_env = torch.tensor(np.repeat(np.arange(len(nfr)), nfr))
    first = torch.tensor(np.cumsum(nfr) - np.array(nfr))
    r, loop = normalise_rewards(raw, cand_env, first, len(nfr))
    off = 0
    for i, x in enumerate(raws):
        # exploration_env.py:151-161 with key_size NaNs in front (they do not affect nanmin / nanmax / nanargmax order)
        rewards = np.concatenate([[np.nan] * 3, x])
        nearest_frontier_point = 3
        if np.nanargmax(rewards) == nearest_frontier_point:
            exp, lc = np.interp(rewards, (np.nanmin(rewards), np.nanmax(rewards)), (-1.0, 0.0)), False
        else:
            exp, lc = np.interp(rewards, (np.nanmin(rewards), np.nanmax(rewards)), (-1.0, 1.0)), True
        np.testing.assert_allclose(r[off:off + len(x)].numpy(), exp[3:], rtol=0, atol=1e-15)
        assert bool(loop[i]) == lc
        off += len(x)


class _TinyQ(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(5, 1)

    def forward(self, data, prob, batch=None):
        return self.lin(data.x)


def test_dqn_targets_equal_reference_loop(tmp_path):
    """DeepQ.td_targets builds the same (a_batch, y_batch) as the reference's numpy loop (policy.py:152-177), whose
    literal restatement lives in oracle/dqn_ref.py - including its read-out window offsets by the CURRENT-state node
    counts - in float64; the "aligned" variant reads each sample's own next-state frontier nodes."""
    import random
    from oracle import dqn_ref
    from drl_graph_exploration_amd.networks import GraphData
    from drl_graph_exploration_amd.policy import DeepQ
    torch.manual_seed(0)
    rng = np.random.RandomState(1)
    dq = DeepQ("t/", "GCN", data_root=str(tmp_path))
    dq.BATCH = 8
    assert dq.target_window == "reference"

    def graph(n):
        return GraphData(torch.randn(n, 5), torch.zeros(2, 0, dtype=torch.long), torch.zeros(0))
    for _ in range(20):
        n = int(rng.randint(4, 9))
        n1 = n + int(rng.randint(0, 4))  # graphs grow during a step (new poses / landmarks)
        fro, fro1 = int(rng.randint(1, 4)), int(rng.randint(1, 4))
        a = n - fro + int(rng.randint(fro))
        dq.buffer.append((graph(n), a, float(rng.randn()), graph(n1), bool(rng.rand() < 0.3), fro1))
    pol, tgt = _TinyQ(), _TinyQ()
    captured = {}

    def fake_train(data, action, y, device, model, optimizer):
        captured.update(a=action.clone(), y=y.clone(), data=data)
    dq.train = fake_train
    random.seed(5)
    dq._train_minibatch(torch.device("cpu"), pol, tgt, None)
    random.seed(5)
    minibatch = random.sample(dq.buffer, dq.BATCH)
    q1 = tgt(GraphData.collate([d[3] for d in minibatch]), 0.0).detach().numpy()  # [N1, 1] float32 like the reference's
    acts = []
    for d in minibatch:
        act = np.zeros(d[0].x.shape[0])
        act[d[1]] = 1
        acts.append(act)
    a_ref, y_ref = dqn_ref.reference_targets(acts, [d[2] for d in minibatch], [d[4] for d in minibatch],
                                             [d[5] for d in minibatch], q1, dq.GAMMA)
    assert captured["a"].dtype == torch.float64 and captured["y"].dtype == torch.float64
    np.testing.assert_array_equal(captured["a"].numpy(), a_ref)
    np.testing.assert_array_equal(captured["y"].numpy(), y_ref)  # same float64 arithmetic: bit-equal
    # the windows really are the drifting ones: the aligned variant differs on this data and matches its own loop
    dq.target_window = "aligned"
    a_al, y_al = dq.td_targets(minibatch, torch.as_tensor(q1).view(-1), torch.device("cpu"))
    y_exp, start = [], 0
    for d in minibatch:
        n, n1 = d[0].x.shape[0], d[3].x.shape[0]
        t = np.zeros(n)
        t[d[1]] = d[2] if d[4] else d[2] + dq.GAMMA * float(np.max(q1[start:start + n1][-d[5]:]))
        start += n1
        y_exp.append(t)
    np.testing.assert_array_equal(y_al.numpy(), np.concatenate(y_exp))
    assert not np.array_equal(y_al.numpy(), y_ref)
    # loss in float64 (policy.py:246-248), clamp and step on a plain module
    dq2 = DeepQ("t2/", "GCN", data_root=str(tmp_path))
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
    before = pol.lin.weight.detach().clone()
    dq2.BATCH = 8
    pred = pol(captured["data"], 0.5).detach().numpy()
    dq2.train(captured["data"], captured["a"], captured["y"], torch.device("cpu"), pol, opt)
    assert dq2.temp_loss == pytest.approx(dqn_ref.reference_cost(pred, y_ref, a_ref, 8), rel=1e-12)
    assert dq2.temp_loss > 0 and not torch.equal(before, pol.lin.weight.detach())
    # an empty window is the reference's numpy error, not a silent value
    small = [(graph(6), 5, 0.1, graph(2), False, 1), (graph(6), 5, 0.1, graph(2), False, 1)]
    dq.target_window = "reference"
    with pytest.raises(ValueError):
        dq.td_targets(small, torch.zeros(4), torch.device("cpu"))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from drl_graph_exploration_amd.policy import allreduce_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))
    broadcast_parameters(m)  # rank 0's init everywhere
    torch.manual_seed(7 + rank)
    x = torch.randn(16, 5)  # each rank = its own env shard
    m(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    allreduce_gradients(m)
    # numpy copies, pickled by value: tensors would travel as shared-memory file descriptors, which the parent can no
    # longer receive once this process has exited
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()], [g.numpy().copy() for g in local],
           [p.grad.numpy().copy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, l0, g0), (_, w1, l1, g1) = [(r, *[[torch.from_numpy(a) for a in part] for part in parts]) for r, *parts in res]
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)  # broadcast made the replicas identical
    for a, b, x, y in zip(g0, g1, l0, l1):
        assert torch.equal(a, b)  # every rank holds the same averaged gradient
        assert torch.allclose(a, (x + y) / 2, rtol=1e-6, atol=1e-7)


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from drl_graph_exploration_amd.optim import GradientBucket
    from drl_graph_exploration_amd.policy import broadcast_parameters
    torch.manual_seed(100 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1)).double()
    broadcast_parameters(m)
    bucket = GradientBucket(list(m.parameters()))
    views = [p.grad.data_ptr() for p in m.parameters()]
    gen = torch.Generator().manual_seed(11)
    data = torch.randn(3, world, 16, 5, generator=gen, dtype=torch.float64)  # update, rank, sample, feature
    for u in range(3):
        bucket.flat.zero_()
        (m(data[u, rank]).pow(2).sum() / 16).backward()      # autograd accumulates INTO the flat buffer's views
        assert [p.grad.data_ptr() for p in m.parameters()] == views
        bucket.start()                                         # one in-place SUM all-reduce, asynchronous
        _ = data[(u + 1) % 3, rank].sum()                      # (what the trainer does meanwhile: the next mini-batch)
        scale = bucket.finish()                                # 1 / world, to be folded into the optimiser
        assert scale == 1.0 / world
        with torch.no_grad():
            for p in m.parameters():
                p -= 0.05 * (p.grad * scale)
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_dropout_mask_and_pool_descriptors_host_side():
    """networks._dropout_mask: None at p = 0, zeros at p = 1, else 0 or 1 / (1 - p) drawn by F.dropout on a cached tensor of
    ones that grows with the request.  ReplayPool.put / PoolRef / descriptors: the collation descriptor of a pooled graph =
    (node start in the pool, node count, edge start in the pool, edge count, first node id inside its export)."""
    sys.path.insert(0, ROOT)
    import drl_graph_exploration_amd.networks as NW
    dev = torch.device("cpu")
    assert NW._dropout_mask(10, 8, 0.0, dev) is None
    assert float(NW._dropout_mask(10, 8, 1.0, dev).abs().sum()) == 0.0
    torch.manual_seed(1)
    m = NW._dropout_mask(2000, 8, 0.5, dev)
    assert m.shape == (2000, 8) and set(m.unique().tolist()) == {0.0, 2.0} and abs(float(m.mean()) - 1.0) < 0.05
    assert NW._dropout_mask(5000, 8, 0.5, dev).shape == (5000, 8) and NW._ONES[(str(dev), 8)].shape[0] >= 5000
    pool = NW.ReplayPool(dev, 3, 100, 400)
    node_off, edge_off = np.array([0, 4, 9, 15]), np.array([0, 6, 14, 30])
    g = {"x": torch.randn(15, 5), "edge_index": torch.randint(0, 15, (2, 30)), "edge_attr": torch.rand(30), "node_off_h": node_off,
         "edge_off_h": edge_off}
    pool.put(g)
    slot = pool.put(g)  # the second slot: pool offsets are not zero
    refs = [NW.PoolRef(pool, slot, e) for e in (2, 0)]
    d, n_nodes, n_edges = NW.ReplayPool.descriptors(refs)
    want = np.array([[slot * 100 + 9, 6, slot * 400 + 14, 16, 9], [slot * 100 + 0, 4, slot * 400 + 0, 6, 0]]).T
    assert d.dtype == np.int64 and np.array_equal(d, want) and (n_nodes, n_edges) == (10, 22)
    assert (refs[0].n0, refs[0].nn, refs[0].e0, refs[0].ne, refs[0].loc) == (slot * 100 + 9, 6, slot * 400 + 14, 16, 9)
    assert torch.equal(refs[0].x, g["x"][9:15]) and refs[0].num_nodes == 6


def test_gradient_bucket_rebinds_gradients_to_its_slices():
    """GradientBucket.attach keeps every parameter's .grad bound to ITS slice of the flat tensor (the slices are built once):
    untouched when it still is, re-bound after zero_grad(set_to_none=True), and a gradient tensor someone else assigned is
    copied into the slice first; slices start on 256-byte boundaries."""
    sys.path.insert(0, ROOT)
    from drl_graph_exploration_amd.optim import GradientBucket
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    params = list(m.parameters())
    bucket = GradientBucket(params)
    views = [p.grad for p in params]
    assert all((v.data_ptr() - bucket.flat.data_ptr()) % 256 == 0 for v in views)  # (the device allocation itself is 256-byte aligned)
    bucket.attach()
    assert all(p.grad is v for p, v in zip(params, views))  # nothing rebuilt, nothing rebound
    params[0].grad = None
    foreign = torch.full_like(params[1], 3.5)
    params[1].grad = foreign
    bucket.attach()
    assert all(p.grad is v for p, v in zip(params, views))
    assert torch.equal(views[1], foreign)  # the foreign gradient's values moved into the bucket
    m(torch.randn(4, 5)).sum().backward()  # autograd accumulates into the slices
    off = bucket._offsets
    for p, o in zip(params, off):
        assert torch.equal(bucket.flat[o:o + p.numel()].view_as(p), p.grad)


def test_gradient_bucket_three_updates_gloo_world_size_2():
    """The trainer's exchange (optim.GradientBucket: gradients as views of ONE flat tensor, asynchronous in-place all-reduce,
    1 / world folded into the update) over three updates on two ranks: the replicas stay bit-identical, and equal a single
    process that trains on the concatenation of both ranks' mini-batches (mean loss), up to summation order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w0, w1 = [[torch.from_numpy(a) for a in r[1]] for r in res]
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)
    torch.manual_seed(100)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1)).double()
    gen = torch.Generator().manual_seed(11)
    data = torch.randn(3, 2, 16, 5, generator=gen, dtype=torch.float64)
    for u in range(3):
        m.zero_grad()
        (m(data[u].reshape(32, 5)).pow(2).sum() / 32).backward()
        with torch.no_grad():
            for p in m.parameters():
                p -= 0.05 * p.grad
    for a, p in zip(w0, m.parameters()):
        assert torch.allclose(a, p.detach(), rtol=1e-12, atol=1e-14)


def test_vectorised_action_sampling_equals_per_env_choice():
    """A2C samples one frontier per env with np.random.choice(fro, 1, p=p / p.sum()) (scripts/policy.py:392-394);
    `sample_frontiers` does it for all envs at once from the same stream: same actions, same stream position afterwards."""
    from drl_graph_exploration_amd.policy import sample_frontiers
    gen = np.random.RandomState(5)
    nfr = gen.randint(1, 9, size=200)
    p = gen.rand(int(nfr.sum())).astype(np.float32).astype(np.float64) + 1e-35
    first = np.cumsum(nfr) - nfr
    np.random.seed(11)
    ref = np.array([np.random.choice(int(k), 1, p=p[f:f + k] / p[f:f + k].sum())[0] for f, k in zip(first, nfr)])
    after_ref = np.random.random_sample()
    np.random.seed(11)
    got = sample_frontiers(p, nfr, np.random)
    assert np.array_equal(got, ref)
    assert np.random.random_sample() == after_ref


def test_a2c_costs_and_returns(tmp_path):
    """A2C cost functions and n-step returns against hand evaluation (scripts/policy.py:364-369, 452-472)."""
    import torch
    from drl_graph_exploration_amd.policy import A2C
    a = A2C("t/", data_root=str(tmp_path))
    a.nstep = 2
    # two graphs of 3 and 4 nodes, frontier nodes = last 2 of each; actor output = probs over masked nodes
    mask = torch.tensor([0, 1, 1, 0, 0, 1, 1], dtype=torch.bool)
    prob = torch.tensor([0.25, 0.75, 0.6, 0.4])
    action = torch.tensor([0, 0, 1, 0, 0, 1, 0], dtype=torch.float32)
    adv = torch.tensor([0, 0, 0.5, 0, 0, -2.0, 0])
    expect = (-(np.log(0.75) * 0.5) - (np.log(0.6) * -2.0)) / 2
    assert float(a.policy_cost(prob, adv, action, mask)) == pytest.approx(expect, rel=1e-6)
    ent = -(0.25 * np.log(0.25) + 0.75 * np.log(0.75) + 0.6 * np.log(0.6) + 0.4 * np.log(0.4)) / 2
    assert float(a.entropy_loss(prob)) == pytest.approx(ent, rel=1e-6)
    assert float(a.value_cost(torch.tensor([1.0, 3.0]), torch.tensor([0.0, 1.0]))) == pytest.approx(2.5)
    r = np.array([[1.0, 0.5], [2.0, -1.0], [0.0, 3.0]])
    term = np.array([[False, False], [True, False], [False, False]])
    out = A2C.discounted_returns(r, term, np.array([10.0, 20.0]), 0.9)
    # env 0: t2 = 0 + .9*10 = 9; t1 terminal = 2; t0 = 1 + .9*2 = 2.8;  env 1: t2 = 3 + 18 = 21; t1 = -1 + 18.9; t0 = .5 + .9*17.9
    np.testing.assert_allclose(out, [[2.8, 0.5 + 0.9 * 17.9], [2.0, 17.9], [9.0, 21.0]], rtol=1e-12)


def test_bench_cpu_worker_subprocess():
    """bench.py's all-cores CPU baseline launches `bench.py --cpu-worker <seed> <seconds>` children: one of them, here."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "3", "0.3"], capture_output=True,
                         text=True, timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert out.returncode == 0, out.stderr[-500:]
    n, t = out.stdout.strip().split()[-2:]
    assert int(n) >= 1 and 0.3 <= float(t) < 30.0


def test_bench_launch_contract():
    """`bench.py --gpus N` without a launcher becomes its own launcher (one rank per GPU, 127.0.0.1 rendezvous - the
    command the driver uses); under a launcher a WORLD_SIZE that differs from --gpus is an error, not a silent 1-rank run."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--print-launch"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    cmd = out.stdout.split()
    assert "torch.distributed.run" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr


def test_build_alias_resolves_the_reference_import_lines():
    """`import build.ss2d as ss2d` / `import build.planner2d as planner2d` (scripts/envs/pyss2d.py:7, pyplanner2d.py:6), in a fresh
    interpreter: `compat.enable()` puts the alias package on sys.path (it is not at the repository root, where it would shadow
    the PyPA `build` module)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import drl_graph_exploration_amd.compat as compat; compat.enable()\n"
            "import build.planner2d as planner2d_alias\nimport build.ss2d as ss2d_alias\n"
            "from build import planner2d as p2, ss2d as s2\n"
            "from drl_graph_exploration_amd import planner2d, ss2d\n"
            "assert ss2d_alias is ss2d is s2 and planner2d_alias is planner2d is p2\n"
            "assert hasattr(ss2d_alias, 'Simulator2D') and hasattr(planner2d_alias, 'EMPlanner2D')\n"
            "print('alias ok')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and "alias ok" in out.stdout, out.stderr


def test_module_objects_never_join_across_simulations():
    """`ss2d.SLAM2D` / `VirtualMap` without `simulator=` join the latest Simulator2D once; a second one raises (no engine
    is created here: that happens at `add_prior`)."""
    from drl_graph_exploration_amd import ss2d
    sp, cp, ep = ss2d.BearingRangeSensorModelParameter(), ss2d.SimpleControlModelParameter(), ss2d.EnvironmentParameter()
    a = ss2d.Simulator2D(sp, cp, 1)
    b = ss2d.Simulator2D(sp, cp, 2)
    sb = ss2d.SLAM2D(ep)
    assert sb._ses is b._ses
    with pytest.raises(RuntimeError):
        ss2d.SLAM2D(ep)
    sa = ss2d.SLAM2D(ep, simulator=a)
    va = ss2d.VirtualMap(ss2d.VirtualMapParameter(ep), 0, simulator=a)
    assert sa._ses is a._ses is va._ses and b._ses.vm is None
    with pytest.raises(RuntimeError):  # b's vehicle handed to a's SLAM2D
        sa.add_prior(ss2d.VehicleBeliefState(b.vehicle, np.eye(3)))
    with pytest.raises(RuntimeError):
        va.update_probability(sb, b.sensor_model)
    assert ss2d.trajectory_distance([ss2d.Pose2(0, 0, 0), ss2d.Pose2(3, 4, 1.0)]) == pytest.approx(
        math.sqrt(25.0 + (0.5 * math.atan2(4, 3)) ** 2))


def test_replay_list_side_table_and_vectorised_minibatch_preparation(tmp_path, monkeypatch):
    """ReplayList's numeric side table stays aligned with the list through append / extend / popleft (compaction and growth
    included) and switches off for transitions that are not pool-backed; `DeepQ._prepare_updates` built from it equals the
    per-update formulation (`ReplayPool.descriptors` + `_td_meta` on `random.sample(buffer, BATCH)`) for both window modes."""
    import random
    sys.path.insert(0, ROOT)
    import drl_graph_exploration_amd.networks as NW
    from drl_graph_exploration_amd.policy import DeepQ, ReplayList
    dev = torch.device("cpu")
    rng = np.random.RandomState(0)
    n_env = 12
    pool = NW.ReplayPool(dev, 6, 2400, 2400)
    slots = []
    for s in range(5):
        nn = rng.randint(4, 30, n_env) + 30 * s  # (next states larger than current ones: no empty reference-mode window)
        ne = rng.randint(3, 40, n_env) * 2
        no, eo = np.concatenate([[0], np.cumsum(nn)]), np.concatenate([[0], np.cumsum(ne)])
        slots.append(pool.put({"x": torch.randn(int(no[-1]), 5), "edge_index": torch.zeros(2, int(eo[-1]), dtype=torch.int64),
                               "edge_attr": torch.rand(int(eo[-1])), "node_off_h": no, "edge_off_h": eo}))

    def transition(i):
        a, b = NW.PoolRef(pool, slots[i % 4], i % n_env), NW.PoolRef(pool, slots[i % 4 + 1], (i * 5) % n_env)
        return (a, int(a.num_nodes - 1 - i % 3), float(rng.randn()), b, i % 7 == 0, 1 + i % 3)
    buf = ReplayList()
    kept = []
    for i in range(9000):  # beyond the first 4096 rows: growth; with popleft: compaction
        t = transition(i)
        buf.append(t)
        kept.append(t)
        if len(buf) > 3000:
            assert buf.popleft() is kept.pop(0)
    p, rows, rew = buf.table()
    assert p is pool and rows.shape == (3000, 14) and len(buf) == 3000
    for k in (0, 1, 1499, 2999):
        t = kept[k]
        assert rows[k, 0:5].tolist() == t[0].d5.tolist() and rows[k, 5:10].tolist() == t[3].d5.tolist()
        assert rows[k, 10:14].tolist() == [t[1], t[5], int(t[4]), t[3].slot] and rew[k] == t[2]
    assert buf[5] is kept[5] and pickle_roundtrip_len(buf) == 3000
    # ---- _prepare_updates from the table == the per-update formulation
    dq = DeepQ("side/", "GCN", data_root=str(tmp_path))
    dq.BATCH = 16
    dq.buffer = buf
    monkeypatch.setattr(DeepQ, "_refresh_target_readout", lambda self, pool, slots, device, net: setattr(self, "_seen_slots", slots))
    for mode in ("reference", "aligned"):
        dq.target_window = mode
        random.seed(11)
        prepared, idx = dq._prepare_updates(5, dev, None)
        random.seed(11)
        for u in range(5):
            mb = random.sample(dq.buffer, dq.BATCH)
            assert [kept.index(t) for t in mb] == idx[u].tolist()  # the same draws as sampling the buffer itself
            d, n, e = NW.ReplayPool.descriptors([t[0] for t in mb])
            d1, n1, e1 = NW.ReplayPool.descriptors([t[3] for t in mb])
            meta, r, n_tot = dq._td_meta(mb, n1)
            pr = prepared[u]
            assert (pr["N"], pr["E"], pr["N1"], pr["E1"], pr["ME"], pr["ME1"]) == (n, e, n1, e1, int(d[3].max()), int(d1[3].max()))
            assert np.array_equal(pr["desc_j"].numpy(), d) and np.array_equal(pr["desc_j1"].numpy(), d1)
            assert np.array_equal(pr["meta"].numpy(), meta) and np.array_equal(pr["r"].numpy(), r) and n_tot == n
            assert pr["p_meta"] == pr["meta"].data_ptr() and pr["p_r"] == pr["r"].data_ptr() and pr["p_desc_j1"] == pr["desc_j1"].data_ptr()
        assert dq._seen_slots == sorted({t[3].slot for u in range(5) for t in [kept[i] for i in idx[u]]})
    # ---- a transition that is not pool-backed switches the table off (generic path)
    buf.append((object(), 0, 0.0, object(), False, 1))
    assert buf.table() is None
    prepared, batches = dq._prepare_updates(2, dev, None)
    assert prepared is None and len(batches) == 2 and len(batches[0]) == dq.BATCH


def pickle_roundtrip_len(buf):
    import pickle
    class _P(object):  # PoolRefs are not picklable as such (they hold the pool): only the container protocol is checked here
        pass
    b2 = pickle.loads(pickle.dumps(type(buf)([(1, 0, 0.0, 2, False, 1)] * len(buf))))
    assert type(b2) is type(buf) and b2.table() is None
    return len(b2)

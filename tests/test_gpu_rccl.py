"""GPU test: RCCL itself on the test box (one MI355X).  The multi-rank control flow of the trainer is covered over gloo
(tests/test_host_logic.py, tests/test_gpu_bench.py); what those cannot show is that the `nccl` backend (= RCCL on ROCm)
initialises on gfx950 and reduces the trainer's flat gradient bucket in place.  A one-rank process group does exactly that
much: communicator creation, the all-reduce launch on RCCL's stream, the stream hand-back.  An 8-GPU curve is the driver's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, socket
sys.path.insert(0, %r)
import torch, torch.distributed as dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.optim import FusedAdam
from drl_graph_exploration_amd.policy import allreduce_gradients, broadcast_parameters
torch.manual_seed(0)
model = GCN().to(dev)
for p in model.parameters():
    dist.broadcast(p.data, 0)                     # RCCL broadcast of the six parameter tensors (broadcast_parameters skips a world of 1)
opt = FusedAdam(model.parameters(), lr=1e-5, grad_clamp=0.5)
bucket = opt.bucket
opt.grads()
n = sum(p.numel() for p in model.parameters())
assert n == 1008001 and bucket.flat.numel() >= n
for p in model.parameters():
    p.grad.copy_(torch.randn_like(p))
before = bucket.flat.clone()
work = dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, async_op=True)   # the collective the bucket issues for world > 1
work.wait()
torch.cuda.synchronize()
assert torch.equal(bucket.flat, before)            # the sum over one rank
scale = allreduce_gradients(model, optimizer=opt)  # the trainer's own entry (a no-op exchange at world 1, factor 1)
w0 = [p.detach().clone() for p in model.parameters()]
opt.step(grad_scale=scale if isinstance(scale, float) else 1.0)
torch.cuda.synchronize()
assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, model.parameters()))
t = torch.ones(4, device=dev)
dist.all_reduce(t); dist.barrier()
assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("RCCL_OK", n)
""" % ROOT


def test_rccl_initialises_and_reduces_the_flat_gradient_bucket():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "RCCL_OK 1008001" in out.stdout


# ---- >= 2 GPUs: the configuration the driver measures (one rank per GPU over RCCL).  The test box has ONE MI355X, so these
# skip there; they run the moment a multi-GPU box collects them.
def _n_gpus():
    import torch
    return torch.cuda.device_count()


BUCKET_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == world
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.optim import FusedAdam
from drl_graph_exploration_amd.policy import broadcast_parameters
torch.manual_seed(100 + rank)
model = GCN().to(dev)
broadcast_parameters(model)                         # rank 0's initialisation everywhere
opt = FusedAdam(model.parameters(), lr=1e-3, grad_clamp=0.5)
bucket = opt.bucket
opt.grads()
views = [p.grad.data_ptr() for p in model.parameters()]
for u in range(3):                                  # three updates, each rank with ITS OWN gradients (its env shard)
    g = torch.Generator(device=dev).manual_seed(1000 * u + rank)
    for p in model.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g, device=dev) * 0.1)
    local = bucket.flat.clone()
    bucket.start()                                  # ONE in-place SUM all-reduce of the flat 1 008 001-float bucket, asynchronous
    scale = bucket.finish()
    assert scale == 1.0 / world and [p.grad.data_ptr() for p in model.parameters()] == views
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(bucket.flat, torch.stack(gathered).sum(0), rtol=1e-5, atol=1e-6)
    opt.step(grad_scale=scale)                      # 1 / world folded into the Adam kernel
flat_w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
gathered = [torch.empty_like(flat_w) for _ in range(world)]
dist.all_gather(gathered, flat_w)
assert all(torch.equal(gathered[0], w) for w in gathered), "replicas diverged"
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("RCCL_BUCKET_OK", world, flat_w.numel())
""" % ROOT


def _torchrun(n, script_args, timeout=900, extra_env=None):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one rank per GPU over RCCL")
def test_gradient_bucket_three_updates_nccl_two_ranks(tmp_path):
    """tests/test_host_logic.py::test_gradient_bucket_three_updates_gloo_world_size_2 on the measured transport: two ranks, one
    GPU each, `nccl` (= RCCL over xGMI).  Every rank's bucket equals the sum of all ranks' local gradients after the in-place
    all-reduce, and the replicas' parameters stay BIT-identical over three fused-Adam updates."""
    script = tmp_path / "bucket_nccl.py"
    script.write_text(BUCKET_SCRIPT)
    out = _torchrun(2, [str(script)])
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "RCCL_BUCKET_OK 2 1008001" in out.stdout


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one rank per GPU over RCCL")
def test_bench_two_gpus_over_rccl():
    """`bench.py --gpus 2` launched the way the driver launches it (torch.distributed.run, one rank per GPU): one JSON line from
    rank 0, two ranks in the group, backend nccl, a measured gradient all-reduce."""
    import json
    out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-policy"])
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_in_process_group"] == 2 and r["collective_backend"] == "nccl"
    assert r["scaling"] == "weak" and r["value"] > 0 and r["config"]["parallelism"] == "env-sharded x2"
    t = r["train_allreduce"]
    assert t["ranks"] == 2 and t["allreduce_ms"] > 0

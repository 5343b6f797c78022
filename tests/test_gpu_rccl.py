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

"""Dev tool: wall-clock rate of the whole DQN loop (DeepQ.running: graph export, look-ahead rewards, policy forward,
env step, replay, one train step per vector step) over a vectorised env."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.policy import DeepQ

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    dq = DeepQ("bench/", "GCN", data_root=tmp)
    dq.OBSERVE, dq.epoch = n_envs, n_envs * 3  # warm-up: 3 vector steps (training from the 2nd)
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    dq.running(pol, tgt, test=True, n_envs=n_envs)
    dq.epoch = n_envs * iters
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dq.running(pol, tgt, test=True, n_envs=n_envs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("n_envs %d: %.1f ms per vector step, %.0f RL iterations/s (decisions incl. replay + one train step of 64 graphs per vector step)" % (n_envs, dt / iters * 1e3, n_envs * iters / dt))

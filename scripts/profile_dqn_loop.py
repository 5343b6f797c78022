"""Dev tool: cProfile of DeepQ.running over a vectorised env (host-side cost of the training loop)."""
import os, sys, tempfile, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.policy import DeepQ
n_envs = int(sys.argv[1]); iters = 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    dq = DeepQ("bench/", "GCN", data_root=tmp)
    dq.OBSERVE, dq.epoch = n_envs, n_envs * 3
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    dq.running(pol, tgt, test=True, n_envs=n_envs)
    dq.epoch = n_envs * iters
    pr = cProfile.Profile(); pr.enable()
    dq.running(pol, tgt, test=True, n_envs=n_envs)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

"""Dev tool: where A2C.running goes (bench.py's a2c_loop workload), by synchronised timers around the env / policy calls."""
import os, sys, time, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from drl_graph_exploration_amd.networks import PolicyGCN, ValueGCN
from drl_graph_exploration_amd.policy import A2C
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n_envs, iters = 256, 40
acc = collections.OrderedDict()
def timed(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
with tempfile.TemporaryDirectory() as tmp:
    a2c = A2C("b/", data_root=tmp)
    actor, critic = PolicyGCN().to(dev), ValueGCN().to(dev)
    env = VecExplorationEnv(bench.MAP, n_envs, env_index=0, test=True, device=0)
    a2c.epoch, a2c.nstep = n_envs * 2, 2  # warm-up incl. one small update (first-use allocations)
    a2c.running(actor, critic, test=True, env=env)
    a2c.nstep = 40
    a2c.buffer.clear()
    for name, label in (("graph_matrix", "graph export"), ("actions_all_goals", "line plans"), ("rewards_all_goals", "look-ahead rewards"),
                        ("step", "env.step"), ("reset", "env.reset")):
        timed(env, name, label)
    timed(a2c, "test", "actor / critic forward (acting)")
    timed(a2c, "train", "update (all transitions)")
    a2c.epoch = n_envs * iters
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a2c.running(actor, critic, test=True, env=env)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = env.engine.counts_dev().cpu().numpy()
    print("poses at the end: mean %.1f max %d" % (c[:, 0].mean(), c[:, 0].max()))
    env.close()
print("%d envs, %d vector steps: %.1f ms per vector step (synchronised)" % (n_envs, iters, dt / iters * 1e3))
tot = 0.0
for k, v in acc.items():
    print("  %-34s %8.2f ms per vector step" % (k, v / iters * 1e3)); tot += v
print("  %-34s %8.2f ms per vector step" % ("host bookkeeping / other", (dt - tot) / iters * 1e3))

"""Dev tool: where one vector step of DeepQ.running goes (bench.py's dqn_loop workload), by synchronised timers around the
env / policy calls.  The synchronisation removes the host-device overlap, so the parts add up to more than the loop's
wall-clock rate in bench.py; it is the split that matters."""
import os, sys, time, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.policy import DeepQ
from drl_graph_exploration_amd.vecenv import VecExplorationEnv

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
acc = collections.OrderedDict()


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


dev = torch.device("cuda", 0)
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    dq = DeepQ("bench/", "GCN", data_root=tmp)
    dq.OBSERVE, dq.epoch = n_envs, n_envs * 3
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    env = VecExplorationEnv(bench.MAP, n_envs, env_index=0, test=True, device=0)
    dq.running(pol, tgt, test=True, env=env)
    for name, label in (("graph_matrix", "graph export"), ("actions_all_goals", "line plans"), ("rewards_all_goals", "look-ahead rewards"),
                        ("step", "env.step (plan execution)"), ("reset", "env.reset")):
        timed(env, name, label)
    timed(dq, "test", "policy forward (acting)")
    prof = None
    if len(sys.argv) > 3 and sys.argv[3] == "cprofile":  # host-side cost of the updates: no synchronising timer around them
        import cProfile, pstats
        prof = cProfile.Profile()
    else:
        timed(dq, "_train_minibatches", "train minibatches")
    dq.epoch = n_envs * iters
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    dq.running(pol, tgt, test=True, env=env)
    if prof:
        prof.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof:
        print("host returned after %.1f ms per vector step, device drained after %.1f ms" % ((t1 - t0) / iters * 1e3, dt / iters * 1e3))
        pstats.Stats(prof).sort_stats("tottime").print_stats(45)
    c = env.engine.counts_dev().cpu().numpy()
    print("poses at the end: mean %.1f max %d; landmarks mean %.1f" % (c[:, 0].mean(), c[:, 0].max(), c[:, 1].mean()))
    env.close()
print("%d envs, %d vector steps: %.1f ms per vector step (synchronised)" % (n_envs, iters, dt / iters * 1e3))
tot = 0.0
for k, v in acc.items():
    print("  %-28s %8.2f ms" % (k, v / iters * 1e3)); tot += v
print("  %-28s %8.2f ms" % ("host bookkeeping / other", (dt - tot) / iters * 1e3))

"""Dev tool (NOPROF=1: no profiler, for rocprofv3 --kernel-trace --stats around it - the device-busy time of the loop): cProfile over A2C.running (bench.py's a2c_loop workload): where the host side of a vector step goes.  (NOGC=1: with the
cyclic garbage collector off - under the profiler the allocation bursts of a step trigger full collections that a plain run does not
show: a gc-quiet wrapper around the loops measured 12.2-12.6 against 11.6 ms per vector step and was dropped.)"""
import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from drl_graph_exploration_amd.networks import PolicyGCN, ValueGCN
from drl_graph_exploration_amd.policy import A2C
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n_envs, iters = 256, 40
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
with tempfile.TemporaryDirectory() as tmp:
    a2c = A2C("b/", data_root=tmp)
    actor, critic = PolicyGCN().to(dev), ValueGCN().to(dev)
    env = VecExplorationEnv(bench.MAP, n_envs, env_index=0, test=True, device=0)
    a2c.epoch, a2c.nstep = n_envs * 2, 2
    a2c.running(actor, critic, test=True, env=env)
    a2c.nstep = 40
    if os.environ.get("PRIME"):  # a few large blocks into torch's caching allocator before the timed region (GB each)
        blocks = [torch.empty(int(float(os.environ["PRIME"]) * (1 << 30)), dtype=torch.uint8, device=dev) for _ in range(4)]
        del blocks
    a2c.graphs_per_pass = int(os.environ.get("GPP", a2c.graphs_per_pass))
    a2c.buffer.clear()
    a2c.epoch = n_envs * iters
    import gc
    upd = [0.0]
    _train = a2c.train
    def timed_train(*a, **k):  # the update's share of the wall clock (drains the stream before and after: SPLIT=1 only)
        torch.cuda.synchronize(); t = time.perf_counter()
        _train(*a, **k)
        torch.cuda.synchronize(); upd[0] += time.perf_counter() - t
    if os.environ.get("SPLIT"): a2c.train = timed_train
    if os.environ.get("NOUPD"): a2c.train = lambda *a, **k: None  # (stepping only)
    stamps, resets = [], []
    if os.environ.get("PERSTEP"):  # wall clock between consecutive graph exports = one vector step each; resets marked
        _gm, _rs = env.graph_matrix, env.reset
        def gm(*a, **k):
            stamps.append(time.perf_counter()); return _gm(*a, **k)
        def rs(ids=None):
            resets.append((len(stamps), 0 if ids is None else len(ids))); return _rs(ids)
        env.graph_matrix, env.reset = gm, rs
    if os.environ.get("NOGC"): gc.disable()
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if not os.environ.get("NOPROF"): pr.enable()
    a2c.running(actor, critic, test=True, env=env)
    torch.cuda.synchronize()
    if not os.environ.get("NOPROF"): pr.disable()
    dt = time.perf_counter() - t0
    env.close()
print("%.2f ms per vector step" % (dt / iters * 1e3))
if os.environ.get("SPLIT"): print("  of which the update: %.2f ms per vector step (%.1f ms per update); stepping %.2f ms" % (upd[0] / iters * 1e3, upd[0] * 1e3, (dt - upd[0]) / iters * 1e3))
if not os.environ.get("NOPROF"):
    st = pstats.Stats(pr); st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "28")))

if os.environ.get("PERSTEP"):
    d = np.diff(np.array(stamps)) * 1e3
    marks = dict(resets)
    print("per-step ms (r = a reset of n envs inside): " + " ".join("%.2f%s" % (v, ("r%d" % marks[i + 1]) if (i + 1) in marks else "") for i, v in enumerate(d)))

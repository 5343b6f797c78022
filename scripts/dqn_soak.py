"""Dev tool: a longer DeepQ.running run (episodes to completion, resets, pool recycling, target refreshes) - prints the
episode statistics; any engine status / capacity error raises."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.policy import DeepQ
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    dq = DeepQ("soak/", "GCN", data_root=tmp)
    dq.OBSERVE, dq.epoch, dq.TARGET_UPDATE, dq.REPLAY_MEMORY = n_envs * 2, n_envs * steps, n_envs * 40, 2000
    dq.updates_per_vector_step = 8
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    env = VecExplorationEnv(40, n_envs, env_index=0, test=False, device=0, seed=1)
    resets = [0]
    orig = env.reset
    def counted(ids=None):
        if ids is not None:
            resets[0] += len(ids)
        return orig(ids)
    env.reset = counted
    t0 = time.time()
    dq.running(pol, tgt, test=False, env=env)
    torch.cuda.synchronize()
    dt = time.time() - t0
    c = env.engine.counts_dev().cpu().numpy()
    print("%d envs x %d vector steps in %.1f s (%.0f RL it/s); episodes finished / truncated: %d; poses now mean %.0f max %d; "
          "explored mean %.2f; loss %.4g; buffer %d; pool slots in use %d/%d" % (
              n_envs, steps, dt, n_envs * steps / dt, resets[0], c[:, 0].mean(), c[:, 0].max(), float(env.status().mean()),
              dq.temp_loss, len(dq.buffer), sum(1 for r in dq._pool.ref if r > 0), dq._pool.n_slots))
    env.close()

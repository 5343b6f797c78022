"""Dev tool: a longer A2C.running run (episodes to completion, resets, several updates, the n-step window's pool recycling) with the
process's memory and the cyclic collector's state printed at the end; any engine status / capacity error raises.
python scripts/a2c_soak.py [n_envs] [vector steps]"""
import gc, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd.networks import PolicyGCN, ValueGCN
from drl_graph_exploration_amd.policy import A2C
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
with tempfile.TemporaryDirectory() as tmp:
    a2c = A2C("soak/", data_root=tmp)
    actor, critic = PolicyGCN().to(dev), ValueGCN().to(dev)
    env = VecExplorationEnv(40, n_envs, env_index=0, test=False, device=0, seed=1)
    resets = [0]
    orig = env.reset
    def counted(ids=None):
        if ids is not None:
            resets[0] += len(ids)
        return orig(ids)
    env.reset = counted
    a2c.epoch, a2c.nstep = n_envs * steps, 40
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.time()
    a2c.running(actor, critic, test=False, env=env)
    torch.cuda.synchronize()
    dt = time.time() - t0
    c = env.engine.counts_dev().cpu().numpy()
    print("%d envs x %d vector steps in %.1f s (%.2f ms per vector step): %d episode resets, loss %.4g, entropy %.4g; poses now mean %.0f max %d" % (
        n_envs, steps, dt, dt / steps * 1e3, resets[0], a2c.temp_loss, a2c.entro, c[:, 0].mean(), c[:, 0].max()))
    print("max RSS %.0f -> %.0f MB; gc enabled again: %s; device memory %.0f MB allocated / %.0f MB reserved" % (
        rss0 / 1024, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, gc.isenabled(), torch.cuda.memory_allocated() / 2**20,
        torch.cuda.memory_reserved() / 2**20))
    assert np.isfinite(a2c.temp_loss) and gc.isenabled()
    env.close()

"""Dev tool: DeepQ.running twice from the same seeds - updates as two host calls over one arena with the replay pool's CSR cache
(the default) against the same launches issued one by one from Python with the graph data rebuilt per mini-batch - the policy
parameters after the run must be bit-equal.   ab_dqn_update_paths.py [envs = 32] [vector steps = 40]"""
import os, sys, tempfile, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd.networks import GCN
from drl_graph_exploration_amd.policy import DeepQ
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
out = []
for fused in (True, False):
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        dq = DeepQ("ab/", "GCN", data_root=tmp)
        dq.fused_update = fused
        dq.OBSERVE, dq.epoch, dq.TARGET_UPDATE, dq.REPLAY_MEMORY = n_envs * 2, n_envs * steps, n_envs * 15, 1500
        dq.updates_per_vector_step = 6
        pol, tgt = GCN().to(dev), GCN().to(dev)
        tgt.load_state_dict(pol.state_dict())
        env = VecExplorationEnv(40, n_envs, env_index=0, test=True, device=0)
        dq.running(pol, tgt, test=True, env=env)
        torch.cuda.synchronize()
        out.append(([v.clone() for v in pol.state_dict().values()], dq.temp_loss, len(dq.buffer), dq.step_t))
        env.close()
same = all(torch.equal(a, b) for a, b in zip(out[0][0], out[1][0]))
print("%d envs x %d vector steps, %d updates: parameters bit-equal: %s; loss %.17g vs %.17g; buffer %d / %d" % (
    n_envs, steps, 6 * (steps - 2), same, out[0][1], out[1][1], out[0][2], out[1][2]))
sys.exit(0 if same and out[0][1] == out[1][1] else 1)

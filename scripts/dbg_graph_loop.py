"""Dev tool: the decision loop of tests/test_gpu_graph.py with, per decision, the explored fractions of engine and oracle,
the number of knife-edge cells (oracle.knife_edge_cells) and the largest estimate difference."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
MAP = 40
n = 6
num_lm = None
cfg = default_config(MAP, num_landmarks=num_lm)
eng = Engine(cfg, n, 64)
starts = np.array([O.start_pose(lo, MAP / 2 + 20) for lo in range(n)]) + np.array([0.3183, -0.2718, 0.1234])
envs = [O.OracleEnv(MAP, lo, num_landmarks=num_lm, start=tuple(starts[lo])) for lo in range(n)]
seeds = np.arange(n); todo = np.arange(n)
while len(todo):
    eng.reset(todo, seeds[todo], starts=starts[todo])
    active = torch.zeros(n, dtype=torch.uint8, device=eng.device); active[torch.as_tensor(todo)] = 1
    for _ in range(4):
        eng.step(torch.tensor([(1, 1, math.pi / 2)] * n, dtype=torch.float64, device=eng.device), active)
    todo = np.array([i for i in todo if eng.counts(int(i))["landmarks"] < 1], dtype=np.int64)
    seeds[todo] += 50
for decision in range(5):
    g = eng.graph()
    nfr = g["n_frontier"].cpu().numpy(); fxy = g["frontier_xy"]
    chosen = []
    for i, env in enumerate(envs):
        A, X, _, fro = env.graph_matrix()
        all_actions = env.actions_all_goals()
        ks = A.shape[0] - fro
        chosen.append(all_actions[ks + (decision % fro)])
    maxlen = max(len(a) for a in chosen)
    for k in range(maxlen):
        odom = torch.zeros(n, 3, dtype=torch.float64, device=eng.device)
        active = torch.zeros(n, dtype=torch.uint8, device=eng.device)
        for i in range(n):
            if k < len(chosen[i]):
                odom[i] = torch.tensor(chosen[i][k], dtype=torch.float64); active[i] = 1
                envs[i].step(chosen[i][k])
        eng.step(odom, active)
    ex = eng.explored().cpu().numpy()
    for i, env in enumerate(envs):
        xyt, _ = eng.poses(i); oxyt, _ = env._sim.poses()
        knife = env._sim.knife_edge_cells(1e-9)
        pe = eng.virtual_map(i)[0].reshape(-1); po = env._sim.virtual_map()[0].reshape(-1)
        print("decision %d env %d: explored %.4f / %.4f, poses %d, landmarks %d, knife-edge cells %d, max |pose diff| %.2e, differing cells %d (of them knife-edge %d)" % (
            decision, i, ex[i], env.status(), len(xyt), eng.counts(i)["landmarks"], int(knife.sum()), np.abs(xyt - oxyt).max(),
            int((pe != po).sum()), int(((pe != po) & knife).sum())))

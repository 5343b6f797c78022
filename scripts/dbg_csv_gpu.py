"""Dev tool: replay the reference's evaluation CSV pins (tests/golden/csv_pin.json) through the HIP engine + HIP GCN."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
from drl_graph_exploration_amd.networks import GCN, GraphData

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pins = json.load(open(os.path.join(root, "tests/golden/csv_pin.json")))["seeds"]
seeds = [int(k) for k, v in pins.items() if len(v["rows"]) >= 4]
n = len(seeds)
env = VecExplorationEnv(40, n, env_index=0, test=True, max_poses=41)
env.env_index = np.array(seeds, dtype=np.int64)
env.reset()
print("env_index after reset", list(env.env_index))
dev = env.device
model = GCN()
model.load_state_dict(torch.load(os.path.join(root, "tests/golden/DQN_GCN_MyModel.pt"), map_location="cpu"))
model.to(dev)


def entropy(obs):
    return float(-(obs * np.log(obs)).sum() + 0.5 * np.log(0.5) * 1200)


row = [0] * n
ok = [True] * n
worst = np.zeros(3)
ndec = max(len(pins[str(s)]["choices"]) for s in seeds)
for d in range(ndec):
    g = env.graph_matrix()
    with torch.no_grad():
        q = model(GraphData(g["x"], g["edge_index"], g["edge_attr"], g["batch"]), 0.0, batch=g["batch"]).view(-1).cpu().numpy()
    node_off = g["node_off"].cpu().numpy()
    nfr = g["n_frontier"].cpu().numpy()
    acts, nact = env.actions_all_goals()
    cand_env, cand_node, first = env.candidates
    first = first.cpu().numpy()
    choice = np.zeros(n, dtype=np.int64)
    live = np.zeros(n, dtype=bool)
    for i, s in enumerate(seeds):
        pin = pins[str(s)]
        if not ok[i] or d >= len(pin["choices"]):
            continue
        qi = q[node_off[i + 1] - nfr[i]:node_off[i + 1]]
        if pin["choices"][d] >= nfr[i]:
            print("seed", s, "decision", d, "frontier count", nfr[i], "< choice", pin["choices"][d]); ok[i] = False; continue
        if int(np.argmax(qi)) != pin["gcn_choices"][d]:
            print("seed", s, "decision", d, "gcn argmax", int(np.argmax(qi)), "pin", pin["gcn_choices"][d], qi)
        choice[i] = pin["choices"][d]
        live[i] = True
    c = torch.as_tensor(first + choice, device=dev)
    a = acts[c]
    na = nact[c] * torch.as_tensor(live, device=dev).to(nact.dtype)
    for k in range(int(na.max().item())):
        active = (na > k).to(torch.uint8)
        env.engine.step(a[:, k].contiguous(), active)
        act_h = active.cpu().numpy()
        for i, s in enumerate(seeds):
            if not act_h[i] or not ok[i]:
                continue
            ref = np.array(pins[str(s)]["rows"][row[i]])
            got = np.array([env.get_landmark_error(i), entropy(env.obs(i)), env.max_uncertainty_of_trajectory(i)])
            rel = np.abs(got - ref) / np.abs(ref)
            if rel[0] > 1e-4 or rel[2] > 1e-4 or rel[1] > 5e-3:
                print("seed", s, "row", row[i], "got", got, "ref", ref, "rel", rel); ok[i] = False
            else:
                worst = np.maximum(worst, rel)
            row[i] += 1
    env._graph = None
done = [row[i] == len(pins[str(s)]["rows"]) for i, s in enumerate(seeds)]
print("seeds", n, "fully matched", sum(d and o for d, o in zip(done, ok)), "rows matched", sum(row), "of", sum(len(pins[str(s)]["rows"]) for s in seeds))
print("worst rel (landmark err, entropy, max trace)", worst)
print("not ok:", [s for i, s in enumerate(seeds) if not ok[i] or not done[i]])

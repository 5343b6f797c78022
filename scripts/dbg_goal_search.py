"""Dev tool: at the first decision where no frontier candidate of the oracle matches the reference CSV, try every
interior cell centre as the goal of the line plan."""
import sys, os, json
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O
from replay_csv_search import clone_env
lo = int(sys.argv[1])
pin = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "/tmp/csv_pin_full.json"))["seeds"][str(lo)]
ref = pd.read_csv("/root/reference/data/test_result/40_DQN_GCN.csv")
ref = ref[ref["Step"].notna()].reset_index(drop=True)
starts = np.nonzero(ref["Step"].values == 1.0)[0]
seg = ref.iloc[starts[lo]:starts[lo] + 400][["Landmarks error", "Map entropy", "Max localization uncertainty"]].values
env = O.OracleEnv(40, lo); st = 0
for ch in pin["choices"]:
    A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
    for a in acts[ks + ch]:
        obs, _, _ = env.step(a); st += 1
        print(st, O.map_entropy(obs), seg[st - 1][1], O.map_entropy(obs) - seg[st - 1][1])
print("tracked", st, "pose", env.vehicle_position(), "frontiers", env.frontier() if False else "")
A, X, _, fro = env.graph_matrix()
print("frontier goals", env._frontier)
rows_, cols_ = env._sim.vm_shape()
res = env.cfg.resolution
found = []
for r in range(rows_):
    for c in range(cols_):
        gx = env.cfg.map_min_x + (c + 0.5) * res; gy = env.cfg.map_min_y + (r + 0.5) * res
        if abs(gx) > 20 or abs(gy) > 20: continue
        acts = env._sim.line_plan((gx, gy))
        e2 = clone_env(env); rows = []
        for a in acts:
            obs, d2, _ = e2.step(a)
            rows.append((e2.get_landmark_error(), O.map_entropy(obs), e2.max_uncertainty_of_trajectory()))
        rows = np.array(rows); rr = seg[st:st + len(rows)]
        d = np.abs(rows - rr) / np.abs(rr); err = max(d[:, 0].max(), d[:, 2].max())
        if err < 1e-3: found.append((err, gx, gy, len(acts)))
found.sort()
print(found[:10])

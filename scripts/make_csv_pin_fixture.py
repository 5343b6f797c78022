"""Generates tests/golden/csv_pin.json from the reference's shipped evaluation CSV
(/root/reference/data/test_result/40_DQN_GCN.csv, written by scripts/test.py:136-142) — run in the
build container only.  For each seed it stores the first rows of (Landmarks error, Map entropy,
Max localization uncertainty) and, per decision, the index of the frontier candidate the reference
run took (inferred by replaying every candidate on the CPU oracle and keeping the one whose rows
match the CSV) together with the candidate the GCN restatement picks with the shipped
DQN_GCN/MyModel.pt.  Data only; no reference source is copied."""
import json, os, sys
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import oracle as O, gcn_ref
from replay_csv_search import clone_env

ref = pd.read_csv("/root/reference/data/test_result/40_DQN_GCN.csv")
ref = ref[ref["Step"].notna()].reset_index(drop=True)
starts = np.nonzero(ref["Step"].values == 1.0)[0]
params = torch.load("/root/reference/data/torch_weights/DQN_GCN/MyModel.pt", map_location="cpu")
out = {"source": "data/test_result/40_DQN_GCN.csv", "map_size": 40, "seeds": {}}
MAX_STEPS = 24
for lo in range(50):
    seg = ref.iloc[starts[lo]:starts[lo] + 400][["Landmarks error", "Map entropy", "Max localization uncertainty"]].values
    env = O.OracleEnv(40, lo); st = 0; choices = []; gcn_choices = []
    while st < MAX_STEPS:
        A, X, _, fro = env.graph_matrix(); ei, ea, x = O.data_process(A, X)
        acts = env.actions_all_goals(); ks = A.shape[0] - fro
        with torch.no_grad():
            q = gcn_ref.gcn_forward(params, torch.tensor(x), torch.tensor(ei), torch.tensor(ea)).view(-1).numpy()
        best = None
        for i in range(fro):
            e2 = clone_env(env); rows = []
            for a in acts[ks + i]:
                obs, d2, _ = e2.step(a)
                rows.append((e2.get_landmark_error(), O.map_entropy(obs), e2.max_uncertainty_of_trajectory()))
            rows = np.array(rows); r = seg[st:st + len(rows)]
            d = np.abs(rows - r) / np.abs(r); err = max(d[:, 0].max(), d[:, 2].max())
            if best is None or err < best[0]: best = (err, i, e2, len(rows))
        if best[0] > 1e-4: break
        choices.append(best[1]); gcn_choices.append(int(np.argmax(q[-fro:]))); env = best[2]; st += best[3]
    out["seeds"][str(lo)] = {"rows": seg[:st].tolist(), "choices": choices, "gcn_choices": gcn_choices}
    print(lo, st, choices, gcn_choices)
json.dump(out, open(os.path.join(ROOT, "tests/golden/csv_pin.json"), "w"))

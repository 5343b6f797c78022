"""Replay scripts/test.py (DQN+GCN, map 40) on the CPU oracle and compare with the reference's
shipped per-step CSV (data/test_result/40_DQN_GCN.csv).  Dev tool; the committed test uses fixtures."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from oracle import gcn_ref

def replay(lo, params, max_steps=400, verbose=False):
    env = O.OracleEnv(40, lo)
    rows = []
    done = False
    step_t = 0
    while not done and step_t < max_steps:
        A, X, _, fro = env.graph_matrix()
        ei, ea, x = O.data_process(A, X)
        all_actions = env.actions_all_goals()
        with torch.no_grad():
            q = gcn_ref.gcn_forward(params, torch.tensor(x), torch.tensor(ei), torch.tensor(ea)).view(-1).numpy()
        key_size = A.shape[0] - fro
        ai = int(np.argmax(q[-fro:]))
        for act in all_actions[key_size + ai]:
            obs, done, _ = env.step(act)
            step_t += 1
            rows.append((env.get_landmark_error(), O.map_entropy(obs), env.max_uncertainty_of_trajectory()))
            if done: break
    return np.array(rows)

if __name__ == "__main__":
    import pandas as pd
    ref = pd.read_csv("/root/reference/data/test_result/40_DQN_GCN.csv")
    ref = ref[ref["Step"].notna()].reset_index(drop=True)
    steps = ref["Step"].values
    starts = np.nonzero(steps == 1.0)[0]
    params = torch.load("/root/reference/data/torch_weights/DQN_GCN/MyModel.pt", map_location="cpu")
    los = [int(a) for a in sys.argv[1:]] or [0]
    for lo in los:
        seg = ref.iloc[starts[lo]:starts[lo]+400][["Landmarks error","Map entropy","Max localization uncertainty"]].values
        t0=time.time()
        rows = replay(lo, params)
        n = min(len(rows), 400)
        print("lo", lo, "steps", len(rows), "time %.1fs"%(time.time()-t0))
        for k in range(min(n, 12)):
            print(k+1, rows[k], seg[k])
        d = np.abs(rows[:n]-seg[:n])/np.maximum(np.abs(seg[:n]),1e-9)
        ok = np.all(d < 5e-3, axis=1)
        first_bad = int(np.argmin(ok)) if not ok.all() else n
        print("  first step with rel diff > 5e-3:", first_bad+1, "of", n, " max rel diff before:", d[:first_bad].max(axis=0) if first_bad else None)

"""Dev tool: follow the CSV pin choices of one seed on the oracle, then print candidate rows and iSAM deltas at the
first decision where no candidate matches the reference CSV."""
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
        env.step(a); st += 1
print("tracked", st, "steps; isam count", env._sim.isam_state()[4], "P", env._sim.num_poses(), "L", env._sim.num_landmarks())
A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
for i in range(fro):
    e2 = clone_env(env)
    print("candidate", i, "n_actions", len(acts[ks + i]))
    for k, a in enumerate(acts[ks + i]):
        thp, dp, thl, dl, cnt = e2._sim.isam_state()
        e2.step(a)
        thp2, dp2, thl2, dl2, cnt2 = e2._sim.isam_state()
        row = (e2.get_landmark_error(), O.map_entropy(e2._obs), e2.max_uncertainty_of_trajectory())
        print("  step", st + k + 1, "count", cnt2, "max|d_pose| before %.4f after %.4f" % (np.abs(dp).max(), np.abs(dp2).max()),
              "max|d_lm| before %.4f" % (np.abs(dl).max() if len(dl) else 0), "P", len(thp2), "L", len(thl2),
              "rows", np.array(row), "ref", seg[st + k])

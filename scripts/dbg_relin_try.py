"""Dev tool: follow a seed's pinned choices up to the last tracked step, then run every candidate with different
relinearisation thresholds and print the rows against the reference CSV."""
import sys, os, json, ctypes as C
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O
from replay_csv_search import clone_env
lo = int(sys.argv[1])
pin = json.load(open("/tmp/csv_pin_full.json"))["seeds"][str(lo)]
ref = pd.read_csv("/root/reference/data/test_result/40_DQN_GCN.csv")
ref = ref[ref["Step"].notna()].reset_index(drop=True)
starts = np.nonzero(ref["Step"].values == 1.0)[0]
seg = ref.iloc[starts[lo]:starts[lo] + 400][["Landmarks error", "Map entropy", "Max localization uncertainty"]].values
L = O.lib(); L.orc_dev_set_relin.argtypes = [C.c_double, C.c_int, C.c_int]
env = O.OracleEnv(40, lo); st = 0
for ch in pin["choices"]:
    A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
    for a in acts[ks + ch]:
        env.step(a); st += 1
A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
thp, dp, thl, dl, cnt = env._sim.isam_state()
print("count", cnt, "d_pose maxima per pose", np.round(np.abs(dp).max(axis=1), 3), "d_lm", np.round(np.abs(dl).max(axis=1), 3))
for thr, mode in [(0.1, 0), (0.09, 0), (0.05, 0), (0.01, 0), (0.0, 0), (0.09, 1), (0.05, 1)]:
    L.orc_dev_set_relin(thr, 10, mode)
    for i in range(min(fro, 1)):
        e2 = clone_env(env)
        for k, a in enumerate(acts[ks + i][:3]):
            e2.step(a)
            row = (e2.get_landmark_error(), O.map_entropy(e2._obs), e2.max_uncertainty_of_trajectory())
            print("thr", thr, "mode", mode, "cand", i, "step", st + k + 1, np.array(row), "ref", seg[st + k])
L.orc_dev_set_relin(0.1, 10, 0)

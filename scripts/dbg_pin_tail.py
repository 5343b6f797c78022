"""Dev tool: replay a seed's pinned choices (fixture JSON given) and print the oracle's candidate rows after the last
tracked step next to the reference CSV rows."""
import sys, os, json
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O
from replay_csv_search import clone_env
import make_csv_pin_fixture as MK
lo = int(sys.argv[1])
seg, best, fin, used = MK.search(lo)
print("tracked", best["st"], "choices", best["choices"])
env, st = best["envs"][-1]
plan = MK.plan_of(env, best["choices"][-1])
r = MK.rollout(env, plan, seg, st); env = r[1]; st += r[2]
A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
print("pose", env.vehicle_position(), "frontiers", [tuple(map(float, f)) for f in env._frontier])
thp, dp, thl, dl, cnt = env._sim.isam_state()
print("count", cnt, "max|d|", np.abs(dp).max(), "L", len(thl))
for i in range(fro):
    e2 = clone_env(env)
    print("cand", i, [tuple(np.round(a, 4)) for a in acts[ks + i]])
    for k, a in enumerate(acts[ks + i]):
        obs, _, _ = e2.step(a)
        print("   ", st + k + 1, e2.get_landmark_error(), O.map_entropy(obs), e2.max_uncertainty_of_trajectory(), "| ref", seg[st + k])

"""Dev tool: for each seed of a csv_pin fixture, where does the oracle's map first differ from the reference's
(entropy column) relative to where the (landmark error, max uncertainty) columns stop matching."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
pins = json.load(open(sys.argv[1]))["seeds"]
for lo in range(50):
    pin = pins[str(lo)]; rows = np.array(pin["rows"])
    if not len(rows): print(lo, "none"); continue
    env = O.OracleEnv(40, lo); st = 0; first = None
    for ch in pin["choices"]:
        A, X, _, fro = env.graph_matrix(); acts = env.actions_all_goals(); ks = A.shape[0] - fro
        for a in acts[ks + ch]:
            obs, _, _ = env.step(a); st += 1
            if first is None and abs(O.map_entropy(obs) - rows[st - 1][1]) > 1e-6: first = st
    thp, dp, thl, dl, cnt = env._sim.isam_state()
    print(lo, "tracked", st, "of", pin["episode_rows"], "first map difference at", first, "| isam count", cnt,
          "max|d| %.3f" % max(np.abs(dp).max(), np.abs(dl).max() if len(dl) else 0), "L", len(thl))

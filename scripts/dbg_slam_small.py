"""Debug helper: one env, a few steps; print pose covariance traces / info of engine vs oracle."""
import sys, os, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
cfg = default_config(40)
eng = Engine(cfg, 2, 0)
start = np.array([O.start_pose(lo, 40.0) for lo in range(2)]) + np.array([0.3183, -0.2718, 0.1234])
eng.reset([0, 1], [0, 1], starts=start)
sims = [O.OracleSim(O.default_config(40), lo, lo, start=tuple(start[lo])) for lo in range(2)]
for s in range(8):
    od = (1.0, 1.0, math.pi / 2) if s < 4 else (2.0, 0.0, 0.0)
    eng.step(torch.tensor([od] * 2, dtype=torch.float64, device=eng.device))
    for sm in sims: sm.simulate(od)
    xyt, info = eng.poses(0)
    oxyt, oinfo = sims[0].poses()
    lt, pt = eng.cov_traces(0)
    olt, opt = sims[0].cov_traces()
    print("step", s, "P", len(xyt), "status", eng.status(), "max|dxyt|", np.abs(xyt - oxyt).max(), "pose traces eng", np.round(pt, 6), "orc", np.round(opt, 6))

"""Dev tool: belief-step time of the bench workload (256 envs, ~37 poses) as a function of the engine's pose capacity
(which k_slam variant runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine

for mp in [int(a) for a in sys.argv[1:]] or [41, 58, 86, 127]:
    cfg = default_config(bench.MAP, num_landmarks=bench.NUM_LM, max_poses=mp, max_landmarks=100, max_factors=12 * mp, max_snapshots=1)
    eng = Engine(cfg, bench.N_ENVS, 0, 0)
    eng.reset(np.arange(bench.N_ENVS), np.arange(bench.N_ENVS), los=np.arange(bench.N_ENVS))
    for a in bench.WARM_SCRIPT if hasattr(bench, "WARM_SCRIPT") else []:
        eng.step(torch.tensor([a] * bench.N_ENVS, dtype=torch.float64, device=eng.device))
    eng.synchronize()
    eng.snapshot(0)
    odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
    eng.timing_enable(2)
    for it in range(30):
        if it == 10:
            eng.timing_read()
        eng.restore(0)
        eng.step(odom)
    tm = eng.timing_read()
    c = eng.counts(0)
    print("max_poses %3d (poses now %d): " % (mp, c["poses"]) + ", ".join("%s %.1f us" % (k, v[0] / v[1] * 1e3) for k, v in tm.items() if v[1]))
    eng.close()

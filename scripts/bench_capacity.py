"""Dev tool: belief-step time of the bench workload (256 envs, ~37 poses) as a function of the engine's pose capacity:
the fused step kernel and the snapshot restore by HIP events (minus the empty-span overhead), then - argument `staged` - the
three stage kernels.  usage: bench_capacity.py [staged] [max_poses ...]   (default 41 256)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine

STAGED = 'staged' in sys.argv[1:]
for mp in [int(a) for a in sys.argv[1:] if a != 'staged'] or [41, 256]:
    cfg = default_config(bench.MAP, num_landmarks=bench.NUM_LM, max_poses=mp, max_landmarks=100, max_factors=12 * mp, max_snapshots=1)
    eng = Engine(cfg, bench.N_ENVS, 0, 0)
    eng.reset(np.arange(bench.N_ENVS), np.arange(bench.N_ENVS), los=np.arange(bench.N_ENVS))
    for a in bench.WARM_SCRIPT if hasattr(bench, "WARM_SCRIPT") else []:
        eng.step(torch.tensor([a] * bench.N_ENVS, dtype=torch.float64, device=eng.device))
    eng.synchronize()
    eng.snapshot(0)
    odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
    eng.timing_enable(2 if STAGED else 1)
    for it in range(30):
        if it == 10:
            eng.timing_read()
        eng.restore(0)
        eng.step(odom)
    tm = eng.timing_read()
    c = eng.counts(0)
    ov = tm["t7"][0] / max(tm["t7"][1], 1) * 1e3
    print("max_poses %3d (poses now %d, incremental/full %s): " % (mp, c["poses"], eng.inc_stats()) +
          ", ".join("%s %.1f us" % (k, v[0] / v[1] * 1e3 - ov) for k, v in tm.items() if v[1] and k != "t7") + " (event overhead %.1f us subtracted)" % ov)
    eng.close()

"""Dev tool: BASELINE config 5 scale (50 m map, 500 landmarks, ~110-pose graphs), update by update: which SLAM path served it
(incremental rank-k update / full solve), how many factors and first sightings the step added, and the stage kernels' times."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
n = 256
cfg = default_config(50, num_landmarks=500, max_poses=127, max_landmarks=127, max_factors=3800)
eng = Engine(cfg, n, 0)
rng = np.random.RandomState(0)
starts = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-3, 3, n)], 1)
eng.reset(np.arange(n), np.arange(n), starts=starts)
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
for s in range(108):
    eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
eng.timing_enable(2)
for s in range(12):
    eng.inc_stats(True); eng.timing_read()
    c0 = eng.counts_dev().cpu().numpy()
    eng.step(torch.tensor([loop[(108 + s) % len(loop)]] * n, dtype=torch.float64, device=eng.device))
    eng.synchronize()
    c1 = eng.counts_dev().cpu().numpy()
    tm = eng.timing_read()
    print("update #%d: inc/full %s, new factors mean %.1f max %d, new lms max %d, slam %.1f us map %.1f sim %.1f" % (c1[0, 4], eng.inc_stats(), (c1[:, 2] - c0[:, 2]).mean(), (c1[:, 2] - c0[:, 2]).max(), (c1[:, 1] - c0[:, 1]).max(), tm["slam"][0] * 1e3, tm["map"][0] * 1e3, tm["sim"][0] * 1e3))

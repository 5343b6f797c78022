"""Dev tool: the same belief step twice from one snapshot must give bit-identical state (every k_slam variant)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine

n = 64
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
for cap, poses, msize, nlm in [(41, 37, 40, 100), (58, 50, 40, 100), (85, 75, 40, 100), (127, 105, 50, 500)]:
    cfg = default_config(msize, num_landmarks=nlm, max_poses=cap, max_landmarks=128, max_factors=30 * cap, max_snapshots=1)
    eng = Engine(cfg, n, 0, 0)
    rng = np.random.RandomState(1)
    starts = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-3, 3, n)], 1)
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    od = [torch.tensor([a] * n, dtype=torch.float64, device=eng.device) for a in loop]
    for s in range(poses - 1):
        eng.step(od[s % len(loop)])
    assert eng.status() == 0
    eng.snapshot(0)
    outs = []
    for rep in range(3):
        eng.restore(0)
        eng.step(od[(poses - 1) % len(loop)])
        assert eng.status() == 0
        st = []
        for i in range(0, n, 7):
            xyt, info = eng.poses(i)
            keys, xy, linfo = eng.landmarks(i)
            prob, vinfo, tr, upd = eng.virtual_map(i)
            st += [xyt, info, xy, linfo, prob, vinfo, tr]
        outs.append(np.concatenate([a.reshape(-1) for a in st]))
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    print("capacity %3d, %3d poses: %s (%d values)" % (cap, poses + 1, "bit-identical" if same else "DIFFERENT max |d| %.3g" % max(np.abs(outs[0] - o).max() for o in outs[1:]), outs[0].size))
    eng.close()

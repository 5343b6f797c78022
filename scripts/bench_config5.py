"""Dev tool: belief-step time at BASELINE config 5 scale (50 m map, 500 landmarks, ~110-pose graphs; per-stage kernels)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 108
cfg = default_config(50, num_landmarks=500, max_poses=127, max_landmarks=127, max_factors=3800)
eng = Engine(cfg, n, 0)
rng = np.random.RandomState(0)
starts = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-3, 3, n)], 1)
eng.reset(np.arange(n), np.arange(n), starts=starts)
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
for s in range(warm):
    eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
assert eng.status() == 0, eng.status()
c = eng.counts_dev().cpu().numpy()
print("poses %.1f landmarks %.1f factors %.1f" % (c[:, 0].mean(), c[:, 1].mean(), c[:, 2].mean()))
eng.timing_enable(2)
t0 = time.time()
K = 8
for s in range(K):
    eng.step(torch.tensor([loop[(warm + s) % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
dt = (time.time() - t0) / K
print("n_envs %d: %.3f ms per step, %.0f env-steps/s" % (n, dt * 1e3, n / dt))
print({k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in eng.timing_read().items() if v[1]})
# in-kernel phase stamps of block 0 of the SLAM kernel (k_slam_arrow)
import ctypes as C
out = (C.c_int64 * 64)()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 1, None)
eng.step(torch.tensor([loop[(warm + K) % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.L.drlgx_debug_phase_clocks_host(eng.h, 0, out)
order = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]
names = ["tables+linearise", "blocks", "leaf factor", "leaf rhs down", "separator CR", "leaf up + selinv", "landmark system", "sweep", "lm out", "pose out"]
a = np.array(out[:11], dtype=np.float64)[order]
c = eng.counts(0)
print("block 0 (%d poses, %d landmarks, %d factors), us: " % (c["poses"], c["landmarks"], c["factors"]) +
      ", ".join("%s %.1f" % (names[k], (a[k + 1] - a[k]) / 100.0) for k in range(10)))

"""Dev tool: phases of the RELINEARISING update of the pose-chain solver (k_slam_arrow) at BASELINE config 5 scale, incl. the pieces of
its pose-output phase: phase_profile_config5_relin.py [workgroup = 0]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
n = 256
BLK = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = default_config(50, num_landmarks=500, max_poses=127, max_landmarks=127, max_factors=3800)
eng = Engine(cfg, n, 0)
rng = np.random.RandomState(0)
starts = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-3, 3, n)], 1)
eng.reset(np.arange(n), np.arange(n), starts=starts)
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
s = 0
while (eng.counts(0)["isam_count"] + 1) % 10 != 0 or eng.counts(0)["poses"] < 108:
    eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device)); s += 1
eng.synchronize()
out = (C.c_int64 * 1024)()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 1 | (BLK << 8), None)
eng.timing_enable(2); eng.timing_read(); eng.inc_stats(True)
eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
tm = eng.timing_read()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 5 | (BLK << 8), C.cast(out, C.POINTER(C.c_int64)))
a = np.array(out[:], dtype=np.float64)
c = eng.counts(BLK)
order = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]
names = ["tables+linearise", "blocks", "leaf factor", "leaf rhs down", "separator CR", "leaf up + selinv", "landmark system", "sweep", "lm out", "pose out"]
t = a[order]
print("update #%d, incremental / full %s, slam stage %.1f us; workgroup %d (%d poses, %d landmarks, %d factors):" % (c["isam_count"], eng.inc_stats(), tm["slam"][0] * 1e3, BLK, c["poses"], c["landmarks"], c["factors"]))
print("  " + ", ".join("%s %.1f" % (names[k], (t[k + 1] - t[k]) / 100.0) for k in range(10)))
us = lambda i, j: (a[i] - a[j]) / 100.0
print("  pose out: mirror of -C^-1 %.1f, Z = X_B [-C^-1 | delta_l] tiles + epilogues %.1f, estimates / marginals %.1f, panel: Sigma[., pn] %.1f, Sigma_ll + Sigma[l][pn] %.1f" % (
    us(100, 8), us(101, 100), us(9, 101), us(103, 102), us(104, 103)))
print("  wide Z product, wave 0 of the workgroup (us): staging %.1f, K loops %.1f, epilogues %.1f, row sums + stores %.1f" % (a[105] / 100.0, a[106] / 100.0, a[107] / 100.0, a[108] / 100.0))

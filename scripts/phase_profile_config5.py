"""Dev tool: in-kernel phases of the incremental SLAM stage with the covariance panel in HBM / L2 (k_inc.hip: inc_stream_batch) at
BASELINE config 5 scale (50 m map, 500 landmarks, ~110-pose graphs, stage kernels):
phase_profile_config5.py [workgroup = 0] [updates = 8]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
n = 256
BLK = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NUP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = default_config(50, num_landmarks=500, max_poses=127, max_landmarks=127, max_factors=3800)
eng = Engine(cfg, n, 0)
rng = np.random.RandomState(0)
starts = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-3, 3, n)], 1)
eng.reset(np.arange(n), np.arange(n), starts=starts)
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
for s in range(108):
    eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
out = (C.c_int64 * 64)()
ARM = 1 | (BLK << 8)
eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, None)
eng.timing_enable(2)
for s in range(NUP):
    eng.inc_stats(True); eng.timing_read()
    c0 = eng.counts_dev().cpu().numpy()
    eng.step(torch.tensor([loop[(108 + s) % len(loop)]] * n, dtype=torch.float64, device=eng.device))
    eng.synchronize()
    c1 = eng.counts_dev().cpu().numpy()
    tm = eng.timing_read()
    eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, out)
    a = np.array(out[:], dtype=np.float64)
    us = lambda i, j: (a[i] - a[j]) / 100.0
    print("update #%d wg %d (P %d L %d, +%d factors, +%d lms; launch mean +%.1f factors): slam %.1f us | loads %.1f, new pose %.1f, lists+lin %.1f, "
          "B %.1f (first batch: Ya %.1f, T+inverse %.1f, walk %.1f; a second batch: Ya + T + inverse %.1f, walk %.1f), C %.1f, D %.1f, meta %.1f; total %.1f" % (
              c1[0, 4], BLK, c1[BLK, 0], c1[BLK, 1], c1[BLK, 2] - c0[BLK, 2], c1[BLK, 1] - c0[BLK, 1], (c1[:, 2] - c0[:, 2]).mean(), tm["slam"][0] * 1e3,
              us(1, 0), us(2, 1), us(35, 2), us(3, 35), us(36, 35), us(37, 36), us(38, 37), us(39, 38) if a[39] > a[38] else 0.0, us(47, 39) if a[47] > a[39] else 0.0, us(4, 3), us(5, 4), us(7, 5), us(7, 0)))

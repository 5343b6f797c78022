"""Dev tool: start / end of every workgroup of the SLAM stage kernel (k_slam_arrow, incremental updates with the covariance panel in
HBM / L2) at BASELINE config 5 scale, against what each instance had to do (re-observed landmarks, landmarks, walks over the panel)."""
import ctypes as C, math, os, sys
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
for s in range(110):
    eng.step(torch.tensor([loop[s % len(loop)]] * n, dtype=torch.float64, device=eng.device))
eng.synchronize()
out = (C.c_int64 * 1024)()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 1, None)
eng.timing_enable(2)
for s in range(5):
    c0 = eng.counts_dev().cpu().numpy()
    eng.timing_read()
    eng.step(torch.tensor([loop[(110 + s) % len(loop)]] * n, dtype=torch.float64, device=eng.device))
    eng.synchronize()
    tm = eng.timing_read()
    c1 = eng.counts_dev().cpu().numpy()
    eng.L.drlgx_debug_phase_clocks_host(eng.h, 5, C.cast(out, C.POINTER(C.c_int64)))
    a = np.array(out[:], dtype=np.float64)[128:128 + 2 * n].reshape(-1, 2) / 100.0
    dur = a[:, 1] - a[:, 0]
    start = a[:, 0] - a[:, 0].min()
    nf = c1[:, 2] - c0[:, 2]; nl = c1[:, 1] - c0[:, 1]; nre = nf - nl; L0 = c0[:, 1]
    print("update #%d: slam %.1f us by events; workgroups: start max %.1f, duration min %.1f mean %.1f p90 %.1f max %.1f; first start -> last end %.1f" % (
        c1[0, 4], tm["slam"][0] * 1e3, start.max(), dur.min(), dur.mean(), np.percentile(dur, 90), dur.max(), a[:, 1].max() - a[:, 0].min()))
    for lo, hi in ((0, 16), (17, 24), (25, 32), (33, 64)):
        sel = (nre >= lo) & (nre <= hi)
        if sel.any():
            print("   re-observed %2d..%2d: %3d workgroups, duration mean %.1f max %.1f us (landmarks mean %.0f)" % (lo, hi, sel.sum(), dur[sel].mean(), dur[sel].max(), L0[sel].mean()))
    o = np.argsort(dur)[-4:]
    print("   slowest:", [(int(i), round(float(dur[i]), 1), "re-observed %d" % nre[i], "landmarks %d" % L0[i]) for i in o])

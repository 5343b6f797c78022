"""Dev tool: belief-step stage times as the trajectories grow (256 envs, 40 m map).
usage: bench_vs_poses.py [pose capacity = 206] [landmarks in the world = 100] [phases]     (PP_SAMPLE=49,59: sample these pose counts too)
With `phases`, the in-kernel phase stamps of block 0 of the SLAM kernel are printed too (k_slam_arrow beyond 42 poses)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine

n = 256
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 206
num_lm = int(sys.argv[2]) if len(sys.argv) > 2 else bench.NUM_LM
phases = len(sys.argv) > 3
cfg = default_config(bench.MAP, num_landmarks=num_lm, max_poses=cap, max_factors=14 * cap, max_snapshots=1,
                     max_landmarks=int(os.environ["PP_MAXLM"]) if os.environ.get("PP_MAXLM") else None)  # (PP_MAXLM: landmark capacity, selects the k_slam_arrow instantiation)
eng = Engine(cfg, n, 0, 0)
rng = np.random.RandomState(0)
starts = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-3, 3, n)], 1)
eng.reset(np.arange(n), np.arange(n), starts=starts)
loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
odoms = [torch.tensor([a] * n, dtype=torch.float64, device=eng.device) for a in loop]
names = ["tables+linearise", "blocks", "leaf factor", "leaf rhs down", "separator CR", "leaf up + selinv", "landmark system", "sweep", "lm out", "pose out"]
order = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]  # stamp slots in program order
out = (C.c_int64 * 64)()
extra_samples = [int(v) for v in os.environ.get("PP_SAMPLE", "").split(",") if v]  # more pose counts to sample (e.g. one whose update relinearises)
print("# %d envs, %d landmarks in the world, capacity %d poses" % (n, num_lm, cap))
for s in range(cap - 3):
    eng.step(odoms[s % len(loop)])
    p = s + 2
    if (p < 48 and p % 8 == 0) or p in (41, 42, 43) or (p >= 48 and p % 16 == 0) or p == cap - 2 or p in extra_samples:
        assert eng.status() == 0 or os.environ.get("DRLGX_LIB_DEV")  # (kernel-variant timing experiments compute nonsense on purpose)
        eng.snapshot(0)
        eng.timing_enable(2)
        for it in range(6):
            if it == 2:
                eng.timing_read()
            eng.restore(0)
            eng.step(odoms[(s + 1) % len(loop)])
        tm = eng.timing_read()
        eng.timing_enable(True)
        for it in range(6):
            if it == 2:
                eng.timing_read()
            eng.restore(0)
            eng.step(odoms[(s + 1) % len(loop)])
        tf = eng.timing_read()
        eng.timing_enable(False)
        extra = ""
        if phases and p > 42:
            eng.timing_enable(2)
            eng.L.drlgx_debug_phase_clocks_host(eng.h, 1, None)
            eng.restore(0)
            eng.inc_stats(True)
            eng.step(odoms[(s + 1) % len(loop)])
            n_inc, n_full = eng.inc_stats(True)
            eng.L.drlgx_debug_phase_clocks_host(eng.h, 0, out)
            eng.timing_enable(False)
            a = np.array(out[:11], dtype=np.float64)[order]
            d = (a[1:] - a[:-1]) / 100.0
            # (the stamps are the pose-chain solver's: an update served by the rank-k path leaves them untouched)
            if n_full and (d >= 0).all() and d.sum() < 1e4:
                extra = " | full solve (%d of %d envs), block 0: " % (n_full, n_inc + n_full) + ", ".join("%s %.1f" % (names[k], d[k]) for k in range(10))
            elif n_full:  # (block 0 itself may have been served by the rank-k path: no consistent stamps)
                extra = " | full solve (%d of %d envs)" % (n_full, n_inc + n_full)
            else:
                extra = " | rank-k update (%d of %d envs)" % (n_inc, n_inc + n_full)
        eng.restore(0)
        c = eng.counts_dev().cpu().numpy()
        ov = tm["t7"][0] / tm["t7"][1] * 1e3
        print("poses %3d -> %3d landmarks %.0f (max %d) factors %.0f: sim %.1f slam %.1f map %.1f us%s%s" % (
            p, p + 1, c[:, 1].mean(), c[:, 1].max(), c[:, 2].mean(), tm["sim"][0] / tm["sim"][1] * 1e3 - ov,
            tm["slam"][0] / tm["slam"][1] * 1e3 - ov, tm["map"][0] / tm["map"][1] * 1e3 - ov,
            (", fused %.1f us" % (tf["step"][0] / tf["step"][1] * 1e3 - ov)) if tf["step"][1] else "", extra), flush=True)
eng.close()

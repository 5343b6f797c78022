"""Dev: fused k_step vs stage kernels with the incremental path, per-step stats and first differences."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_belief import make_engine, generic_starts, SCRIPT
n = 5
fused, cfg = make_engine(n, num_landmarks=60)
staged, _ = make_engine(n, num_landmarks=60)
staged.timing_enable(2)
starts = generic_starts(n)
for e in (fused, staged):
    e.reset(np.arange(n), np.arange(n), starts=starts)
print("after reset", fused.inc_stats(), staged.inc_stats())
for s, act in enumerate(SCRIPT[:12]):
    odom = torch.tensor([act] * n, dtype=torch.float64, device=fused.device)
    fused.step(odom); staged.step(odom)
    d = 0.0
    for i in range(n):
        a, ai = fused.poses(i); b, bi = staged.poses(i)
        d = max(d, np.max(np.abs(a - b)))
        if np.max(np.abs(a - b)) > 0:
            w = np.argwhere(a != b)
            print("   env", i, "diff at", w.tolist()[:6], "counts", fused.counts(i))
    print("step", s, "stats", fused.inc_stats(), staged.inc_stats(), "max pose diff %.3e" % d)

"""Dev: per-env shape of the bench step for the incremental update: landmarks before the step, re-observed / new landmarks in it."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
eng.restore(0)
c0 = eng.counts_dev().cpu().numpy()
eng.step(odom)
c1 = eng.counts_dev().cpu().numpy()
L0, M0 = c0[:, 1], c0[:, 2]
nf = c1[:, 2] - M0
nn = c1[:, 1] - L0
nre = nf - nn
print("landmarks before: min %d mean %.1f max %d;  >= 25: %d envs, >= 29: %d, >= 33: %d, >= 40: %d" % (L0.min(), L0.mean(), L0.max(), (L0 >= 25).sum(), (L0 >= 29).sum(), (L0 >= 33).sum(), (L0 >= 40).sum()))
print("re-observed in the step: hist", np.bincount(nre), " > 8: %d envs" % (nre > 8).sum())
print("new landmarks in the step: hist", np.bincount(nn))
big = np.argsort(-L0)[:12]
print("largest: ", [(int(i), int(L0[i]), int(nre[i]), int(nn[i])) for i in big])

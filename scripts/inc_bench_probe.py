import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
eng, cfg = bench.make_engine(0, 0)
print("after warm-up", eng.inc_stats(True), eng.counts(0))
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
for k in range(3):
    eng.restore(0); eng.step(odom)
    print("restore+step", eng.inc_stats(True), eng.counts(0), eng.status())
eng.step(odom); print("step", eng.inc_stats(True), eng.counts(0))
eng.timing_enable(2)
eng.restore(0); eng.step(odom); print("staged restore+step", eng.inc_stats(True), eng.counts(0))
